// Diagnostic: read bandwidth as a function of the working set, to see what the memory-side cache (MALL / Infinity Cache, 256 MB)
// delivers to the CUs: every workgroup streams its contiguous slice of a buffer of S bytes, the whole grid passes over the buffer
// `reps` times inside ONE launch (so S <= cache sees the later passes hit).  nt = 1: non-temporal loads (what the sweeps use).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mall_bw tools/mall_bw.hip && tools/mall_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ x, size_t n16_per_wg, int reps, float* out)
{
    const f32x4* p = x + (size_t)blockIdx.x * n16_per_wg;
    f32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (size_t i = threadIdx.x; i + 7 * 256 < n16_per_wg; i += 8 * 256) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[blockIdx.x] = acc[0];
}

int main()
{
    const size_t cap = (size_t)4 << 30;
    char* x; float* out;
    if (hipMalloc(&x, cap) != hipSuccess || hipMalloc(&out, 1 << 20) != hipSuccess) { std::printf("alloc failed\n"); return 1; }
    hipMemset(x, 0, cap);
    const int grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::printf("%10s %6s %4s %10s %10s\n", "set MiB", "reps", "nt", "ms", "GB/s");
    for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096}) {
        const size_t S = mb << 20;
        const int reps = (int)(((size_t)16 << 30) / S);
        const size_t n16_per_wg = S / 16 / grid;
        for (int nt = 0; nt < 2; ++nt) {
            float best = 1e9;
            for (int t = 0; t < 3; ++t) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(rd<1>, dim3(grid), dim3(256), 0, 0, (const f32x4*)x, n16_per_wg, reps, out);
                else hipLaunchKernelGGL(rd<0>, dim3(grid), dim3(256), 0, 0, (const f32x4*)x, n16_per_wg, reps, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            std::printf("%10zu %6d %4d %10.3f %10.1f\n", mb, reps, nt, best, (double)S * reps / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
