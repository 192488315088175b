// Diagnostic: pure streaming-read bandwidth of the chip (what a sweep could get at best): every workgroup sums a
// contiguous span with 16-byte non-temporal loads, U loads in flight per lane.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, bool NT>
__global__ __launch_bounds__(256) void rd(const f32x4* __restrict__ x, size_t n4, float* out)
{
    const size_t per = n4 / gridDim.x;
    const f32x4* p = x + (size_t)blockIdx.x * per;
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i + (U - 1) * 256 < per; i += U * 256) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[blockIdx.x] = acc[0];
}

template <int U, bool NT>
void run(const f32x4* x, size_t n4, float* out, int grid, const char* name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<U, NT>), dim3(grid), dim3(256), 0, 0, x, n4, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("%-10s U=%2d grid=%5d: %.3f ms  %.2f TB/s\n", name, U, grid, best, (double)n4 * 16 / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)16 << 30;
    f32x4* x; float* out;
    hipMalloc(&x, bytes); hipMalloc(&out, 1 << 20);
    hipMemset(x, 0, bytes);
    const size_t n4 = bytes / 16;
    for (int grid : {256, 512, 1024, 2048, 8192}) {
        run<4, true>(x, n4, out, grid, "nt");
        run<8, true>(x, n4, out, grid, "nt");
        run<16, true>(x, n4, out, grid, "nt");
        run<8, false>(x, n4, out, grid, "plain");
    }
    return 0;
}
