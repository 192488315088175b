// Diagnostic: sustained v_mfma_f32_32x32x2_f32 rate of the whole chip for a few-ms burst, in the same
// shape as the sweeps (8 independent accumulators per wave), at 1 or 2 waves per SIMD, plus the in-kernel
// clock (s_memtime / s_memrealtime).  Not part of the library.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256, 2) void burn(float* out, int iters, unsigned long long* clk)
{
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        a += 1e-6f;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main()
{
    const int iters = 8000;
    for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
        const int grid = 256 * wgs_per_cu;
        float* out; unsigned long long* clk;
        hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 16);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(burn, dim3(grid), dim3(256), 0, 0, out, iters, clk);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid * 2);
            hipMemcpy(h.data(), clk, grid * 16, hipMemcpyDeviceToHost);
            double ghz = 0; for (int i = 0; i < grid; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; ghz /= grid;
            const double flops = (double)grid * 4 * iters * 8 * 4096.0;
            printf("wg/CU=%d rep=%d  %.3f ms  %.1f TFLOP/s  in-kernel clock %.3f GHz  (cycles/MFMA/SIMD = %.1f)\n", wgs_per_cu, rep, ms,
                   flops / ms / 1e9, ghz, (double)h[0] / (iters * 8.0 * wgs_per_cu));
        }
    }
    return 0;
}
