// Diagnostic: how many float32 VALU FMACs can ride along 4 v_mfma_f32_32x32x2_f32 per loop iteration, at 2 waves/SIMD?
// Variants: plain v_fmac (VGPR operands), DPP row_newbcast v_fmac, interleaved vs grouped.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int I, int N, bool DPP>
__device__ __forceinline__ void fmacs(float (&v)[32], float p, float x)
{
    if constexpr (I < N) {
        if constexpr (DPP) asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(v[I % 32]) : "v"(p), "v"(x), "n"(I & 15));
        else asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[I % 32]) : "v"(p), "v"(x));
        fmacs<I + 1, N, DPP>(v, p, x);
    }
}

template <int NV, bool DPP, bool INTERLEAVE>
__global__ __launch_bounds__(256, 2) void burn(float* out, int iters, unsigned long long* clk)
{
    f32x16 acc[4];
    float v[32];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int i = 0; i < 32; ++i) v[i] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f + 0.5f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                fmacs<0, NV / 4, DPP>(v, a, b);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            fmacs<0, NV, DPP>(v, a, b);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 32; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NV, bool DPP, bool IL>
void run(const char* name)
{
    const int iters = 4000, grid = 512;
    float* out; unsigned long long* clk;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9; unsigned long long h[512];
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((burn<NV, DPP, IL>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(h, clk, grid * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < grid; ++i) cyc += h[i]; cyc /= grid;
    printf("%-28s NV=%3d  %.3f ms  cycles per iteration per wave %.0f (2 waves/SIMD -> per SIMD per iteration %.0f); MFMA-only would be %d\n",
           name, NV, best, cyc / iters, cyc / iters / 2, 4 * 64);
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0, false, false>("mfma only");
    run<28, false, true>("plain interleaved");
    run<28, true, true>("dpp interleaved");
    run<56, false, true>("plain interleaved");
    run<56, true, true>("dpp interleaved");
    run<112, false, true>("plain interleaved");
    run<112, true, true>("dpp interleaved");
    run<112, false, false>("plain grouped");
    run<112, true, false>("dpp grouped");
    return 0;
}
