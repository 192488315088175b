// Diagnostic: do VALU ops (the and / sub / perm mix of an in-register bf16 split) ride along v_mfma_f32_32x32x16_bf16
// at 2 waves/SIMD, or are they additive as they are for the float32 MFMA (tools/coissue.hip)?  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int I, int N>
__device__ __forceinline__ void valu(unsigned (&v)[16], unsigned m)
{
    if constexpr (I < N) {
        if constexpr (I % 3 == 0) asm volatile("v_and_b32 %0, %1, %0" : "+v"(v[I % 16]) : "v"(m));
        else if constexpr (I % 3 == 1) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[I % 16]) : "v"(v[(I + 5) % 16]));
        else asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[I % 16]) : "v"(v[(I + 7) % 16]), "v"(m));
        valu<I + 1, N>(v, m);
    }
}

template <int NV, bool INTERLEAVE>
__global__ __launch_bounds__(256, 2) void burn(float* out, int iters, unsigned long long* clk)
{
    f32x16 acc[4];
    unsigned v[16];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 77u + i;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(blockIdx.x + e); }
    unsigned m = 0xffff0000u | (threadIdx.x & 1);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (INTERLEAVE) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                valu<0, NV / 4>(v, m);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            valu<0, NV>(v, m);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    for (int i = 0; i < 16; ++i) s += (float)v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NV, bool IL>
void run(const char* name)
{
    const int iters = 4000, grid = 512;
    float* out; unsigned long long* clk;
    hipMalloc(&out, grid * 256 * 4); hipMalloc(&clk, grid * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9; unsigned long long h[512];
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((burn<NV, IL>), dim3(grid), dim3(256), 0, 0, out, iters, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    hipMemcpy(h, clk, grid * 8, hipMemcpyDeviceToHost);
    double cyc = 0; for (int i = 0; i < grid; ++i) cyc += h[i]; cyc /= grid;
    const double tf = 512.0 * 4 * iters * 4 * (2.0 * 32 * 32 * 16) / (best * 1e-3) / 1e12;
    printf("%-14s 4 MFMA + %3d VALU: %.3f ms (%.0f TF bf16), cycles per iteration per wave %.0f (memtime ticks)\n", name, NV, best, tf, cyc / iters);
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0, false>("mfma only");
    run<16, true>("interleaved");
    run<32, true>("interleaved");
    run<64, true>("interleaved");
    run<64, false>("grouped");
    run<128, true>("interleaved");
    return 0;
}
