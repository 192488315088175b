// Diagnostic: is the K in (64, 128] x3 sweep (one workgroup per CU, one wave per SIMD, 512-column tiles) limited by the BYTES IT KEEPS IN
// FLIGHT?  The sweep's access pattern (a wave owns 128 columns = two 64-column groups; per group and 32-row stage 8 global_load_dwordx4
// per lane = 8 KiB per wave) with a register ring of D group-stages in flight and NM 16x16x32 bf16 MFMAs per 16-column tile between a
// group-stage's arrival and its re-issue (NM = 21: count data at K = 105, 42: full significands, 0: pure streaming).  stream_gemm_x3w_kernel
// <4, 1, *> is D = 2 (its 256 + 256 registers leave no room for more); the K <= 64 kernels keep 4 group-stages in flight per wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/inflight_bw tools/inflight_bw.hip && tools/inflight_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the sweep's in-register split of 8 float32 into three packed bf16 planes (kernels_x3.hpp: x3_split8_scalar)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{a, b}), bf16x2v)); }
__device__ __forceinline__ void split8(const float (&v)[8], u32x4 (&b)[3])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = v[2 * q], x1 = v[2 * q + 1];
        const unsigned hi = cvt2(x0, x1);
        const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
        const unsigned mid = cvt2(r0, r1);
        const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
        b[0][q] = hi; b[1][q] = mid; b[2][q] = cvt2(s0, s1);
    }
}

// SPLIT = 0: operands packed with 4 xors; 1: the real split, all NM MFMAs on plane 0, planes 1 / 2 consumed by one or; 2: the sweep's
// structure -- NM / 2 MFMAs on plane 0, the zero-plane test (wave-uniform branch), NM / 2 MFMAs on planes 1 / 2 when it fails
template <int D, int NM, int SPLIT = 0>
__global__ __launch_bounds__(256, 1) void rd(const float* __restrict__ S, long ldS, int n_tiles, int rows_total, int rows_per_wg, float* out)
{
    const long span0 = (long)blockIdx.x * rows_per_wg;           // (tile, row) space, tile-major, like the sweep's stream-K spans
    const int tile = (int)(span0 / rows_total);
    const int r0 = (int)(span0 % rows_total);
    if (tile >= n_tiles) return;
    const int r1 = min(rows_total, r0 + rows_per_wg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, kg = lane >> 4;
    const int nu = (r1 - r0) / 32 * 2;                           // group-stages of this span: unit u = (stage u / 2, group u % 2)
    auto addr = [&](int u, int e) { return S + (long)(r0 + 32 * (u >> 1) + 8 * kg + e) * ldS + tile * 512 + wave * 128 + 64 * (u & 1) + 4 * c16; };
    f32x4 x[D][8];
    f32x4 acc[8][8];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[m][i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[d][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addr(d < nu ? d : 0, e)));
    for (int u0 = 0; u0 < nu; u0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int u = u0 + d;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ti = (4 * d + t) % 8;
                if constexpr (SPLIT == 0) {
                    u32x4 b;
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = __float_as_uint(x[d][2 * q][t]) ^ (__float_as_uint(x[d][2 * q + 1][t]) >> 16);
                    if (NM == 0) { acc[7][t][0] += __uint_as_float(b[0] ^ b[1] ^ b[2] ^ b[3]); }
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        acc[m % 7][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, b), acc[m % 7][ti], 0, 0, 0);
                } else {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = x[d][e][t];
                    u32x4 b[3];
                    split8(v, b);
                    const unsigned rest = (b[1][0] | b[1][1] | b[1][2] | b[1][3]) & 0x7fff7fffu;
                    if constexpr (SPLIT == 1) {
                        acc[7][ti][0] += __uint_as_float(rest | b[2][0] | b[2][1] | b[2][2] | b[2][3]);
#pragma unroll
                        for (int m = 0; m < NM; ++m)
                            acc[m % 7][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[0]), acc[m % 7][ti], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int m = 0; m < NM / 2; ++m)
                            acc[m % 7][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[0]), acc[m % 7][ti], 0, 0, 0);
                        if (__builtin_amdgcn_ballot_w64(rest != 0u) != 0ull) {
#pragma unroll
                            for (int m = 0; m < NM / 2; ++m)
                                acc[m % 7][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[m < 14 ? 1 : 2]), __builtin_bit_cast(bf16x8, b[0]), acc[m % 7][ti], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int un = u + D < nu ? u + D : u;               // (the tail re-reads its own unit: same instruction stream)
#pragma unroll
            for (int e = 0; e < 8; ++e) x[d][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addr(un, e)));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[m][i][0] + acc[m][i][1] + acc[m][i][2] + acc[m][i][3];
#pragma unroll
    for (int d = 0; d < D; ++d) s += x[d][0][0];
    if (s == 12345.f) out[blockIdx.x] = s;
}

template <int M, int N, int VF>
struct Pat {
    static __device__ __forceinline__ void run()
    {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        if constexpr (M % VF == 0) __builtin_amdgcn_sched_group_barrier(0x2, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x2, 1, 0);
        if constexpr (M + 1 < N) Pat<M + 1, N, VF>::run();
    }
};

// SPLIT = 3: the sweep's structure SOFTWARE-PIPELINED over tiles: while the MFMAs of tile i run, the float32 of tile i + 1 are split
// (1-2 VALU per MFMA gap, pinned with sched_group_barrier); the zero-plane decision of tile i + 1 is taken at the END of tile i's block,
// so each block (hi-only: NM / 2 MFMAs, full: NM MFMAs) is straight-line code; a group's 8 loads are re-issued as soon as its last tile
// has been split, i.e. one tile's MFMAs earlier than in the unpipelined form.
template <int D, int NM, int VHI, int VFULL>
__global__ __launch_bounds__(256, 1) void rdp(const float* __restrict__ S, long ldS, int n_tiles, int rows_total, int rows_per_wg, float* out)
{
    const long span0 = (long)blockIdx.x * rows_per_wg;
    const int tile = (int)(span0 / rows_total);
    const int r0 = (int)(span0 % rows_total);
    if (tile >= n_tiles) return;
    const int r1 = min(rows_total, r0 + rows_per_wg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, kg = lane >> 4;
    const int nu = (r1 - r0) / 32 * 2;
    auto addr = [&](int u, int e) { return S + (long)(r0 + 32 * (u >> 1) + 8 * kg + e) * ldS + tile * 512 + wave * 128 + 64 * (u & 1) + 4 * c16; };
    f32x4 x[D][8];
    f32x4 acc[7][8];                                             // as in the sweep: every tile of a stage has its own 7 accumulator tiles (224 AGPRs)
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[m][i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[d][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addr(d < nu ? d : 0, e)));
    u32x4 bc[3];
    {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = x[0][e][0];
        split8(v, bc);
    }
    for (int u0 = 0; u0 < nu; u0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int u = u0 + d;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ti = (4 * d + t) % 8;
                const int dn = t < 3 ? d : (d + 1) % D, tn = (t + 1) & 3;
                if (t == 3) {                                     // every tile of group-stage u has been split: its registers are free
                    const int un = u + D < nu ? u + D : u;
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[d][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addr(un, e)));
                }
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x[dn][e][tn];
                const unsigned rest = (bc[1][0] | bc[1][1] | bc[1][2] | bc[1][3]) & 0x7fff7fffu;
                u32x4 bn[3];
                if (__builtin_amdgcn_ballot_w64(rest != 0u) == 0ull) {
#pragma unroll
                    for (int m = 0; m < NM / 2; ++m)
                        acc[m % 7][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[0]), __builtin_bit_cast(bf16x8, bc[0]), acc[m % 7][ti], 0, 0, 0);
                    split8(v, bn);
#pragma unroll
                    for (int m = 0; m < NM / 2; ++m) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x2, VHI, 0); }
                } else {
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        acc[m % 7][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bc[m < NM / 2 ? 0 : (m < NM / 2 + 14 ? 1 : 2)]), __builtin_bit_cast(bf16x8, bc[0]), acc[m % 7][ti], 0, 0, 0);
                    split8(v, bn);
                    Pat<0, NM, VFULL>::run();
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) bc[q] = bn[q];
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[m][i][0] + acc[m][i][1] + acc[m][i][2] + acc[m][i][3];
#pragma unroll
    for (int d = 0; d < D; ++d) s += x[d][0][0];
    s += __uint_as_float(bc[0][0] ^ bc[1][1] ^ bc[2][2]);
    if (s == 12345.f) out[blockIdx.x] = s;
}

template <int D, int NM, int SPLIT = 0, int VHI = 2, int VFULL = 8>
void run(const float* x, long ld, int rows_total, float* out)
{
    const int n_tiles = (int)(ld / 512), grid = 256;
    const long total = (long)n_tiles * rows_total;
    int rows_per_wg = (int)((total + grid - 1) / grid);
    rows_per_wg = (rows_per_wg + 32 * D - 1) / (32 * D) * (32 * D);
    const int g = (int)((total + rows_per_wg - 1) / rows_per_wg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        if constexpr (SPLIT == 3) hipLaunchKernelGGL((rdp<D, NM, VHI, VFULL>), dim3(g), dim3(256), 0, 0, x, ld, n_tiles, rows_total, rows_per_wg, out);
        else hipLaunchKernelGGL((rd<D, NM, SPLIT>), dim3(g), dim3(256), 0, 0, x, ld, n_tiles, rows_total, rows_per_wg, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n_tiles * 512 * 4.0 * rows_total;
    printf("split=%d (vhi %d vfull %d) D=%d group-stages (%2d KiB per wave) in flight, %2d MFMAs per tile: ld=%6ld rows=%6d grid=%3d  %.3f ms  %.2f TB/s\n", SPLIT, VHI, VFULL, D, 8 * D, NM, ld, rows_total, g, best,
           bytes / (best * 1e-3) / 1e12);
}

// TWO waves per SIMD (round 4, second half): 8 waves per workgroup, a wave owns ONE 64-column group x 7 component tiles (112 accumulator
// registers), D group-stages (32 rows each) of its group in flight; 256 registers per wave.  The question: does the hardware's own
// interleave of two waves hide the split's vector work behind the other wave's MFMAs (which pinning the schedule of ONE wave did not)?
template <int D, int NM, int SPLIT = 0>
__global__ __launch_bounds__(512, 1) void rd2(const float* __restrict__ S, long ldS, int n_tiles, int rows_total, int rows_per_wg, float* out)
{
    const long span0 = (long)blockIdx.x * rows_per_wg;
    const int tile = (int)(span0 / rows_total);
    const int r0 = (int)(span0 % rows_total);
    if (tile >= n_tiles) return;
    const int r1 = min(rows_total, r0 + rows_per_wg);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c16 = lane & 15, kg = lane >> 4;
    const int nu = (r1 - r0) / 32;                               // group-stages of this wave's group
    auto addr = [&](int u, int e) { return S + (long)(r0 + 32 * u + 8 * kg + e) * ldS + tile * 512 + wave * 64 + 4 * c16; };
    f32x4 x[D][8];
    f32x4 acc[7][4];
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[m][i] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int e = 0; e < 8; ++e) x[d][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addr(d < nu ? d : 0, e)));
    for (int u0 = 0; u0 < nu; u0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int u = u0 + d;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (SPLIT == 0) {
                    u32x4 b;
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[q] = __float_as_uint(x[d][2 * q][t]) ^ (__float_as_uint(x[d][2 * q + 1][t]) >> 16);
#pragma unroll
                    for (int m = 0; m < NM; ++m)
                        acc[m % 7][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b), __builtin_bit_cast(bf16x8, b), acc[m % 7][t], 0, 0, 0);
                } else {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = x[d][e][t];
                    u32x4 b[3];
                    split8(v, b);
                    const unsigned rest = (b[1][0] | b[1][1] | b[1][2] | b[1][3]) & 0x7fff7fffu;
                    if constexpr (SPLIT == 1) {
                        acc[6][t][0] += __uint_as_float(rest | b[2][0] | b[2][1] | b[2][2] | b[2][3]);
#pragma unroll
                        for (int m = 0; m < NM; ++m)
                            acc[m % 7][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[0]), acc[m % 7][t], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int m = 0; m < NM / 2; ++m)
                            acc[m % 7][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[0]), __builtin_bit_cast(bf16x8, b[0]), acc[m % 7][t], 0, 0, 0);
                        if (__builtin_amdgcn_ballot_w64(rest != 0u) != 0ull) {
#pragma unroll
                            for (int m = 0; m < NM / 2; ++m)
                                acc[m % 7][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, b[m < 14 ? 1 : 2]), __builtin_bit_cast(bf16x8, b[0]), acc[m % 7][t], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const int un = u + D < nu ? u + D : u;
#pragma unroll
            for (int e = 0; e < 8; ++e) x[d][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(addr(un, e)));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[m][i][0] + acc[m][i][1] + acc[m][i][2] + acc[m][i][3];
#pragma unroll
    for (int d = 0; d < D; ++d) s += x[d][0][0];
    if (s == 12345.f) out[blockIdx.x] = s;
}

template <int D, int NM, int SPLIT = 0>
void run2(const float* x, long ld, int rows_total, float* out)
{
    const int n_tiles = (int)(ld / 512), grid = 256;
    const long total = (long)n_tiles * rows_total;
    int rows_per_wg = (int)((total + grid - 1) / grid);
    rows_per_wg = (rows_per_wg + 32 * D - 1) / (32 * D) * (32 * D);
    const int g = (int)((total + rows_per_wg - 1) / rows_per_wg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd2<D, NM, SPLIT>), dim3(g), dim3(512), 0, 0, x, ld, n_tiles, rows_total, rows_per_wg, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)n_tiles * 512 * 4.0 * rows_total;
    printf("TWO WAVES PER SIMD split=%d D=%d group-stages (%2d KiB per wave, %3d per CU) in flight, %2d MFMAs per tile: ld=%6ld rows=%6d grid=%3d  %.3f ms  %.2f TB/s\n", SPLIT, D, 8 * D, 64 * D, NM, ld, rows_total, g, best,
           bytes / (best * 1e-3) / 1e12);
}

__global__ void fill_random(float* __restrict__ x, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761ull) ^ 12345u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        x[i] = (float)(h >> 8) * (1.0f / 16777216.0f) * 3.7f;
    }
}

int main()
{
    const size_t bytes = (size_t)11 << 30;
    float* x; float* out;
    hipMalloc(&x, bytes); hipMalloc(&out, 1 << 20);
    hipMemset(x, std::getenv("ONEPLANE") ? 0 : 0x3c, bytes);
    // RANDOMX=1: full 24-bit significands, every value different (the constant fill toggles few bits in the matrix pipe: a higher clock)
    if (std::getenv("RANDOMX")) hipLaunchKernelGGL(fill_random, dim3(8192), dim3(256), 0, 0, x, bytes / 4);   // 0x3c3c3c3c = 0.0115 (three non-zero planes); ONEPLANE=1: zeros (the zero-plane test passes)
    // cfg4's per-GPU share: X_gn = 20096 rows x 125056 cols (W^TX sweep), X_ng = 125056 rows x 20096 cols (XH^T sweep)
    for (int pass = 1; pass < 2; ++pass) {       // (pass 0 -- spans that cross a tile are cut at the tile's end: its byte count is wrong)
        const long ld = pass == 0 ? 125056 - 125056 % 512 : 20096 - 20096 % 512;
        const int rows = pass == 0 ? 20096 : 125056;
        run<2, 0>(x, ld, rows, out);  run<3, 0>(x, ld, rows, out);  run<4, 0>(x, ld, rows, out);  run<6, 0>(x, ld, rows, out);
        run<2, 21>(x, ld, rows, out); run<3, 21>(x, ld, rows, out); run<4, 21>(x, ld, rows, out); run<6, 21>(x, ld, rows, out);
        run<2, 42>(x, ld, rows, out); run<3, 42>(x, ld, rows, out); run<4, 42>(x, ld, rows, out);
        run<2, 21, 1>(x, ld, rows, out); run<3, 21, 1>(x, ld, rows, out); run<2, 42, 1>(x, ld, rows, out);
        run<2, 42, 2>(x, ld, rows, out); run<3, 42, 2>(x, ld, rows, out);
        if (std::getenv("TWO")) {
            for (int again = 0; again < 2; ++again) {
                run2<1, 0>(x, ld, rows, out); run2<2, 0>(x, ld, rows, out); run2<3, 0>(x, ld, rows, out);
                run2<1, 42>(x, ld, rows, out); run2<2, 42>(x, ld, rows, out); run2<3, 42>(x, ld, rows, out);
                run2<1, 42, 1>(x, ld, rows, out); run2<2, 42, 1>(x, ld, rows, out); run2<3, 42, 1>(x, ld, rows, out);
                run2<1, 42, 2>(x, ld, rows, out); run2<2, 42, 2>(x, ld, rows, out); run2<3, 42, 2>(x, ld, rows, out);
                run2<2, 21>(x, ld, rows, out); run2<2, 21, 1>(x, ld, rows, out);
                run<2, 42>(x, ld, rows, out); run<2, 42, 1>(x, ld, rows, out); run<2, 42, 2>(x, ld, rows, out);
            }
            continue;
        }
        run<2, 42, 3, 2, 8>(x, ld, rows, out); run<3, 42, 3, 2, 8>(x, ld, rows, out);
        run<2, 42, 3, 3, 8>(x, ld, rows, out); run<2, 42, 3, 2, 4>(x, ld, rows, out); run<2, 42, 3, 2, 100>(x, ld, rows, out); run<2, 42, 3, 1, 100>(x, ld, rows, out);
    }
    return 0;
}
