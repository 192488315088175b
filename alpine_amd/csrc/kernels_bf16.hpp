// bf16-storage path of the two streaming sweeps (BASELINE config 5): X and the MFMA operand copies of W / H are
// bf16, accumulation and every update stay float32 (master W, H, B are float32; kernels.hpp).
//
// Layout ("k-packed"): a matrix whose ROW index r is the contraction index is stored as
//     packed[r / 8][col][r % 8]            (bf16; 8 consecutive rows of one column are 16 contiguous bytes)
// v_mfma_f32_32x32x16_bf16 takes, per lane (i = lane & 31, hh = lane >> 5), the 8 values A[i][8hh..8hh+7] and
// B[8hh..8hh+7][i].  With the k-packed layout BOTH fragments are one aligned 16-byte load:
//     B (streamed X):   packedS[(r0/8 + hh)][f0 + i][0..7]   -- 32 lanes x 16 B = 512 contiguous bytes per half-wave,
//                       straight from HBM into the MFMA operand registers: no LDS, no transposes, no shuffles
//     A (panel W or H): packedP[(r0/8 + hh)][32m + i][0..7]  -- staged through LDS as a plain contiguous copy
// X is static, so the packing is paid once at ingest (pack_bf16_kernel); the panels are re-packed from the float32
// masters once per sweep (KP columns only, ~1 % of the traffic of the sweep).
// Arithmetic intensity is K flop/B, the bf16 MFMA rate is 16x the fp32 one: this path is HBM-bound by a wide margin.
#pragma once
#include "kernels.hpp"
#include <hip/hip_bf16.h>

namespace alpine {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float x)
{
    return __builtin_bit_cast(unsigned short, __float2bfloat16(x));     // round to nearest even, NaN stays NaN
}

// dst k-packed from a float32 row-major source tile: dst[(k / 8)][f][k % 8] = src[r][c] with
//   rows_are_k != 0:  k = k0 + r, f = f0 + c          rows_are_k == 0:  k = k0 + c, f = f0 + r        (k0 % 8 == 0)
// dst has `dst_cols` f-columns.  Block = 64 x 64 source tile through LDS; every 16-byte destination granule is
// written whole by one thread (granules whose 8 k's are not all inside the source get zeros for the missing ones).
__global__ __launch_bounds__(256)
void pack_bf16_kernel(const float* __restrict__ src, int64_t ld_src, int rows, int cols, unsigned short* __restrict__ dst,
                      int64_t dst_cols, int64_t k0, int64_t f0, int rows_are_k)
{
    __shared__ float tile[64][65];
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    for (int i = t; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(int64_t)(r0 + r) * ld_src + c0 + c] : 0.f;
    }
    __syncthreads();
    for (int i = t; i < 512; i += 256) {
        const int kb = i >> 6, fi = i & 63;                      // 8 k-blocks x 64 f positions of this tile
        unsigned short v[8];
        int64_t k_abs, f_abs;
        if (rows_are_k) {
            if (c0 + fi >= cols || r0 + 8 * kb >= rows) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f32_to_bf16_bits(tile[8 * kb + j][fi]);
            k_abs = k0 + r0 + 8 * kb; f_abs = f0 + c0 + fi;
        } else {
            if (r0 + fi >= rows || c0 + 8 * kb >= cols) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = f32_to_bf16_bits(tile[fi][8 * kb + j]);
            k_abs = k0 + c0 + 8 * kb; f_abs = f0 + r0 + fi;
        }
        u32x4 o = {(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                   (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
        *reinterpret_cast<u32x4*>(dst + ((k_abs / 8) * dst_cols + f_abs) * 8) = o;
    }
}

// sum of squares of a k-packed bf16 array (n8 granules of 8; optional second plane x2 added element-wise) in float64
__global__ __launch_bounds__(256)
void sqnorm_bf16_kernel(const unsigned short* __restrict__ x, const unsigned short* __restrict__ x2, int64_t n8,
                        double* __restrict__ part)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const u32x4 v = reinterpret_cast<const u32x4*>(x)[i];
        u32x4 v2 = {0u, 0u, 0u, 0u};
        if (x2) v2 = reinterpret_cast<const u32x4*>(x2)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float lo = __uint_as_float(v[j] << 16) + __uint_as_float(v2[j] << 16);
            const float hi = __uint_as_float(v[j] & 0xffff0000u) + __uint_as_float(v2[j] & 0xffff0000u);
            a += (double)lo * lo + (double)hi * hi;
        }
    }
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// Exact bf16 planes of a float32 value: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid).  hi + mid + lo == x
// for every finite float32 whose residuals stay normal (24 significand bits = 3 x 8); count-like data needs only hi
// (integers < 256) or hi + mid (16 significant bits, integers < 65536).
__device__ __forceinline__ void f32_split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo)
{
    hi = f32_to_bf16_bits(x);
    const float r1 = x - __uint_as_float((unsigned)hi << 16);
    mid = f32_to_bf16_bits(r1);
    const float r2 = r1 - __uint_as_float((unsigned)mid << 16);
    lo = f32_to_bf16_bits(r2);
}

// k-packed split of a float32 source tile into `planes` bf16 planes (plane p at dst + p * plane_stride), same
// geometry conventions as pack_bf16_kernel.  flags[0] |= 1 if some element is NOT exactly the sum of the stored planes,
// flags[1] |= 1 if some element has a non-zero second plane (so that a one-plane X can drop it).
__global__ __launch_bounds__(256)
void pack_split_kernel(const float* __restrict__ src, int64_t ld_src, int rows, int cols, unsigned short* __restrict__ dst,
                       int64_t plane_stride, int planes, int64_t dst_cols, int64_t k0, int64_t f0, int rows_are_k,
                       int* __restrict__ flags)
{
    __shared__ float tile[64][65];
    const int t = threadIdx.x;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    for (int i = t; i < 64 * 64; i += 256) {
        const int r = i >> 6, c = i & 63;
        tile[r][c] = (r0 + r < rows && c0 + c < cols) ? src[(int64_t)(r0 + r) * ld_src + c0 + c] : 0.f;
    }
    __syncthreads();
    int inexact = 0, has_mid = 0;
    for (int i = t; i < 512; i += 256) {
        const int kb = i >> 6, fi = i & 63;
        int64_t k_abs, f_abs;
        float v[8];
        if (rows_are_k) {
            if (c0 + fi >= cols || r0 + 8 * kb >= rows) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[8 * kb + j][fi];
            k_abs = k0 + r0 + 8 * kb; f_abs = f0 + c0 + fi;
        } else {
            if (r0 + fi >= rows || c0 + 8 * kb >= cols) continue;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = tile[fi][8 * kb + j];
            k_abs = k0 + c0 + 8 * kb; f_abs = f0 + r0 + fi;
        }
        unsigned short pl[3][8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32_split3(v[j], pl[0][j], pl[1][j], pl[2][j]);
            float sum = __uint_as_float((unsigned)pl[0][j] << 16);
            if (planes > 1) sum += __uint_as_float((unsigned)pl[1][j] << 16);
            if (planes > 2) sum += __uint_as_float((unsigned)pl[2][j] << 16);
            inexact |= (sum != v[j]);
            has_mid |= (pl[1][j] & 0x7fff) != 0;
        }
        for (int p = 0; p < planes; ++p) {
            u32x4 o = {(unsigned)pl[p][0] | ((unsigned)pl[p][1] << 16), (unsigned)pl[p][2] | ((unsigned)pl[p][3] << 16),
                       (unsigned)pl[p][4] | ((unsigned)pl[p][5] << 16), (unsigned)pl[p][6] | ((unsigned)pl[p][7] << 16)};
            *reinterpret_cast<u32x4*>(dst + p * plane_stride + ((k_abs / 8) * dst_cols + f_abs) * 8) = o;
        }
    }
    if (flags) {
        if (inexact) atomicOr(&flags[0], 1);
        if (has_mid) atomicOr(&flags[1], 1);
    }
}

// A wave-uniform 64-bit address as a SCALAR register pair (the compiler's divergence analysis already knows most of these are uniform; the
// readfirstlane pins it), and loads through it in the saddr form of global_load: uniform base + ONE 32-bit lane offset.  Per-lane 64-bit
// addresses of 8 rows (and of up to 12 panel pieces), which the compiler hoists out of a stage loop as loop invariants, cost 16 - 40
// vector registers in the kernels that have none to spare -- spilled, and re-loaded in the loop behind an s_waitcnt vmcnt(0).
__device__ __forceinline__ unsigned long long x3_uniform_u64(unsigned long long v)
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32)
           | (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
}
template <typename V>
__device__ __forceinline__ V sg_load_nt_saddr(const char* uniform_row, unsigned lane_off)
{
    typedef const __attribute__((address_space(1))) char* gchar_p;
    typedef const __attribute__((address_space(1))) V* gv_p;
    const unsigned long long urow = x3_uniform_u64(reinterpret_cast<unsigned long long>(uniform_row));
    asm volatile("" : "+v"(lane_off));          // (opaque: keeps the compiler from folding the lane offset into a hoisted per-lane 64-bit base)
    return __builtin_nontemporal_load(reinterpret_cast<gv_p>(reinterpret_cast<gchar_p>(urow) + lane_off));
}

// ----------------------------------------------------------------------------------------------
// stream_gemm on the bf16 matrix pipe: out[f][k] = sum_r S[r][f] * P[r][k] with S given as NPX bf16 planes and P as NPP
// bf16 planes (all k-packed).  Products of planes whose magnitudes matter at float32 precision are accumulated into the
// same float32 accumulators: term (xp, pp) is used iff xp + pp <= NPP - 1.
//   NPX = 1, NPP = 1 : plain bf16 operands (BASELINE config 5): operands rounded to bf16
//   NPX = 1, NPP = 3 : X exactly one bf16 plane (small integer counts), panel exact in 3 planes -> float32-grade result, 3 MFMAs
//   NPX = 2, NPP = 3 : X exact in two planes (16 significant bits), 5 MFMAs
// bf16 x bf16 products are exact in float32, so the split forms differ from the float32 MFMA path only in summation order.
// Same stream-K work division, pieces and wave tiling (NW waves x 128 f columns x all KP) as the float32 kernel; NW = 8
// (one 512-thread workgroup per CU, 1024-column tiles) halves the panel re-reads of NW = 4.  One
// k-step = 16 rows = one v_mfma_f32_32x32x16_bf16 depth; per k-step and X plane a wave issues 4 loads of 1 KiB.  The
// X ring holds BF_RING k-steps (16 KiB per wave in flight in every variant); one panel stage per ring pass.
// x_lane != ~0u: xnext is a WAVE-UNIFORM base (the tile's first column of the stage's first 8-row block) and the lane's part is the
// 32-bit byte offset x_lane (saddr loads: the two-waves-per-SIMD forms, which have no registers for per-lane 64-bit row addresses)
template <int KT, int NPX, int NPP, int BF_RING, bool LAST, int NJ = 4, bool SADDR = false>
__device__ __forceinline__ void bf_stage(f32x16 (&acc)[KT][NJ], u32x4 (&x)[BF_RING][NPX][NJ], const unsigned short* __restrict__ lrow,
                                         const unsigned short* __restrict__ xnext, int64_t f_stride8, int64_t x_plane, int lds_plane, unsigned x_lane = 0u)
{
    // Order inside a k-step: tile j outermost, so that the X register of tile j is dead after its NPP*KT (one-plane X)
    // MFMAs and is refilled at once -- a ring slot is in flight for (ring period - one tile's MFMAs) instead of (ring
    // period - one k-step's MFMAs): with the split forms' 3-5 MFMAs per load that is 7/8 instead of 1/2 of the ring in
    // flight, for the same registers.  A fragments (all panel planes of the k-step): double-buffered over k-steps when
    // there is one plane; with 3 planes a single set, each fragment re-read from LDS right after its last use (tile 3).
    constexpr int KP = 32 * KT;
    constexpr int NB = NPP == 1 ? 2 : 1;
    u32x4 a[NB][NPP][KT];
    auto lda = [&](int p, int pp, int m) {
        return *reinterpret_cast<const u32x4*>(lrow + pp * lds_plane + ((2 * p) * KP + 32 * m) * 8);
    };
#pragma unroll
    for (int pp = 0; pp < NPP; ++pp)
#pragma unroll
        for (int m = 0; m < KT; ++m) a[0][pp][m] = lda(0, pp, m);
#pragma unroll
    for (int p = 0; p < BF_RING; ++p) {
        const int cur = NB == 2 ? (p & 1) : 0;
        if (NB == 2 && p + 1 < BF_RING) {
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp)
#pragma unroll
                for (int m = 0; m < KT; ++m) a[cur ^ 1][pp][m] = lda(p + 1, pp, m);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int pp = 0; pp < NPP; ++pp) {
#pragma unroll
                for (int xp = 0; xp < NPX; ++xp) {
                    if (xp + pp <= NPP - 1) {
#pragma unroll
                        for (int m = 0; m < KT; ++m)
                            acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[cur][pp][m]),
                                                                                __builtin_bit_cast(bf16x8, x[p][xp][j]), acc[m][j], 0, 0, 0);
                    }
                }
                if (NB == 1 && j == NJ - 1 && p + 1 < BF_RING) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int m = 0; m < KT; ++m) a[0][pp][m] = lda(p + 1, pp, m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!LAST) {
#pragma unroll
                for (int xp = 0; xp < NPX; ++xp) {
                    if constexpr (SADDR)
                        x[p][xp][j] = sg_load_nt_saddr<u32x4>(reinterpret_cast<const char*>(xnext + xp * x_plane + (2 * p) * f_stride8 + j * (32 * 8)), x_lane);
                    else
                        x[p][xp][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xnext + xp * x_plane + (2 * p) * f_stride8 + j * (32 * 8)));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// NJ = 32-column sub-tiles per wave: 4 (a wave owns 128 columns x all KP: 256 accumulator registers at KT = 4, one wave per SIMD) or -- round 4,
// K > 64 -- 2 with NW = 8: the same 512-column workgroup tile from eight 64-column waves, TWO per SIMD (128 accumulator registers): the K > 64 forms
// ran at 0.41 of the HBM roof on one-plane X (3 MFMAs per load, issued by one wave per SIMD between its own loads and LDS reads).  The flush
// scratch of eight waves (135 KB at KT = 4) then shares its LDS with the panel stages (one more barrier per span).
template <int KT, int NPX, int NPP, int NW, int NJ = 4>
__global__ __launch_bounds__(64 * NW, (KT <= 2 && NW == 4 ? 2 : 1))
void stream_gemm_bf16_kernel(const unsigned short* __restrict__ S, int64_t x_plane, const unsigned short* __restrict__ P,
                             int64_t p_plane, const float* __restrict__ Pf, float* __restrict__ pieces, SweepGeom g, int* __restrict__ xcc_out)
{
    sg_report_xcc(xcc_out);
    // Panel source: NPP == 1 reads the rounded bf16 k-packed copy P (made by pack_bf16_kernel once per sweep);
    // NPP == 3 reads the float32 master Pf[R][KP] directly and splits it into its three exact bf16 planes while
    // staging it into LDS (4 instead of 6 bytes per panel element through the fabric, and no pack pass: every
    // workgroup re-reads the panel rows of its span, which is ~1/4 of the sweep's fabric traffic in the split forms).
    constexpr int KP = 32 * KT;
    constexpr int NT = 64 * NW;                                     // threads: NW waves x 128 f columns share one panel stage
    constexpr int BF_RING = NPP == 1 ? 4 : 2;                       // k-steps in the X ring (rounded bf16: 16 loads in flight;
                                                                    // split forms: 8 or 16 loads, 3-5x the MFMA work per load)
    constexpr int BF_ROWS = 16 * BF_RING;                           // rows per panel stage
    static_assert(SG_ROW_ALIGN % BF_ROWS == 0, "stream-K spans are multiples of one stage");
    constexpr int STAGE_BF16 = BF_ROWS * KP;                        // bf16 elements of one panel stage, per plane
    constexpr int GRAN = STAGE_BF16 / 8;                            // 16-byte granules per plane per stage
    constexpr int PV = NPP == 1 ? (GRAN + NT - 1) / NT : 1;     // granules per thread per stage (bf16 panel copy)
    constexpr int WAVE_F = 32 * NJ;
    constexpr bool ALIAS = NJ != 4;                                 // the flush scratch overlays the panel stages
    constexpr int LDS_PANEL_BYTES = 2 * NPP * STAGE_BF16 * 2, LDS_FLUSH_BYTES = NW * 32 * (KP + 4) * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[ALIAS ? (LDS_PANEL_BYTES > LDS_FLUSH_BYTES ? LDS_PANEL_BYTES : LDS_FLUSH_BYTES) : LDS_PANEL_BYTES + LDS_FLUSH_BYTES];
    unsigned short (*lds)[NPP * STAGE_BF16] = reinterpret_cast<unsigned short (*)[NPP * STAGE_BF16]>(smem);
    float (*flush_tr)[32 * (KP + 4)] = reinterpret_cast<float (*)[32 * (KP + 4)]>(smem + (ALIAS ? 0 : LDS_PANEL_BYTES));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    SgWalk walk;
    int team, member;                                               // round 4: workgroups in teams of one XCD (SweepGeom::gw, kernels.hpp): the members of a team walk
    sg_team_of_block(g, blockIdx.x, team, member);                  // the same contraction rows on adjacent column tiles and share the panel through that XCD's L2
    sg_walk_init(walk, g, team);
    constexpr int BLOCK_F = NW * WAVE_F;
    const int64_t f_stride8 = (int64_t)g.F * 8;                     // bf16 elements between consecutive 8-row blocks of S

    u32x4 preg[PV];                                                 // NPP == 1: staged bf16 granules
    float pf[NPP == 1 ? 1 : ((BF_ROWS / 8) * KP + NT - 1) / NT][8];   // NPP == 3: staged float32 panel values
    u32x4 x[BF_RING][NPX][NJ];

    int ft, r_begin, r_end;
    int64_t slot;
    while (sg_walk_next(walk, g, ft, r_begin, r_end, slot)) {
        const int nst = (r_end - r_begin) / BF_ROWS;
        const int wt = ft * g.gw + member;                 // this workgroup's BLOCK_F-wide tile (ft = the team's tile)
        if ((int64_t)wt * BLOCK_F >= g.F) continue;        // a member past the last column of a partly filled team tile: nothing to do (block-uniform)
        const int f0 = (wt * NW + wave) * WAVE_F;
        const bool active = f0 < g.F;

        constexpr int SETS = (BF_ROWS / 8) * KP;                   // (8-row block, column) granule positions of one stage
        constexpr int PVS = (SETS + NT - 1) / NT;
        const unsigned short* pptr = P + (g.panel_fixed ? (int64_t)0 : (int64_t)(r_begin / 8) * KP * 8);
        const float* pfptr = Pf + (g.panel_fixed ? (int64_t)0 : (int64_t)r_begin * KP);
        auto load_p = [&](int t) {
            if constexpr (NPP == 1) {
#pragma unroll
                for (int v = 0; v < PV; ++v) {
                    const int gi = tid + NT * v;                // granule index
                    if (GRAN % NT == 0 || gi < GRAN)
                        preg[v] = *reinterpret_cast<const u32x4*>(pptr + (int64_t)(g.panel_fixed ? 0 : t) * STAGE_BF16 + 8 * gi);
                }
            } else {
#pragma unroll
                for (int v = 0; v < PVS; ++v) {
                    const int si = tid + NT * v;                // set index = rb * KP + col
                    if (SETS % NT == 0 || si < SETS) {
                        const int rb = si / KP, col = si % KP;
                        const float* src = pfptr + ((int64_t)(g.panel_fixed ? 0 : t) * BF_ROWS + 8 * rb) * KP + col;
#pragma unroll
                        for (int e = 0; e < 8; ++e) pf[v][e] = src[e * KP];
                    }
                }
            }
        };
        auto store_p = [&](int b) {
            if constexpr (NPP == 1) {
#pragma unroll
                for (int v = 0; v < PV; ++v) {
                    const int gi = tid + NT * v;
                    if (GRAN % NT == 0 || gi < GRAN) *reinterpret_cast<u32x4*>(&lds[b][8 * gi]) = preg[v];
                }
            } else {
#pragma unroll
                for (int v = 0; v < PVS; ++v) {
                    const int si = tid + NT * v;
                    if (SETS % NT == 0 || si < SETS) {
                        unsigned short pl[3][8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) f32_split3(pf[v][e], pl[0][e], pl[1][e], pl[2][e]);
#pragma unroll
                        for (int q = 0; q < 3; ++q) {
                            u32x4 o = {(unsigned)pl[q][0] | ((unsigned)pl[q][1] << 16), (unsigned)pl[q][2] | ((unsigned)pl[q][3] << 16),
                                       (unsigned)pl[q][4] | ((unsigned)pl[q][5] << 16), (unsigned)pl[q][6] | ((unsigned)pl[q][7] << 16)};
                            *reinterpret_cast<u32x4*>(&lds[b][q * STAGE_BF16 + 8 * si]) = o;
                        }
                    }
                }
            }
        };

        __syncthreads();
        load_p(0);
        store_p(0);
        if (nst > 1) load_p(1);

        if (!active) {
            __syncthreads();
            for (int t = 0; t + 1 < nst; ++t) {
                store_p((t + 1) & 1);
                if (t + 2 < nst) load_p(t + 2);
                __syncthreads();
            }
            if (ALIAS) __syncthreads();                    // (the computing waves' barrier before their flush)
            continue;
        }

        f32x16 acc[KT][NJ];
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

        // this lane's granule of (row block r_begin/8 + h, column f0 + c); tile j is +32 columns, k-step p is +2 blocks
        // (SADDR: xrow is the wave-uniform part -- row block r_begin/8, column f0 -- and x_lane the lane's byte offset, < 2^32 for F <= 2^27)
        constexpr bool SADDR = NJ != 4;
        const unsigned x_lane = (unsigned)(((int64_t)h * g.F + c) * 16);
        const unsigned short* xrow = SADDR ? S + ((int64_t)(r_begin / 8) * g.F + f0) * 8 : S + ((int64_t)(r_begin / 8 + h) * g.F + f0 + c) * 8;
        const int64_t x_stage = (int64_t)(BF_ROWS / 8) * f_stride8;
        const int lds_lane = (h * KP + c) * 8;

        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < BF_RING; ++p)
#pragma unroll
            for (int xp = 0; xp < NPX; ++xp)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if constexpr (SADDR)
                        x[p][xp][j] = sg_load_nt_saddr<u32x4>(reinterpret_cast<const char*>(xrow + xp * x_plane + (2 * p) * f_stride8 + j * (32 * 8)), x_lane);
                    else
                        x[p][xp][j] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(xrow + xp * x_plane + (2 * p) * f_stride8 + j * (32 * 8)));
                }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        int t = 0;
        for (; t + 2 < nst; ++t) {
            store_p((t + 1) & 1);
            load_p(t + 2);
            __builtin_amdgcn_sched_barrier(0);
            bf_stage<KT, NPX, NPP, BF_RING, false, NJ, SADDR>(acc, x, &lds[t & 1][lds_lane], xrow + (t + 1) * x_stage, f_stride8, x_plane, STAGE_BF16, x_lane);
            __syncthreads();
        }
        if (t + 1 < nst) {
            store_p((t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            bf_stage<KT, NPX, NPP, BF_RING, false, NJ, SADDR>(acc, x, &lds[t & 1][lds_lane], xrow + (t + 1) * x_stage, f_stride8, x_plane, STAGE_BF16, x_lane);
            __syncthreads();
            ++t;
        }
        bf_stage<KT, NPX, NPP, BF_RING, true, NJ, SADDR>(acc, x, &lds[t & 1][lds_lane], xrow, f_stride8, x_plane, STAGE_BF16, x_lane);
        if (ALIAS) __syncthreads();                        // every wave has read its last panel stage: the flush scratch may overwrite it

        // D: row = k within tile m (8q + 4h + e), column = lane & 31 -> f_local = WAVE_F*wave + 32*j + c
        float* out = pieces + (slot * g.bf + member * BLOCK_F + wave * WAVE_F) * KP;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const f32x16* d[KT];
#pragma unroll
            for (int m = 0; m < KT; ++m) d[m] = &acc[m][j];
            sg_flush_tile<KT>(flush_tr[wave], d, out + (int64_t)(32 * j) * KP, KP, lane);
        }
    }
}

}  // namespace alpine
