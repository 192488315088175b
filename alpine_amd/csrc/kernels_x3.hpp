// Float32-grade sweeps on the bf16 matrix pipe for ARBITRARY float32 X ("x3"): out[f][k] = sum_r S[r][f] * P[r][k].
//
// A bf16 x bf16 product is exact in float32, and every float32 is exactly the sum of three bf16 planes (8 + 8 + 8
// significand bits: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)).  So
//     x * p = sum_{a,b} x_a * p_b ,   and the six terms with a + b <= 2 carry everything above 2^-24 |x p|
// (the three dropped terms are each below one float32 rounding of the product).  The float32 MFMA executes at the
// float32 vector rate (157 TF: tools/coissue.hip); six bf16 MFMAs cost 6/16 of one float32 MFMA, so the sweep turns
// from MFMA-bound (3.05 ms per launch at cfg3) into HBM-bound (16 GB: 2.0 ms at 8 TB/s) while X stays float32 in HBM,
// unlike the pre-split storage of kernels_bf16.hpp (which needs X to be exactly one or two planes).
//
// X is streamed as float32 exactly like stream_gemm_kernel (one global_load_dwordx4 per lane = 4 interleaved tiles,
// tile t / column j <-> f0 + 4j + t) and split into its planes IN REGISTERS right before use; the panel (W or H,
// float32 master) is split while it is staged into LDS, as in the NPP = 3 form of stream_gemm_bf16_kernel.
//
// Tiling: 4 waves per workgroup, ONE workgroup per CU (1 wave per SIMD, up to 512 registers per lane).  A wave owns
// NH halves of 128 f columns x all KP components = 4*NH x KT accumulator tiles of v_mfma_f32_32x32x16_bf16 -- 256
// accumulator registers in both shapes: K <= 64 -> NH = 2 (256 columns per wave, 1024-column workgroup tiles: half the
// panel re-reads of 512-column tiles), K <= 128 -> NH = 1.  Ring of 2 k-steps (16 rows each) of float32 X: 16 KiB per
// half in flight per wave.  Six bf16 MFMAs per fragment pair cost 6/16 of one float32 MFMA, so the sweep stays near the
// HBM bound up to K = 128 while the float32-MFMA sweep's time grows with K.
#pragma once
#include <type_traits>
#include "kernels_bf16.hpp"

namespace alpine {

constexpr int X3_RING = 2;                  // k-steps in the X ring
constexpr int X3_ROWS = 16 * X3_RING;       // rows per panel stage

// exact planes of 8 consecutive rows of one column, packed as MFMA B operands (8 bf16 = 4 dwords per plane).  Works on
// PAIRS so that every conversion is one v_cvt_pk_bf16_f32 whose result already is the packed operand dword (even row
// in the low half): per pair 3 conversions + 4 bit operations + 2 packed subtractions.  Same values as f32_split3.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x3_cvt_pk(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2v)); }
__device__ __forceinline__ f32x2 x3_unpack(unsigned p) { return f32x2{__uint_as_float(p << 16), __uint_as_float(p & 0xffff0000u)}; }

__device__ __forceinline__ void x3_split8(const float (&v)[8], u32x4 (&b)[3])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2 x = {v[2 * q], v[2 * q + 1]};
        const unsigned hi = x3_cvt_pk(x);
        const f32x2 r1 = x - x3_unpack(hi);
        const unsigned mid = x3_cvt_pk(r1);
        const f32x2 r2 = r1 - x3_unpack(mid);
        b[0][q] = hi; b[1][q] = mid; b[2][q] = x3_cvt_pk(r2);
    }
}

// The same split written on scalars (used by the 16x16x32 form below): no float2 temporaries, so the compiler neither needs
// register pairs (v_mov copies) nor forms v_pk_add_f32; 11 single-issue VALU ops per pair of values.  Same values as x3_split8.
// (A/B on the 32x32x16 kernel, together with sched_group_barrier interleaves of the split between the MFMAs: 0.4 - 2 %
// SLOWER than the compiler's own schedule of the float2 form, so that kernel keeps x3_split8.  A second attempt -- the split
// written stage-wise, one tile ahead, "1 MFMA : 5 VALU" pinned -- was 3 % faster on full significands, 2.5 % slower on counts;
// the 16x16x32 form below is faster than either on full significands, and pinning ITS schedule loses 4 %: DESIGN.md 4.2c.)
__device__ __forceinline__ unsigned x3_cvt2(float a, float b) { return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{a, b}), bf16x2v)); }
__device__ __forceinline__ void x3_split8_scalar(const float (&v)[8], u32x4 (&b)[3])
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x0 = v[2 * q], x1 = v[2 * q + 1];
        const unsigned hi = x3_cvt2(x0, x1);
        const float r0 = x0 - __uint_as_float(hi << 16), r1 = x1 - __uint_as_float(hi & 0xffff0000u);
        const unsigned mid = x3_cvt2(r0, r1);
        const float s0 = r0 - __uint_as_float(mid << 16), s1 = r1 - __uint_as_float(mid & 0xffff0000u);
        b[0][q] = hi; b[1][q] = mid; b[2][q] = x3_cvt2(s0, s1);
    }
}

// (x3_uniform_u64 / sg_load_nt_saddr: kernels_bf16.hpp)
__device__ __forceinline__ f32x4 x3_load_nt_saddr(const char* uniform_row, unsigned lane_off) { return sg_load_nt_saddr<f32x4>(uniform_row, lane_off); }

// One panel stage = X3_RING k-steps.  Per k-step: A fragments of all three panel planes (double-buffered over k-steps),
// then per 128-column half: 4 x (split one tile's 8 x float32 into planes, 6*KT MFMAs), then the half's 8 loads are
// re-issued for the same k-step of the next stage.
template <int KT, int NH, bool LAST, int ABL = 0>
__device__ __forceinline__ void x3_stage(f32x16 (&acc)[KT][4 * NH], f32x4 (&x)[X3_RING][NH][8], const unsigned short* __restrict__ lrow,
                                         const float* __restrict__ xnext0, const float* __restrict__ xnext1, int64_t ldS, int lds_plane)
{
    constexpr int KP = 32 * KT;
    u32x4 a[2][3][KT];
    auto lda = [&](int p, int pp, int m) {
        return *reinterpret_cast<const u32x4*>(lrow + pp * lds_plane + ((2 * p) * KP + 32 * m) * 8);
    };
#pragma unroll
    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
        for (int m = 0; m < KT; ++m) a[0][pp][m] = lda(0, pp, m);
#pragma unroll
    for (int p = 0; p < X3_RING; ++p) {
        const int cur = p & 1;
        if (p + 1 < X3_RING) {
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int m = 0; m < KT; ++m) a[cur ^ 1][pp][m] = lda(p + 1, pp, m);
        }
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x[p][hf][e][t];
                u32x4 b[3];
                if constexpr (ABL == 2) {                      // timing-only ablation: no split (raw bits as operands)
#pragma unroll
                    for (int q = 0; q < 3; ++q) b[q] = u32x4{__float_as_uint(v[0]) + __float_as_uint(v[1]), __float_as_uint(v[2]) + __float_as_uint(v[3]),
                                                             __float_as_uint(v[4]) + __float_as_uint(v[5]), __float_as_uint(v[6]) + __float_as_uint(v[7])};
                } else {
                    x3_split8(v, b);
                }
                // products with the hi plane of X: always
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                    for (int m = 0; m < KT; ++m)
                        acc[m][4 * hf + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[cur][pp][m]),
                                                                                     __builtin_bit_cast(bf16x8, b[0]), acc[m][4 * hf + t], 0, 0, 0);
                // products with the mid / lo planes: nothing to add when the whole 16 x 32 tile of X is exactly one bf16
                // plane (small integer counts, zeros) -- a wave-uniform test, no loads inside the branch
                const unsigned rest = (b[1][0] | b[1][1] | b[1][2] | b[1][3]) & 0x7fff7fffu;
                if (ABL != 3 && __builtin_amdgcn_ballot_w64(rest != 0u) != 0ull) {
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int m = 0; m < KT; ++m)
                            acc[m][4 * hf + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[cur][pp][m]),
                                                                                         __builtin_bit_cast(bf16x8, b[1]), acc[m][4 * hf + t], 0, 0, 0);
#pragma unroll
                    for (int m = 0; m < KT; ++m)
                        acc[m][4 * hf + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[cur][0][m]),
                                                                                     __builtin_bit_cast(bf16x8, b[2]), acc[m][4 * hf + t], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!LAST) {
                const float* src = (hf == 0 ? xnext0 : xnext1) + (int64_t)(16 * p) * ldS;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    x[p][hf][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)e * ldS));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int KT, int NH, int ABL = 0>
__global__ __launch_bounds__(256, 1)
void stream_gemm_x3_kernel(const float* __restrict__ S, const float* __restrict__ Pf, float* __restrict__ pieces,
                           int64_t ldS, SweepGeom g, int* __restrict__ xcc_out)
{
    sg_report_xcc(xcc_out);
    static_assert(KT * NH <= 4, "256 accumulator registers per lane");
    constexpr int KP = 32 * KT;
    constexpr int WAVE_F = 128 * NH, BLOCK_F = 4 * WAVE_F;
    constexpr int NT = 256;
    static_assert(SG_ROW_ALIGN % X3_ROWS == 0, "stream-K spans are multiples of one stage");
    constexpr int STAGE_BF16 = X3_ROWS * KP;                        // bf16 elements of one panel stage, per plane
    constexpr int SETS = (X3_ROWS / 8) * KP;                        // (8-row block, column) granule positions of one stage
    constexpr int PVS = (SETS + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][3 * STAGE_BF16];
    __shared__ __attribute__((aligned(16))) float flush_tr[4][32 * (KP + 4)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    SgWalk walk;
    int team, member;
    sg_team_of_block(g, blockIdx.x, team, member);
    sg_walk_init(walk, g, team);
    SG_STAMP_SET(0, SG_NOW());
    unsigned long long st_flush = 0, st_first = 0, st_segs = 0;       // stamps build only
    (void)st_flush; (void)st_first; (void)st_segs;

    float pf[PVS][8];
    f32x4 x[X3_RING][NH][8];

    int ft, r_begin, r_end;
    int64_t slot;
    while (sg_walk_next(walk, g, ft, r_begin, r_end, slot)) {
        const int nst = (r_end - r_begin) / X3_ROWS;
        const int wt = ft * g.gw + member;                 // this workgroup's BLOCK_F-wide tile (ft = the team's tile)
        if ((int64_t)wt * BLOCK_F >= g.F) continue;        // a member past the last column of a partly filled team tile: nothing to do (block-uniform)
        const int f0 = (wt * 4 + wave) * WAVE_F;
        const bool active = f0 < g.F;
        // F is a multiple of 128, not of 256: a wave whose second half lies outside re-reads its first half there (valid
        // memory; those accumulators land in piece columns >= F, which no consumer reads)
        const int f1 = (NH == 2 && f0 + 128 < g.F) ? f0 + 128 : f0;

#ifdef ALPINE_DIAGNOSTICS
        const bool pfix = g.panel_fixed == 1;          // timing-only ablation: every panel stage re-reads the span's FIRST stage (cache-resident,
        const float* pfptr = Pf + (int64_t)r_begin * KP;   // a different address per workgroup: no hot spot in one L2 channel)
#else
        constexpr bool pfix = false;
        const float* pfptr = Pf + (int64_t)r_begin * KP;
#endif
        auto load_p = [&](int t) {
            if (pfix) t = 0;
#pragma unroll
            for (int v = 0; v < PVS; ++v) {
                const int si = tid + NT * v;                        // set index = rb * KP + col
                if (SETS % NT == 0 || si < SETS) {
                    const int rb = si / KP, col = si % KP;
                    const float* src = pfptr + ((int64_t)t * X3_ROWS + 8 * rb) * KP + col;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[v][e] = src[e * KP];
                }
            }
        };
        auto store_p = [&](int b) {
#pragma unroll
            for (int v = 0; v < PVS; ++v) {
                const int si = tid + NT * v;
                if (SETS % NT == 0 || si < SETS) {
                    u32x4 o[3];
                    x3_split8(pf[v], o);
#pragma unroll
                    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(&lds[b][q * STAGE_BF16 + 8 * si]) = o[q];
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
            continue;
        }

        f32x16 acc[KT][4 * NH];
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int j = 0; j < 4 * NH; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

        // this lane's float4 of row r_begin + 8h (+ e), columns f + 4c..4c+3 (element t -> tile t, column c)
        const float* xrow0 = S + (int64_t)(r_begin + 8 * h) * ldS + f0 + 4 * c;
        const float* xrow1 = S + (int64_t)(r_begin + 8 * h) * ldS + f1 + 4 * c;
        const int64_t x_stage = (int64_t)X3_ROWS * ldS;
        const int lds_lane = (h * KP + c) * 8;

        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < X3_RING; ++p)
#pragma unroll
            for (int hf = 0; hf < NH; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    x[p][hf][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>((hf == 0 ? xrow0 : xrow1) + (int64_t)(16 * p + e) * ldS));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (st_first == 0) st_first = SG_NOW();

        int t = 0;
        for (; t + 2 < nst; ++t) {
            store_p((t + 1) & 1);
            load_p(t + 2);
            __builtin_amdgcn_sched_barrier(0);
            x3_stage<KT, NH, false, ABL>(acc, x, &lds[t & 1][lds_lane], xrow0 + (t + 1) * x_stage, xrow1 + (t + 1) * x_stage, ldS, STAGE_BF16);
            if constexpr (ABL != 1) __syncthreads();           // ABL == 1: timing-only ablation of the stage barrier
        }
        if (t + 1 < nst) {
            store_p((t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            x3_stage<KT, NH, false>(acc, x, &lds[t & 1][lds_lane], xrow0 + (t + 1) * x_stage, xrow1 + (t + 1) * x_stage, ldS, STAGE_BF16);
            __syncthreads();
            ++t;
        }
        x3_stage<KT, NH, true>(acc, x, &lds[t & 1][lds_lane], xrow0, xrow1, ldS, STAGE_BF16);

        // D: row = k within tile m (8q + 4h + e), column = lane & 31 = c -> f_local = WAVE_F*wave + 128*hf + 4c + t
        float* out = pieces + (slot * g.bf + member * BLOCK_F + wave * WAVE_F) * KP;
        const unsigned long long st_f0 = SG_NOW();
        SG_STAMP_SET(2, st_f0);
#pragma unroll
        for (int hf = 0; hf < NH; ++hf) {
            if ((hf == 1 && f1 == f0) || g.panel_fixed == 2) break;    // second half outside F: nothing to write (panel_fixed == 2: timing-only ablation of the flush)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f32x16* d[KT];
#pragma unroll
                for (int m = 0; m < KT; ++m) d[m] = &acc[m][4 * hf + tt];
                sg_flush_tile<KT>(flush_tr[wave], d, out + (int64_t)(128 * hf + tt) * KP, 4 * KP, lane);
            }
        }
        st_flush += SG_NOW() - st_f0;
        ++st_segs;
    }
    SG_STAMP_SET(1, st_first);
    SG_STAMP_SET(3, SG_NOW());
    SG_STAMP_SET(4, st_flush);
    SG_STAMP_SET(5, st_segs);
#ifdef ALPINE_STAMPS
    {
        unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); SG_STAMP_SET(6, (unsigned long long)(xcc & 15u));
        // launch history: workgroup 0 claims the launch's slot, workgroup 1 adds its XCC id to the same slot (it reads the launch
        // count before workgroup 0 of the NEXT launch can bump it: launches on one stream do not overlap)
        if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned n = atomicAdd(&alpine::g_sweep_hist[0], 1u); atomicOr(&alpine::g_sweep_hist[4 + (n & 4095)], (xcc & 15u) | (gridDim.x << 16)); }
        if (threadIdx.x == 0 && blockIdx.x == 1) alpine::g_sweep_hist[2] = xcc & 15u;
    }
#endif
}

// ----------------------------------------------------------------------------------------------------------------------
// The same sweep on v_mfma_f32_16x16x32_bf16 ("x3w").  Same X stream (float32, one global_load_dwordx4 per lane and row),
// same in-register split, same six plane products, same pieces -- only the matrix instruction differs: 16 x 16 output
// tiles with a 32-deep contraction instead of 32 x 32 x 16.  MFMA cycles per element are identical (16 cycles per 16x16x32
// vs 32 per 32x32x16), but a 16 x 16 accumulator tile is read and written 256 values per 8192 multiply-adds instead of
// 1024 per 16384, and on data with full significands -- where the chip lowers its clock under the matrix load
// (MI355X_MICROARCH.md, "DVFS give-back" item 7) -- the chip holds a higher clock on this shape.  Results are NOT
// bitwise those of the 32x32x16 form (the order in which a row's 16 / 32 products are added differs); same accuracy.
//
// Lane (c16 = lane & 15, kg = lane >> 4).  B operand of tile t of a 64-column group: rows r + 8 kg + j (j = 0..7) of column
// f_cg + 4 c16 + t -> one float4 per row feeds the group's four interleaved 16-column tiles; a wave covers 32 rows x 64
// columns with 8 loads (256-byte row segments).  A operand: components 16 m + c16, the same 8 rows, from the k-packed
// LDS image [row / 8][component][row % 8].  One panel stage = one k-step of 32 rows; the X registers hold ONE k-step
// (32 KiB per wave, as before), each 64-column group refilled for the next k-step right after its last split.
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int KT>
__device__ __forceinline__ void sg_flush_tile16(float* __restrict__ tr, const f32x4 (&d)[2 * KT], float* __restrict__ out_row0,
                                                int64_t row_stride, int lane)
{
    // 16 columns x KP components in C/D layout (lane (c16, kg) holds components 16 m + 4 kg + e of column c16) -> 16 rows of a
    // piece with whole-row stores, through a wave-private scratch tr[16][KP + 4]
    constexpr int KP = 32 * KT, LD = KP + 4, Q4 = KP / 4, M16 = 2 * KT;
    const int c16 = lane & 15, kg = lane >> 4;
#pragma unroll
    for (int m = 0; m < M16; ++m) *reinterpret_cast<f32x4*>(&tr[c16 * LD + 16 * m + 4 * kg]) = d[m];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < (16 * Q4) / 64; ++i) {
        const int idx = i * 64 + lane, r = idx / Q4, c4 = idx % Q4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tr[r * LD + 4 * c4]);
        *reinterpret_cast<f32x4*>(out_row0 + (int64_t)r * row_stride + 4 * c4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// M16A = 16-component tiles that hold real components (ceil(K / 16) <= 2 KT): the padding tile of a model with
// K <= 32 KT - 16 (K = 105 -> 7 of 8 tiles) is never multiplied -- its accumulators stay zero and are flushed as zeros, so the
// pieces keep their [.][KP] layout and every consumer is unchanged.  12.5 % fewer MFMAs at K = 105.
// (A variant WITHOUT the zero-plane test -- on full significands it always fails, and without the branch a 64-column group
// is one basic block for the scheduler -- was 3 % slower in A/B, cfg3 and K = 105: the test stays.)
// ONEPLANE = true (alpine_finalize_X's census found EVERY element of X to be exactly one bf16 plane -- integer counts < 256 and the
// like): no split and no zero-plane test, one v_cvt_pk_bf16_f32 per pair of values (exact on such data) and the three hi-plane
// products -- the same products in the same order as the general form executes on that data, so bitwise the same pieces.
template <int KT, int NH, int M16A = 2 * KT, bool ONEPLANE = false, bool NOTEST = false>
__global__ __launch_bounds__(256, 1)
void stream_gemm_x3w_kernel(const float* __restrict__ S, const float* __restrict__ Pf, float* __restrict__ pieces,
                            int64_t ldS, SweepGeom g, int* __restrict__ xcc_out)
{
    sg_report_xcc(xcc_out);
    static_assert(KT * NH <= 4, "256 accumulator registers per lane");
    static_assert(M16A >= 1 && M16A <= 2 * KT, "active 16-component tiles");   // < 2 KT - 1 only for the second half of a wide model (K = 150: 2 of 8)
    constexpr int KP = 32 * KT, M16 = 2 * KT;
    constexpr int WAVE_F = 128 * NH, BLOCK_F = 4 * WAVE_F, NCG = 2 * NH;       // 64-column groups per wave
    constexpr int NT = 256;
    constexpr int ROWS = 32;                                          // rows per k-step = per panel stage
    static_assert(SG_ROW_ALIGN % ROWS == 0, "stream-K spans are multiples of one stage");
    constexpr int STAGE_BF16 = ROWS * KP;
    constexpr int SETS = (ROWS / 8) * KP;
    constexpr int PVS = (SETS + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][3 * STAGE_BF16];
    __shared__ __attribute__((aligned(16))) float flush_tr[4][16 * (KP + 4)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kg = lane >> 4;
    SgWalk walk;
    int team, member;
    sg_team_of_block(g, blockIdx.x, team, member);
    sg_walk_init(walk, g, team);

    float pf[PVS][8];
    f32x4 x[NCG][8];

    int ft, r_begin, r_end;
    int64_t slot;
    while (sg_walk_next(walk, g, ft, r_begin, r_end, slot)) {
        const int nst = (r_end - r_begin) / ROWS;
        const int wt = ft * g.gw + member;                 // this workgroup's BLOCK_F-wide tile (ft = the team's tile)
        if ((int64_t)wt * BLOCK_F >= g.F) continue;        // a member past the last column of a partly filled team tile (block-uniform)
        const int f0 = (wt * 4 + wave) * WAVE_F;
        const bool active = f0 < g.F;
        // F is a multiple of 128, not of 256: a wave whose second half lies outside re-reads its first half there (valid
        // memory; those accumulators are never written out)
        const bool half2 = NH == 2 && f0 + 128 < g.F;

        const float* pfptr = Pf + (int64_t)r_begin * KP;
#ifdef ALPINE_DIAGNOSTICS
        const bool pfix = g.panel_fixed == 1;          // timing-only ablation: every panel stage re-reads the span's FIRST stage (cache-resident)
#else
        constexpr bool pfix = false;
#endif
        auto load_p = [&](int t) {
            if (pfix) t = 0;
#pragma unroll
            for (int v = 0; v < PVS; ++v) {
                const int si = tid + NT * v;                        // set index = rb * KP + col
                if (SETS % NT == 0 || si < SETS) {
                    const int rb = si / KP, col = si % KP;
                    const float* src = pfptr + ((int64_t)t * ROWS + 8 * rb) * KP + col;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[v][e] = src[e * KP];
                }
            }
        };
        auto store_p = [&](int b) {
#pragma unroll
            for (int v = 0; v < PVS; ++v) {
                const int si = tid + NT * v;
                if (SETS % NT == 0 || si < SETS) {
                    u32x4 o[3];
                    x3_split8_scalar(pf[v], o);
#pragma unroll
                    for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(&lds[b][q * STAGE_BF16 + 8 * si]) = o[q];
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
            continue;
        }

        f32x4 acc[M16][4 * NCG];
#pragma unroll
        for (int m = 0; m < M16; ++m)
#pragma unroll
            for (int j = 0; j < 4 * NCG; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // this lane's float4 of row r_begin + 8 kg (+ e), columns f_cg + 4 c16 .. + 3 (element t -> tile t, column c16)
        const float* xbase[NCG];
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            const int fcg = (cg >= 2 && !half2) ? f0 + 64 * (cg - 2) : f0 + 64 * cg;
            xbase[cg] = S + (int64_t)(r_begin + 8 * kg) * ldS + fcg + 4 * c16;
        }
        const int64_t x_stage = (int64_t)ROWS * ldS;
        const int lds_lane = (kg * KP + c16) * 8;

        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                x[cg][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xbase[cg] + (int64_t)e * ldS));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        // one stage: A fragments of the three panel planes, then per 64-column group: 4 x (split one tile's 8 float32 into
        // planes, 6 * M16 MFMAs), then the group's 8 loads are re-issued for the next stage
        auto stage = [&](const unsigned short* __restrict__ lrow, int t_next, bool last) {
            u32x4 a[3][M16A];
#pragma unroll
            for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                for (int m = 0; m < M16A; ++m)
                    a[pp][m] = *reinterpret_cast<const u32x4*>(lrow + pp * STAGE_BF16 + (16 * m) * 8);
#pragma unroll
            for (int cg = 0; cg < NCG; ++cg) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = x[cg][e][t];
                    u32x4 b[3];
                    if constexpr (ONEPLANE) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) b[0][q] = x3_cvt2(v[2 * q], v[2 * q + 1]);
                        b[1] = b[2] = u32x4{0u, 0u, 0u, 0u};
                    } else {
                        x3_split8_scalar(v, b);
                    }
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                        for (int m = 0; m < M16A; ++m)
                            acc[m][4 * cg + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[pp][m]),
                                                                                         __builtin_bit_cast(bf16x8, b[0]), acc[m][4 * cg + t], 0, 0, 0);
                    // mid / lo planes of X: nothing to add when the whole 32 x 16 tile is exactly one bf16 plane (wave-uniform test)
                    const unsigned rest = (b[1][0] | b[1][1] | b[1][2] | b[1][3]) & 0x7fff7fffu;
                    if (!ONEPLANE && (NOTEST || __builtin_amdgcn_ballot_w64(rest != 0u) != 0ull)) {
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                            for (int m = 0; m < M16A; ++m)
                                acc[m][4 * cg + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[pp][m]),
                                                                                             __builtin_bit_cast(bf16x8, b[1]), acc[m][4 * cg + t], 0, 0, 0);
#pragma unroll
                        for (int m = 0; m < M16A; ++m)
                            acc[m][4 * cg + t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[0][m]),
                                                                                         __builtin_bit_cast(bf16x8, b[2]), acc[m][4 * cg + t], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!last) {
                    const float* src = xbase[cg] + (int64_t)t_next * x_stage;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        x[cg][e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (int64_t)e * ldS));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };

        int t = 0;
        for (; t + 2 < nst; ++t) {
            store_p((t + 1) & 1);
            load_p(t + 2);
            __builtin_amdgcn_sched_barrier(0);
            stage(&lds[t & 1][lds_lane], t + 1, false);
            __syncthreads();
        }
        if (t + 1 < nst) {
            store_p((t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            stage(&lds[t & 1][lds_lane], t + 1, false);
            __syncthreads();
            ++t;
        }
        stage(&lds[t & 1][lds_lane], 0, true);

        // D: component = 16 m + 4 kg + e, column = c16 -> f_local = WAVE_F * wave + 64 cg + 4 c16 + t
        float* out = pieces + (slot * g.bf + member * BLOCK_F + wave * WAVE_F) * KP;
#pragma unroll
        for (int cg = 0; cg < NCG; ++cg) {
            if (cg >= 2 && !half2) break;                  // second half outside F: nothing to write
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 d[M16];
#pragma unroll
                for (int m = 0; m < M16; ++m) d[m] = acc[m][4 * cg + tt];
                sg_flush_tile16<KT>(flush_tr[wave], d, out + (int64_t)(64 * cg + tt) * KP, 4 * KP, lane);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// The x3w sweep with TWO waves per SIMD ("x3v", round 4): 8 waves per workgroup, a wave owns ONE 64-column group x all KP components
// (M16A x 4 accumulator tiles: 112 registers at K = 105), 256 registers per wave.  Why: on X with full significands one wave per SIMD
// executes the split of a tile (47 vector instructions), its 42 MFMAs and the zero-plane branch strictly one after the other -- the
// matrix pipe idles 40 % of the time (profiles/r04/cfg4_x3_share8_fullsig_*), and neither the compiler's interleave nor a pinned one
// (x3p, removed) hides vector work in the MFMA gaps of a SINGLE wave.  Two waves do it by themselves: while one wave splits, the other
// wave's MFMAs issue (tools/inflight_bw.hip, TWO=1: the sweep's skeleton 3.64 -> 5.14 TB/s).  Same workgroup tile (512 columns), same
// stream-K division and pieces as stream_gemm_x3w_kernel<KT, 1, M16A>; per accumulator the same products in the same order (stages
// ascending; hi x {hi, mid, lo}, mid x {hi, mid}, lo x hi), so the pieces are bitwise those of x3w on data where the zero-plane
// decision -- here one per 64-column group and stage, there one per 16-column tile -- adds nothing but exact zeros (non-negative X).
// Structure of a stage as in x3w2 below: the four column tiles are split up front, which frees the X registers -- the group's 8 loads
// are re-issued for the next stage BEFORE the stage's MFMAs -- then component tile by component tile: 3 ds_read_b128 (one tile ahead),
// 12 or 24 MFMAs.  The panel is staged as in x3w (float32 master -> registers -> split -> LDS, one 8-row set per thread and stage).
template <int KT, int NG = 1, int M16A = 2 * KT, bool ONEPLANE = false>
__global__ __launch_bounds__(512, 1)
void stream_gemm_x3v_kernel(const float* __restrict__ S, const float* __restrict__ Pf, float* __restrict__ pieces,
                            int64_t ldS, SweepGeom g, int* __restrict__ xcc_out)
{
    sg_report_xcc(xcc_out);
    static_assert(M16A >= 1 && M16A <= 2 * KT, "active 16-component tiles");
    static_assert(M16A * NG <= 8, "128 accumulator registers per lane");
    constexpr int KP = 32 * KT, M16 = 2 * KT, KPA = 16 * M16A;       // KPA: panel components that are staged and multiplied
    constexpr int WAVE_F = 64 * NG, NW = 8, BLOCK_F = NW * WAVE_F;   // NG 64-column groups per wave (K <= 64: 2 -> 1024-column workgroup tiles)
    constexpr int NT = 64 * NW;
    constexpr int ROWS = 32;
    static_assert(SG_ROW_ALIGN % ROWS == 0, "stream-K spans are multiples of one stage");
    constexpr int STAGE_BF16 = ROWS * KP;
    constexpr int SETS = (ROWS / 8) * KP;
    static_assert(SETS <= NT && 2 * SETS >= NT, "one 8-row set of the panel per thread and stage");
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][3 * STAGE_BF16];
    __shared__ __attribute__((aligned(16))) float flush_tr[NW][16 * (KP + 4)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kg = lane >> 4;
    SgWalk walk;
    int team, member;
    sg_team_of_block(g, blockIdx.x, team, member);
    sg_walk_init(walk, g, team);

    // this thread's set of a panel stage: rows 8 p_rb .. + 7 of component p_col.  EVERY thread loads a set (those past the last set
    // re-load an earlier one and do not store it): a load under a condition would make the compiler wait for ALL outstanding loads
    // at the join -- the X split would then wait for the panel loads issued just before it, a full memory latency per stage
    const int p_set = tid < SETS ? tid : tid - SETS;
    const int p_rb = p_set / KP, p_col = p_set % KP;
    const bool p_mine = tid < SETS && p_col < KPA;        // (the padding components are loaded -- zeros -- but neither stored nor read)
    float pf[8];
    f32x4 x[8];                                           // ONE 64-column group x 32 rows in flight per wave (two waves per SIMD: 64 KiB per CU)

    int ft, r_begin, r_end;
    int64_t slot;
    while (sg_walk_next(walk, g, ft, r_begin, r_end, slot)) {
        const int nst = (r_end - r_begin) / ROWS;
        const int wt = ft * g.gw + member;                 // this workgroup's BLOCK_F-wide tile (ft = the team's tile)
        if ((int64_t)wt * BLOCK_F >= g.F) continue;        // a member past the last column of a partly filled team tile (block-uniform)
        const int f0 = wt * BLOCK_F + wave * WAVE_F;       // (F is a multiple of 128: a wave's columns are all inside or all outside)
        const bool active = f0 < g.F;

        const float* pfptr = Pf + (int64_t)(r_begin + 8 * p_rb) * KP + p_col;
        auto load_p = [&](int t) {
            const float* src = pfptr + (int64_t)t * ROWS * KP;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = src[e * KP];
        };
        auto store_p = [&](int b) {
            if (p_mine) {
                u32x4 o[3];
                x3_split8_scalar(pf, o);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(&lds[b][q * STAGE_BF16 + 8 * p_set]) = o[q];
            }
        };

        __syncthreads();
        load_p(0);
        store_p(0);
        if (nst > 1) load_p(1);

        if (!active) {
            // a wave whose columns lie past F: the panel staging and EVERY barrier of the loop below, nothing else
            __syncthreads();
            for (int t = 0; t + 1 < nst; ++t) {
                store_p((t + 1) & 1);
                if (t + 2 < nst) load_p(t + 2);
                __syncthreads();
            }
            continue;
        }

        f32x4 acc[M16A][4 * NG];                            // (only the tiles with real components; the others are flushed as zeros)
#pragma unroll
        for (int m = 0; m < M16A; ++m)
#pragma unroll
            for (int j = 0; j < 4 * NG; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // this lane's float4 of row r_begin + 8 kg (+ e), columns f0 + 64 group + 4 c16 .. + 3 (element j -> column tile j, column c16):
        // a wave-uniform row address (scalar registers, scalar arithmetic) + ONE 32-bit lane offset -- the saddr form of global_load;
        // per-lane 64-bit addresses of 8 rows x NG groups, which the compiler hoists out of the stage loop, cost 16+ registers of the 256
        const char* xrow0 = reinterpret_cast<const char*>(S + (int64_t)r_begin * ldS + f0);
        const unsigned x_lane = (unsigned)((int64_t)(8 * kg) * ldS + 4 * c16) * 4u;          // < 2^32: 24 rows of at most 2^25 floats
        auto load_x = [&](int t, int cg) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = x3_load_nt_saddr(xrow0 + (((int64_t)t * ROWS + e) * ldS + 64 * cg) * 4, x_lane);
        };
        const int lds_lane = (kg * KP + c16) * 8;

        __builtin_amdgcn_sched_barrier(0);
        load_x(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        // one stage, group by group: split the group's four column tiles (which frees the X registers), re-issue the 8 loads for the
        // NEXT group-stage, then component tile by component tile: 3 ds_read_b128 (one tile ahead), 12 or 24 MFMAs
        auto stage = [&](const unsigned short* __restrict__ lrow, int t_next, bool last) {
#pragma unroll
            for (int cg = 0; cg < NG; ++cg) {
                u32x4 b[4][3];
                unsigned rest = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = x[e][j];
                    if constexpr (ONEPLANE) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) b[j][0][q] = x3_cvt2(v[2 * q], v[2 * q + 1]);       // exact: every value is one bf16 plane
                        b[j][1] = b[j][2] = u32x4{0u, 0u, 0u, 0u};
                    } else {
                        x3_split8_scalar(v, b[j]);
                        rest |= b[j][1][0] | b[j][1][1] | b[j][1][2] | b[j][1][3];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (cg + 1 < NG) load_x(t_next - 1, cg + 1);
                else if (!last) load_x(t_next, 0);
                __builtin_amdgcn_sched_barrier(0);
                auto lda = [&](int m, u32x4 (&a)[3]) {
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp) a[pp] = *reinterpret_cast<const u32x4*>(lrow + pp * STAGE_BF16 + (16 * m) * 8);
                };
                // (the fragments of the next component tile are read one tile ahead where the registers allow it: 128 accumulators leave
                // no room for the second set -- the other wave of the SIMD covers the LDS latency there)
                constexpr bool AHEAD = M16A * NG < 8;
                u32x4 a[AHEAD ? 2 : 1][3];
                lda(0, a[0]);
                // (an if-THEN per component tile, not two copies of the loop: see stream_gemm_x3w2_kernel)
                const bool full = !ONEPLANE && __builtin_amdgcn_ballot_w64((rest & 0x7fff7fffu) != 0u) != 0ull;
#pragma unroll
                for (int m = 0; m < M16A; ++m) {
                    asm volatile("" ::: "memory");
                    if (AHEAD && m + 1 < M16A) lda(m + 1, a[AHEAD ? (m + 1) & 1 : 0]);
                    if (!AHEAD && m > 0) lda(m, a[0]);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[m][4 * cg + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[AHEAD ? m & 1 : 0][pp]), __builtin_bit_cast(bf16x8, b[j][0]), acc[m][4 * cg + j], 0, 0, 0);
                    if (full) {
#pragma unroll
                        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                acc[m][4 * cg + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[AHEAD ? m & 1 : 0][pp]), __builtin_bit_cast(bf16x8, b[j][1]), acc[m][4 * cg + j], 0, 0, 0);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[m][4 * cg + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[AHEAD ? m & 1 : 0][0]), __builtin_bit_cast(bf16x8, b[j][2]), acc[m][4 * cg + j], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };

        // (t_next is the stage whose FIRST group the last group of this stage re-issues; the other groups of a stage re-issue within
        // stage t_next - 1 = this stage)
        int t = 0;
        for (; t + 2 < nst; ++t) {
            store_p((t + 1) & 1);
            load_p(t + 2);
            __builtin_amdgcn_sched_barrier(0);
            stage(&lds[t & 1][lds_lane], t + 1, false);
            __syncthreads();
        }
        if (t + 1 < nst) {
            store_p((t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            stage(&lds[t & 1][lds_lane], t + 1, false);
            __syncthreads();
            ++t;
        }
        stage(&lds[t & 1][lds_lane], t + 1, true);

        // D: component = 16 m + 4 kg + e, column = c16 -> f_local = WAVE_F * wave + 64 group + 4 c16 + tile
        float* out = pieces + (slot * g.bf + member * BLOCK_F + wave * WAVE_F) * KP;
#pragma unroll
        for (int cg = 0; cg < NG; ++cg)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 d[M16];
#pragma unroll
                for (int m = 0; m < M16; ++m) d[m] = m < M16A ? acc[m < M16A ? m : 0][4 * cg + tt] : f32x4{0.f, 0.f, 0.f, 0.f};
                sg_flush_tile16<KT>(flush_tr[wave], d, out + (int64_t)(64 * cg + tt) * KP, 4 * KP, lane);
            }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// The two-wave sweep on X STORED as exact bf16 planes (x_dtype = "split": NPX = 1 or 2 planes, k-packed [row / 8][F][8], kernels_bf16.hpp) for
// 64 < K <= 128 ("bf16v", round 4).  stream_gemm_bf16_kernel multiplies 32 x 32 x 16 tiles -- all 128 padded components of a K = 105 model, 256 or
// 128 accumulator registers per wave -- and is matrix-pipe-bound there (1.36 ms per sweep of cfg4's share for 5 GB of X: 0.46 of the roof).  This is
// stream_gemm_x3v_kernel with the X path replaced: a lane's 16-byte granule of the k-packed planes (8 consecutive rows of one column) IS the B
// operand of v_mfma_f32_16x16x32_bf16 -- no conversion, no split, no zero-plane test; tile j of a 64-column group is columns 16 j .. 16 j + 15
// (contiguous, not interleaved as in the float32 layout); two stages of X in flight per wave (16 or 32 registers each).  Products per accumulator:
// p0 x0, p1 x0, p2 x0 (NPX = 2: then p0 x1, p1 x1): the terms with xp + pp <= 2, as in stream_gemm_bf16_kernel; float32-grade, not bitwise that
// kernel's (another matrix instruction).  Panel staging, stream-K division, teams, pieces: x3v's.
template <int KT, int M16A, int NPX>
__global__ __launch_bounds__(512, 1)
void stream_gemm_bf16v_kernel(const unsigned short* __restrict__ S, int64_t x_plane, const float* __restrict__ Pf, float* __restrict__ pieces,
                              SweepGeom g, int* __restrict__ xcc_out)
{
    sg_report_xcc(xcc_out);
    static_assert(M16A >= 1 && M16A <= 2 * KT && M16A <= 8, "active 16-component tiles");
    static_assert(NPX == 1 || NPX == 2, "planes of X");
    constexpr int KP = 32 * KT, M16 = 2 * KT, KPA = 16 * M16A;
    constexpr int WAVE_F = 64, NW = 8, BLOCK_F = NW * WAVE_F;
    constexpr int NT = 64 * NW;
    constexpr int ROWS = 32;
    static_assert(SG_ROW_ALIGN % (2 * ROWS) == 0, "stream-K spans hold an even number of stages");
    constexpr int STAGE_BF16 = ROWS * KP;
    constexpr int SETS = (ROWS / 8) * KP;
    static_assert(SETS <= NT, "one 8-row set of the panel per thread and stage");
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][3 * STAGE_BF16];
    __shared__ __attribute__((aligned(16))) float flush_tr[NW][16 * (KP + 4)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kg = lane >> 4;
    SgWalk walk;
    int team, member;
    sg_team_of_block(g, blockIdx.x, team, member);
    sg_walk_init(walk, g, team);

    const int p_set = tid % SETS;                         // (every thread loads a set: see stream_gemm_x3v_kernel)
    const int p_rb = p_set / KP, p_col = p_set % KP;
    const bool p_mine = tid < SETS && p_col < KPA;
    float pf[8];
    u32x4 x[2][NPX][4];                                   // ring slot = stage parity: [slot][plane][16-column tile]

    int ft, r_begin, r_end;
    int64_t slot;
    while (sg_walk_next(walk, g, ft, r_begin, r_end, slot)) {
        const int nst = (r_end - r_begin) / ROWS;          // even (SG_ROW_ALIGN = 64 rows)
        const int wt = ft * g.gw + member;
        if ((int64_t)wt * BLOCK_F >= g.F) continue;
        const int f0 = wt * BLOCK_F + wave * WAVE_F;
        const bool active = f0 < g.F;

        const float* pfptr = Pf + (int64_t)(r_begin + 8 * p_rb) * KP + p_col;
        auto load_p = [&](int t) {
            const float* src = pfptr + (int64_t)t * ROWS * KP;
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[e] = src[e * KP];
        };
        auto store_p = [&](int b) {
            if (p_mine) {
                u32x4 o[3];
                x3_split8_scalar(pf, o);
#pragma unroll
                for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4*>(&lds[b][q * STAGE_BF16 + 8 * p_set]) = o[q];
            }
        };

        __syncthreads();
        load_p(0);
        store_p(0);
        load_p(1);                                         // (nst >= 2)

        if (!active) {
            __syncthreads();
            for (int t = 0; t + 1 < nst; ++t) {
                store_p((t + 1) & 1);
                if (t + 2 < nst) load_p(t + 2);
                __syncthreads();
            }
            continue;
        }

        f32x4 acc[M16A][4];
#pragma unroll
        for (int m = 0; m < M16A; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // granule (row block r_begin / 8 + 4 t + kg, column f0 + 16 j + c16) of plane xp: wave-uniform base + one 32-bit lane offset
        // (bytes: 16 per granule; the lane's part (kg F + c16) 16 < 2^32 for F <= 2^26 -- the host keeps longer axes on the 4-wave kernel)
        const char* xbase = reinterpret_cast<const char*>(S) + ((int64_t)(r_begin / 8) * g.F + f0) * 16;
        const unsigned x_lane = (unsigned)(((int64_t)kg * g.F + c16) * 16);
        auto load_x = [&](int t, auto slot_c) {
            constexpr int XS = decltype(slot_c)::value;
#pragma unroll
            for (int xp = 0; xp < NPX; ++xp)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    x[XS][xp][j] = sg_load_nt_saddr<u32x4>(xbase + ((int64_t)(4 * t) * g.F + 16 * j) * 16 + xp * x_plane * 2, x_lane);
        };
        using Slot0 = std::integral_constant<int, 0>;
        using Slot1 = std::integral_constant<int, 1>;
        const int lds_lane = (kg * KP + c16) * 8;

        __builtin_amdgcn_sched_barrier(0);
        load_x(0, Slot0{});
        load_x(1, Slot1{});
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        // one stage: the slot's granules are the B operands; component tile by component tile 3 ds_read_b128 (one tile ahead) and 12 (NPX = 2: 20)
        // MFMAs; the slot is re-issued for stage t + 2 right after its last use
        auto stage = [&](auto slot_c, const unsigned short* __restrict__ lrow, int t_next, bool last) {
            constexpr int XS = decltype(slot_c)::value;
            auto lda = [&](int m, u32x4 (&a)[3]) {
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) a[pp] = *reinterpret_cast<const u32x4*>(lrow + pp * STAGE_BF16 + (16 * m) * 8);
            };
            u32x4 a[2][3];
            lda(0, a[0]);
#pragma unroll
            for (int m = 0; m < M16A; ++m) {
                asm volatile("" ::: "memory");
                if (m + 1 < M16A) lda(m + 1, a[(m + 1) & 1]);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                    for (int xp = 0; xp < NPX; ++xp) {
                        if (xp + pp > 2) continue;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m & 1][pp]), __builtin_bit_cast(bf16x8, x[XS][xp][j]), acc[m][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!last) load_x(t_next, slot_c);
            __builtin_amdgcn_sched_barrier(0);
        };

        int t = 0;
        for (; t + 2 < nst; t += 2) {
            store_p(1);
            load_p(t + 2);
            __builtin_amdgcn_sched_barrier(0);
            stage(Slot0{}, &lds[0][lds_lane], t + 2, false);
            __syncthreads();
            store_p(0);
            load_p(t + 3);
            __builtin_amdgcn_sched_barrier(0);
            stage(Slot1{}, &lds[1][lds_lane], t + 3, false);
            __syncthreads();
        }
        store_p(1);
        __builtin_amdgcn_sched_barrier(0);
        stage(Slot0{}, &lds[0][lds_lane], 0, true);
        __syncthreads();
        stage(Slot1{}, &lds[1][lds_lane], 0, true);

        // D: component = 16 m + 4 kg + e, column = c16 of tile j -> f_local = WAVE_F * wave + 16 j + c16 (contiguous tiles: row stride KP)
        float* out = pieces + (slot * g.bf + member * BLOCK_F + wave * WAVE_F) * KP;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 d[M16];
#pragma unroll
            for (int m = 0; m < M16; ++m) d[m] = m < M16A ? acc[m < M16A ? m : 0][j] : f32x4{0.f, 0.f, 0.f, 0.f};
            sg_flush_tile16<KT>(flush_tr[wave], d, out + (int64_t)(16 * j) * KP, KP, lane);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// One pass over X for 128 < K <= 256 ("x3w2", round 4).  The blocked two-half path of kernels_wide.hpp ran every sweep TWICE, once per
// half of the components (X read four times per iteration: 0.32 of the HBM roof at K = 150).  Here a wave owns 64 columns x ALL 256
// components -- 16 x 4 accumulator tiles of v_mfma_f32_16x16x32_bf16 = 256 registers -- and a workgroup 256 columns, so X is read
// once per sweep.  What changes against stream_gemm_x3w_kernel:
//   * a workgroup tile is only 256 columns wide, so the panel is as many bytes per stage as X itself: splitting it into planes in
//     every workgroup (176 vector instructions per thread and stage, as many as the X split) made a first version no faster than two
//     passes.  The panel is therefore split ONCE per sweep (pack_panel3_wide_kernel: the blocked factor [2][rows][128] -> three exact
//     bf16 planes in the k-packed layout [plane][row / 8][256][8], which IS the LDS image: a stage is three contiguous 16 KB chunks)
//     and a stage is staged by LDS-DMA (global_load_lds_dwordx4), issued at the start of the previous stage: no registers in flight,
//     no ds_write, no vector arithmetic, a whole stage of MFMAs to land in;
//   * the A fragments of 16 component tiles x 3 planes do not fit in registers beside the X ring, so the stage is walked component
//     tile by component tile: 3 ds_read_b128 (one tile ahead), then that tile's products for ALL FOUR column tiles (12 or 24 MFMAs)
//     -- 48 LDS reads per stage and wave, a quarter of the LDS array's rate;
//   * for that the four column tiles of the stage are split up front, and the zero-plane decision is taken once per stage and wave
//     (hi-only only if all four tiles are one-plane); ONEPLANE (census of alpine_finalize_X): no split, no test;
//   * up front also frees the X registers at the START of a stage: the stage's 8 loads are re-issued for two stages ahead before its
//     MFMAs run (ring of two stages, 16 KiB per wave in flight for almost two stages of compute);
//   * the pieces keep the blocked layout the consumers of the two-half path read: components [0, 128) go to pieces0, [128, 256) to
//     pieces1, both [.][128] with the same geometry -- reduce_pieces_kernel, wide_h_apply_kernel, wide_num_kernel are unchanged.
// Per accumulator the same products in the same order as x3w (stages ascending; hi x {hi, mid, lo}, mid x {hi, mid}, lo x hi).
// M16A = 16-component tiles that hold real components (ceil(K / 16), 9..16; the others are never multiplied and flushed as zeros).

// the blocked factor [2][rows_pad][128] -> three exact bf16 planes of its first kpa components (kpa = 16 x the component tiles the
// sweep multiplies: the padding columns of a K = 150 model are a third of the panel's bytes), k-packed: dst[q][rb][comp][8] (rb = row / 8),
// plane stride in u32x4
__global__ __launch_bounds__(256)
void pack_panel3_wide_kernel(const float* __restrict__ P0, const float* __restrict__ P1, int rows, int kpa, u32x4* __restrict__ dst, int64_t plane_stride)
{
    const int64_t n = (int64_t)(rows / 8) * kpa;               // granules: (rb, comp)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t rb = i / kpa;
        const int comp = (int)(i - rb * kpa);
        const float* src = (comp < 128 ? P0 : P1) + rb * 8 * 128 + (comp & 127);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = src[e * 128];
        u32x4 o[3];
        x3_split8_scalar(v, o);
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[q * plane_stride + i] = o[q];
    }
}

// NW = 8 (round 4, second half): TWO waves per SIMD -- 8 waves x 64 columns = 512-column workgroup tiles (half the panel bytes per byte of X), ONE
// stage of X in flight per wave (re-issued right after the split), 256 registers per wave: 16 M16A accumulators + 32 + planes + fragments, which
// fits for the one-plane form up to 10 component tiles (K <= 160).  Same pieces; the division is over 512-column tiles.
template <int M16A, bool ONEPLANE = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1)
void stream_gemm_x3w2_kernel(const float* __restrict__ S, const u32x4* __restrict__ Pk, int64_t plane_stride,
                             float* __restrict__ pieces0, float* __restrict__ pieces1, int64_t ldS, SweepGeom g, int* __restrict__ xcc_out)
{
    sg_report_xcc(xcc_out);
    static_assert(M16A >= 9 && M16A <= 16, "active 16-component tiles of a wide model");
    constexpr int KH = 128;
    constexpr int KP = 16 * M16A;                                     // panel components staged and multiplied (the pieces stay 2 x 128 wide)
    static_assert(NW == 4 || (NW == 8 && ONEPLANE && M16A <= 10), "two waves per SIMD: 256 registers per wave");
    constexpr int WAVE_F = 64, BLOCK_F = NW * WAVE_F;
    constexpr int RING = NW == 4 ? 2 : 1;                             // stages of X in flight per wave
    constexpr int ROWS = 32;                                          // rows per k-step = per panel stage
    static_assert(SG_ROW_ALIGN % ROWS == 0, "stream-K spans are whole stages");
    constexpr int STAGE_BF16 = ROWS * KP;                             // bf16 elements of one plane of a stage (16 KB)
    constexpr int STAGE_V4 = STAGE_BF16 / 8;                          // ... in 16-byte granules (64 per component tile = 1 KiB)
    constexpr int PIECES = 3 * M16A;                                  // 1-KiB pieces of a stage image [plane][M16A KiB]
    __shared__ __attribute__((aligned(16))) unsigned short lds[2][3 * STAGE_BF16];          // 6 KB per component tile and buffer: 96 KB at M16A = 16
    __shared__ __attribute__((aligned(16))) float flush_tr[NW][16 * (KH + 4)];             // 33 KB (NW = 4)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, kg = lane >> 4;
    SgWalk walk;
    int team, member;
    sg_team_of_block(g, blockIdx.x, team, member);
    sg_walk_init(walk, g, team);

    f32x4 x[RING][8];

    int ft, r_begin, r_end;
    int64_t slot;
    while (sg_walk_next(walk, g, ft, r_begin, r_end, slot)) {
        const int nst = (r_end - r_begin) / ROWS;
        const int wt = ft * g.gw + member;                 // this workgroup's BLOCK_F-wide tile (ft = the team's tile)
        if ((int64_t)wt * BLOCK_F >= g.F) continue;        // a member past the last column of a partly filled team tile (block-uniform)
        const int f0 = (wt * NW + wave) * WAVE_F;
        const bool active = f0 < g.F;

        // Panel stage t -> LDS buffer b by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = one contiguous KiB of the LDS image per
        // wave-instruction, no vector registers, no ds_write): the stage image is 3 x M16A pieces of 1 KiB ([plane][M16A KiB]); wave w copies the
        // pieces 4 i + w.  hipcc does not see these loads (inline assembly: M0 = the LDS address, saved and restored around the
        // instruction), so THIS code waits for them -- dma_wait before the barrier that publishes the buffer; they retire in order with the X
        // loads, whose waits the compiler still counts itself (a hidden older load only makes such a wait cover more, never less).
        // (source = a wave-uniform KiB of the packed panel in scalar registers + the lane's 16 bytes: the saddr form -- the per-lane 64-bit
        // addresses of up to 12 pieces, hoisted out of the stage loop, were spilled at M16A >= 14 and re-loaded behind an s_waitcnt vmcnt(0))
        const u32x4* pbase = Pk + (int64_t)(r_begin / 8) * KP;
        const unsigned dma_lane = (unsigned)lane * 16u;
        auto dma_stage = [&](int t, int b) {
            const unsigned lbase = (unsigned)(size_t)(&lds[b][0]);
#pragma unroll
            for (int i = 0; i < (PIECES + NW - 1) / NW; ++i) {
                const int piece = NW * i + wave;                    // wave-uniform
                if (piece >= PIECES) break;
                const int q = piece / M16A, kk = piece % M16A;      // plane, KiB within the plane's M16A KiB
                const unsigned long long src = x3_uniform_u64(reinterpret_cast<unsigned long long>(pbase + q * plane_stride + (int64_t)t * STAGE_V4 + kk * 64));
                const unsigned dst = lbase + (unsigned)(q * 2 * STAGE_BF16 + kk * 1024);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(dma_lane), "s"(src), "s"(dst) : "memory");
            }
        };
        // all but the `younger` youngest vector-memory operations of this wave have completed (the panel DMAs are older than an X reload issued after them)
        auto dma_wait = [&](bool x_reload_after) {
#if defined(X3W2_ABLATE) && (X3W2_ABLATE & 1)
            return;            // (tools/x3w2_bench.hip, timing only, wrong results: is the sweep waiting for its panel?)
#endif
            if (x_reload_after) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };

        __syncthreads();
        dma_stage(0, 0);

        if (!active) {
            // a wave whose 64 columns lie past F: the panel staging and EVERY barrier of the loop below, nothing else (kept apart from
            // the computing path: accumulators that are updated under a condition make the compiler shuffle them between register classes)
            dma_wait(false);
            __syncthreads();
            for (int t = 0; t < nst; ++t) {
                if (t + 1 < nst) dma_stage(t + 1, (t + 1) & 1);
                dma_wait(false);
                __syncthreads();
            }
            continue;
        }

        f32x4 acc[M16A][4];                                 // (only the tiles with real components; the others are flushed as zeros)
#pragma unroll
        for (int m = 0; m < M16A; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        // this lane's float4 of row r_begin + 8 kg (+ e), columns f0 + 4 c16 .. + 3 (element t -> column tile t, column c16): wave-uniform row
        // address + one 32-bit lane offset (x3_load_nt_saddr)
        const char* xrow0 = reinterpret_cast<const char*>(S + (int64_t)r_begin * ldS + f0);
        const unsigned x_lane = (unsigned)((int64_t)(8 * kg) * ldS + 4 * c16) * 4u;          // < 2^32: 24 rows of at most 2^25 floats
        const int lds_lane = (kg * KP + c16) * 8;

        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < RING; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e)         // (a one-stage span re-reads stage 0 into the second slot: valid memory, never used)
                x[p][e] = x3_load_nt_saddr(xrow0 + ((int64_t)(p < nst ? p : 0) * ROWS + e) * ldS * 4, x_lane);
        __builtin_amdgcn_sched_barrier(0);
        dma_wait(false);                 // (stage 0's panel; also drains the X prologue once per span)
        __syncthreads();

        // stage t on ring slot P_ (= t & 1 = its panel buffer): split the four column tiles, start the NEXT stage's panel on its way into
        // the other buffer, re-issue the slot for stage t + 2, then component tile by component tile
        auto stage = [&](auto slot_c, int t) {
            constexpr int P_ = decltype(slot_c)::value;
            const unsigned short* __restrict__ lrow = &lds[P_][lds_lane];
            const bool more = t + 1 < nst;
            u32x4 b[4][3];
            unsigned rest = 0u;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x[P_ % RING][e][j];
                if constexpr (ONEPLANE) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) b[j][0][q] = x3_cvt2(v[2 * q], v[2 * q + 1]);       // exact: every value is one bf16 plane
                    b[j][1] = b[j][2] = u32x4{0u, 0u, 0u, 0u};
                } else {
                    x3_split8_scalar(v, b[j]);
                    rest |= b[j][1][0] | b[j][1][1] | b[j][1][2] | b[j][1][3];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#if defined(X3W2_ABLATE) && (X3W2_ABLATE & 4)
            if (more && t == 0) dma_stage(t + 1, P_ ^ 1);      // (timing only, wrong results: no panel traffic after the first stages)
#else
            if (more) dma_stage(t + 1, P_ ^ 1);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (t + RING < nst) {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[P_ % RING][e] = x3_load_nt_saddr(xrow0 + ((int64_t)(t + RING) * ROWS + e) * ldS * 4, x_lane);
            }
            __builtin_amdgcn_sched_barrier(0);
            auto lda = [&](int m, u32x4 (&a)[3]) {
#pragma unroll
                for (int pp = 0; pp < 3; ++pp) a[pp] = *reinterpret_cast<const u32x4*>(lrow + pp * STAGE_BF16 + (16 * m) * 8);
            };
#ifdef X3W2_AHEAD
            constexpr int AD = X3W2_AHEAD;                  // (tools/x3w2_bench.hip: A/B of the prefetch depth)
#else
            constexpr int AD = 1;                           // component tiles the fragment reads run ahead of the MFMAs
#endif
            u32x4 a[AD + 1][3];
#pragma unroll
            for (int m = 0; m < AD; ++m) lda(m, a[m]);
            // (an if-THEN per component tile, not two copies of the loop under an if / else: with the accumulators updated on two
            // different paths the register allocator no longer keeps each in ONE accumulator register across the join and shuttles them
            // through vector registers -- 1 040 v_accvgpr moves and 356 B of scratch per lane in that form, none in this one)
            const bool full = !ONEPLANE && __builtin_amdgcn_ballot_w64((rest & 0x7fff7fffu) != 0u) != 0ull;
#pragma unroll
            for (int m = 0; m < M16A; ++m) {
                asm volatile("" ::: "memory");              // (no motion of the fragment reads across component tiles before scheduling either)
                if (m + AD < M16A) lda(m + AD, a[(m + AD) % (AD + 1)]);
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pp = 0; pp < 3; ++pp)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m % (AD + 1)][pp]), __builtin_bit_cast(bf16x8, b[j][0]), acc[m][j], 0, 0, 0);
                if (full) {
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m % (AD + 1)][pp]), __builtin_bit_cast(bf16x8, b[j][1]), acc[m][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m % (AD + 1)][0]), __builtin_bit_cast(bf16x8, b[j][2]), acc[m][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            dma_wait(t + RING < nst);                       // the next stage's panel has landed (this wave's share) before the barrier publishes it
        };
        using Slot0 = std::integral_constant<int, 0>;
        using Slot1 = std::integral_constant<int, 1>;

#if defined(X3W2_ABLATE) && (X3W2_ABLATE & 2)
#define X3W2_STAGE_BARRIER() do { } while (0)      // (timing only, wrong results: what do the per-stage barriers cost?)
#else
#define X3W2_STAGE_BARRIER() __syncthreads()
#endif
        int t = 0;
        for (; t + 1 < nst; t += 2) {
            stage(Slot0{}, t);
            X3W2_STAGE_BARRIER();
            stage(Slot1{}, t + 1);
            X3W2_STAGE_BARRIER();
        }
        if (t < nst) { stage(Slot0{}, t); __syncthreads(); }

        // D: component = 16 m + 4 kg + e, column = c16 -> f_local = WAVE_F * wave + 4 c16 + tile; components [0, 128) -> pieces0, the rest -> pieces1
        const int64_t off = (slot * g.bf + member * BLOCK_F + wave * WAVE_F) * KH;        // (BLOCK_F = NW x 64 columns)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            float* out = (hh == 0 ? pieces0 : pieces1) + off;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 d[8];
#pragma unroll
                for (int m = 0; m < 8; ++m) d[m] = (8 * hh + m < M16A) ? acc[(8 * hh + m < M16A) ? 8 * hh + m : 0][tt] : f32x4{0.f, 0.f, 0.f, 0.f};
                sg_flush_tile16<4>(flush_tr[wave], d, out + (int64_t)tt * KH, 4 * KH, lane);
            }
        }
    }
}

}  // namespace alpine
