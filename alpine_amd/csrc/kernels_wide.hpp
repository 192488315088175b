// Wide models: 128 < K <= 1024 (the reference accepts any K, main.py:331-336).
//
// Every kernel of kernels.hpp is built around K x K operands that live whole in LDS (64 KB at K = 128) and around sweep tiles
// whose accumulators fill the register file at 128 components.  A wide model keeps both: its factors are stored as NH = ceil(K / 128)
// column blocks ("halves": the path was built for NH = 2 and generalised to NH <= 8 in round 4) of KH = 128 components each, in a blocked layout
//     W : [NH][Gp][KH]      H : [NH][Np][KH]      K x K matrices : [NH][NH][KH][KH]  (block (a, b) = rows of half a, columns of half b)
// so that each half is exactly the [rows][128] array the KT = 4 kernels work on:
//   * the two streaming sweeps run once per panel half (the same stream_gemm_* instantiations, X is read NH times per sweep; 128 < K <= 224
//     on the x3 sweeps: ONE pass, stream_gemm_x3w2_kernel), and their pieces are reduced per half by the same consumers;
//   * the Gram matrices are NH (NH + 1) / 2 KT = 4 blocks (gram_cross_kernel: A_a^T A_b) and their transposes;
//   * the updates split into  den = A . M  (wide_den_kernel: the MFMA core of w_update_mfma_kernel, the 128 x 128 block (i, o)
//     of M staged through LDS for each pair of halves) and an elementwise apply (wide_w_apply_kernel, wide_h_apply_kernel with
//     the guided terms of main.py:636-650; all guided components must sit in the first half: sum k_i <= 128).
// The block-coordinate branch (k_lo, k_hi, block_orth, only_cov) and mini-batch views (rows_pad = the view's padded cells) use the same
// kernels.  No fused tails, no MFMA form of the guided terms: a wide iteration is ~25 launches and reads X four times instead of twice.
// It exists so that K up to 256 RUNS with the reference's results, not to be fast.
#pragma once
#include "kernels.hpp"

namespace alpine {

constexpr int WIDE_KH = 128;      // components per half
constexpr int WIDE_KT = 4;        // 32-column tiles per half
constexpr int WIDE_MAX_NH = 8;    // halves a model may have (K <= 1024)

// part[blk][k][k'] = sum_{r in rows of block blk} A[r][k] * A2[r][k']   (A, A2: R x KH row-major; A2 == A gives the plain Gram)
// TA / TB: 32-component tiles of A / A2 that hold real components (the others are zero columns: neither loaded nor multiplied, their
// part of the output is written as zeros).  Compile-time: the run-time predicated form of round 3 spilled 32 registers (132 B of scratch
// per lane); a model needs three of the seven instantiations -- (4, 4), (4, t), (t, t) with t = the second half's tiles.
template <int KT, int TA, int TB>
__global__ __launch_bounds__(256, 1)
void gram_cross_kernel(const float* __restrict__ A, const float* __restrict__ A2, float* __restrict__ part, int R, int rows_per_wave)
{
    static_assert(TA >= 1 && TA <= KT && TB >= 1 && TB <= KT, "active 32-component tiles");
    constexpr int KP = 32 * KT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int gw = blockIdx.x * 4 + wave;
    const int r0 = gw * rows_per_wave;
    const int r1 = min(R, r0 + rows_per_wave);         // R and rows_per_wave are multiples of 16

    f32x16 acc[TA][TB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < TB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    for (int r = r0; r < r1; r += 4) {
        float v[2][TA], v2[2][TB];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int m = 0; m < TA; ++m) v[p][m] = A[(int64_t)(r + 2 * p + h) * KP + 32 * m + c];
#pragma unroll
            for (int m = 0; m < TB; ++m) v2[p][m] = A2[(int64_t)(r + 2 * p + h) * KP + 32 * m + c];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[p][a], v2[p][b], acc[a][b], 0, 0, 0);
    }
    // the block's 4 waves are summed in wave order through LDS -> one partial per block (fixed order)
    __shared__ float gl[KP * KP];
    if (TA < KT || TB < KT) {
        for (int idx = threadIdx.x; idx < KP * KP; idx += 256) gl[idx] = 0.f;
        __syncthreads();
    }
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int a = 0; a < TA; ++a)
#pragma unroll
                for (int b = 0; b < TB; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                        float* q = &gl[row * KP + 32 * b + c];
                        *q = (wv == 0 ? 0.f : *q) + acc[a][b][e];
                    }
        }
        __syncthreads();
    }
    float* out = part + (int64_t)blockIdx.x * KP * KP;
    for (int idx = threadIdx.x; idx < KP * KP; idx += 256) out[idx] = gl[idx];
}

// den[o][r][k] = sum over the halves i of  A[i][r][:] . M(i, o)[:][k]      for the 128 rows of a block, every output half o.
//   mode 0 (W update, main.py:599-603):  M = 2 HH^T + orth * (coupled off-diagonal) + l2 * I   restricted to the K real components
//   mode 1 (H update / transform, main.py:654):  M = 2 W^T W
// G = the K x K source (HH^T or W^T W) in blocked layout; the block (i, o) of M is formed while it is staged into LDS.
// The MFMA core is w_update_mfma_kernel's: the contraction index is visited in the order a lane holds it in the C/D layout, so the
// rows of A are B operands straight from the registers tile_load_cd leaves them in, and den comes out in the same layout.
struct WideDenArgs {
    int rows_pad;           // padded rows of A / den (leading extent of a half)
    int K;                  // real components
    int mode;
    float orth, l2;
    int k_lo, k_hi, block_orth;      // block-coordinate branch: orthogonality couples the group's own components only
    int nh;                 // halves of the blocked layout
};

__global__ __launch_bounds__(256, 1)
void wide_den_kernel(const float* __restrict__ A, const float* __restrict__ G, float* __restrict__ den, WideDenArgs a)
{
    constexpr int KT = WIDE_KT, KP = WIDE_KH, LD = KP + 4, TRSZ = 32 * LD;
    extern __shared__ float Ml[];                   // [k'][k] block of M, then the waves' tiles
    float* trall = Ml + KP * KP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    float* tr = trall + wave * TRSZ;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 32;              // rows_pad is a multiple of 128: the rows exist
    for (int o = 0; o < a.nh; ++o) {
        if (o * KP >= a.K) break;                                            // (never: every half holds real components)
        f32x16 acc[KT];
#pragma unroll
        for (int mo = 0; mo < KT; ++mo)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mo][e] = 0.f;
        for (int i = 0; i < a.nh; ++i) {
            __syncthreads();                                                 // the previous block of M has been consumed
            const float* Gb = G + (int64_t)(i * a.nh + o) * KP * KP;
#pragma unroll 4
            for (int idx4 = tid; idx4 < KP * KP / 4; idx4 += 256) {             // float4 per thread and trip: 16 trips, four loads in flight
                const int kp = i * KP + idx4 / (KP / 4), k0 = o * KP + 4 * (idx4 % (KP / 4));     // global component indices (contraction, output)
                const f32x4 gsrc = reinterpret_cast<const f32x4*>(Gb)[idx4];
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + e;
                    float x = 0.f;
                    if (kp < a.K && k < a.K) {
                        x = 2.f * gsrc[e];
                        if (a.mode == 0) {
                            const bool coupled = !a.block_orth || (kp >= a.k_lo && kp < a.k_hi);
                            x += (kp == k) ? a.l2 : (coupled ? a.orth : 0.f);
                        }
                    }
                    v[e] = x;
                }
                reinterpret_cast<f32x4*>(Ml)[idx4] = v;
            }
            __syncthreads();
            f32x4 wreg[KT][4];
            tile_load_cd<KT>(A + ((int64_t)i * a.rows_pad + r0) * KP, tr, lane, wreg);
            // 32-component tiles that are all padding (K = 150: three of the second half's four) are skipped on both sides: their rows of
            // M are zero (contraction side) and their columns of den are never read (output side; they stay zero)
#pragma unroll
            for (int m = 0; m < KT; ++m) {
                if (i * KP + 32 * m >= a.K) continue;                                            // block-uniform
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* mrow = Ml + (32 * m + 8 * q + 4 * h + e) * KP + c;      // A operand [i = k][kk = k'] = M[k'][k]
#pragma unroll
                        for (int mo = 0; mo < KT; ++mo)
                            if (o * KP + 32 * mo < a.K)                                          // block-uniform
                                acc[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(mrow[32 * mo], wreg[m][q][e], acc[mo], 0, 0, 0);
                    }
            }
        }
        f32x4 outv[KT][4];
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) outv[m][q][e] = acc[m][4 * q + e];
        tile_store_cd<KT>(den + ((int64_t)o * a.rows_pad + r0) * KP, tr, lane, outv, 32);
    }
}

// W[g][k] *= (2 XH^T[g][k]) / max(den[g][k] + l1, eps)   for the real genes and the components [k_lo, k_hi)   (main.py:596-605)
// plus dotpart[block] = sum over the block's elements of XH^T * W_old in float64 (trace-form loss).  Blocked arrays [nh][Gp][KH].
__global__ __launch_bounds__(256)
void wide_w_apply_kernel(float* __restrict__ W, const float* __restrict__ XHt, const float* __restrict__ den, double* __restrict__ dotpart,
                         int G, int64_t Gp, int K, float l1, float eps, int do_update, int k_lo, int k_hi, int nh)
{
    __shared__ double red[256];
    const int64_t n4 = (int64_t)nh * Gp * (WIDE_KH / 4);
    double dacc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int half = (int)(i / (Gp * (WIDE_KH / 4)));
        const int64_t j = i - (int64_t)half * Gp * (WIDE_KH / 4);
        const int64_t g = j / (WIDE_KH / 4);
        const int k0 = half * WIDE_KH + 4 * (int)(j % (WIDE_KH / 4));
        f32x4 w = reinterpret_cast<const f32x4*>(W)[i];
        const f32x4 x = reinterpret_cast<const f32x4*>(XHt)[i];
        if (g < G) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dacc += (double)x[e] * (double)w[e];
        }
        if (do_update && g < G) {
            const f32x4 d = reinterpret_cast<const f32x4*>(den)[i];
            bool touched = false;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                if (k < K && k >= k_lo && k < k_hi) { w[e] = w[e] * fast_div(2.f * x[e], fmaxf(d[e] + l1, eps)); touched = true; }
            }
            if (touched) reinterpret_cast<f32x4*>(W)[i] = w;
        }
    }
    red[threadIdx.x] = dacc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) dotpart[blockIdx.x] = red[0];
}

// One wave per cell, two halves per pass: in pass hp lane l holds components half * 128 + 4 * (l & 31) .. + 3 of half = hp + (l >> 5).
//   num = 2 * (sum of the W^TX pieces of the cell's tile, per half) + guided_num,  den = (2 W^TW H)[cell] + guided_den
//   H[cell][k] *= num / max(den, eps)   for k in [k_lo, k_hi)                                                 (main.py:631-656)
// The pieces of half h start at pieces + h * piece_stride.  Guided components all sit in half 0 (alpine_create checks sum k_i <= 128):
// only the first pass has guided terms.  transform (main.py:705-709): no guided terms (only_cov = n_cov), numerator from `num_in`.
template <int LOSS>
__global__ __launch_bounds__(256)
void wide_h_apply_kernel(float* __restrict__ H, const float* __restrict__ den, const float* __restrict__ pieces_all, int64_t piece_stride, int nh,
                         SweepGeom g, const float* __restrict__ num_in, const float* __restrict__ Y, const float* __restrict__ B, CovMeta meta,
                         int N, int64_t Np, int K, float eps, int k_lo, int k_hi, int only_cov)
{
    constexpr int KH = WIDE_KH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t cell = (int64_t)blockIdx.x * 4 + wave; cell < N; cell += (int64_t)gridDim.x * 4) {
        for (int hp = 0; hp < nh; hp += 2) {
            const int half = hp + (lane >> 5), k0 = half * KH + 4 * (lane & 31);
            const bool live = half < nh;                                   // (an odd number of halves: the upper lanes idle in the last pass)
            const int64_t off4 = ((int64_t)(live ? half : 0) * Np + cell) * KH + 4 * (lane & 31);
            const f32x4 hv = live ? *reinterpret_cast<const f32x4*>(H + off4) : f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 dv = live ? *reinterpret_cast<const f32x4*>(den + off4) : f32x4{1.f, 1.f, 1.f, 1.f};
            f32x4 nv = f32x4{0.f, 0.f, 0.f, 0.f};
            if (live) {
                if (num_in != nullptr) {
                    nv = *reinterpret_cast<const f32x4*>(num_in + off4);
                } else {
                    const float* pieces = pieces_all + (int64_t)half * piece_stride;
                    const int ft = (int)(cell / g.bf), fl = (int)(cell % g.bf);
                    int w_lo, w_hi;
                    sg_tile_pieces(g, ft, w_lo, w_hi);
                    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                    for (int w = w_lo; w <= w_hi; ++w) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(pieces + sg_piece_offset(g, w, ft, KH) + (int64_t)fl * KH + 4 * (lane & 31));
                        a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
                    }
                    nv = f32x4{2.f * (float)a0, 2.f * (float)a1, 2.f * (float)a2, 2.f * (float)a3};
                }
            }
            for (int i = 0; i < (hp == 0 ? meta.n_cov : 0); ++i) {         // (wave-uniform: the DPP sums below need every lane)
                if (only_cov >= 0 && i != only_cov) continue;
                const int off = meta.off[i], ki = meta.k[i], Ci = meta.lev[i], bo = meta.boff[i], yo = meta.yoff[i];
                const float lam = (LOSS == 0) ? meta.lam[i] : meta.lam2[i];
                for (int cl = 0; cl < Ci; ++cl) {
                    float coef[4];
                    float part = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kk = k0 + e - off;
                        coef[e] = (live && kk >= 0 && kk < ki) ? B[bo + cl * ki + kk] : 0.f;
                        part = fmaf(coef[e], hv[e], part);
                    }
                    const float bh = wave_sum_f32_dpp(part);
                    const float y = Y[(int64_t)(yo + cl) * Np + cell];
                    const float z = (LOSS == 0) ? y / fmaxf(bh, eps) : y;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lb = lam * coef[e];
                        nv[e] = fmaf(lb, z, nv[e]);
                        dv[e] = (LOSS == 0) ? dv[e] + lb : fmaf(lb, bh, dv[e]);
                    }
                }
            }
            if (live) {
                f32x4 out = hv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + e;
                    if (k < K && k >= k_lo && k < k_hi) out[e] = hv[e] * fast_div(nv[e], fmaxf(dv[e], eps));
                }
                *reinterpret_cast<f32x4*>(H + off4) = out;
            }
        }
    }
}

// num[half][cell][k] = 2 * sum of the W^TX pieces of the cell's tile (transform: the numerator is loop-invariant, main.py:706)
__global__ __launch_bounds__(256)
void wide_num_kernel(float* __restrict__ num, const float* __restrict__ pieces_all, int64_t piece_stride, int nh, SweepGeom g, int N, int64_t Np)
{
    constexpr int KH = WIDE_KH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t cell = (int64_t)blockIdx.x * 4 + wave; cell < N; cell += (int64_t)gridDim.x * 4) {
        for (int hp = 0; hp < nh; hp += 2) {
            const int half = hp + (lane >> 5);
            if (half >= nh) continue;
            const float* pieces = pieces_all + (int64_t)half * piece_stride;
            const int ft = (int)(cell / g.bf), fl = (int)(cell % g.bf);
            int w_lo, w_hi;
            sg_tile_pieces(g, ft, w_lo, w_hi);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            for (int w = w_lo; w <= w_hi; ++w) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(pieces + sg_piece_offset(g, w, ft, KH) + (int64_t)fl * KH + 4 * (lane & 31));
                a0 += v[0]; a1 += v[1]; a2 += v[2]; a3 += v[3];
            }
            *reinterpret_cast<f32x4*>(num + ((int64_t)half * Np + cell) * KH + 4 * (lane & 31)) =
                f32x4{2.f * (float)a0, 2.f * (float)a1, 2.f * (float)a2, 2.f * (float)a3};
        }
    }
}

// direct-form ||X - W H||^2 in float64 for blocked factors: thread = element of the thread axis (gene, or cell when the roles are
// swapped), its row of the first factor in registers one half at a time, rows of the second factor broadcast from LDS.
__global__ __launch_bounds__(256)
void eval_recon_wide_kernel(const float* __restrict__ X, int64_t ldX, const float* __restrict__ Fa, int64_t rowsA_pad,
                            const float* __restrict__ Fb, int64_t rowsB_pad, int A, int Bn, int per_block, double* __restrict__ part, int nh)
{
    constexpr int KH = WIDE_KH;
    __shared__ float hl[EV_CELLS][KH];
    __shared__ double red[256];
    const int t = threadIdx.x;
    const int a = blockIdx.x * 256 + t;
    const int n0 = blockIdx.y * per_block, n1 = min(Bn, n0 + per_block);
    double acc = 0.0;
    for (int nb = n0; nb < n1; nb += EV_CELLS) {
        const int lim = min(EV_CELLS, n1 - nb);
        float p[EV_CELLS];
#pragma unroll
        for (int r = 0; r < EV_CELLS; ++r) p[r] = 0.f;
        for (int half = 0; half < nh; ++half) {
            float w[KH];
#pragma unroll
            for (int k = 0; k < KH; ++k) w[k] = (a < A) ? Fa[((int64_t)half * rowsA_pad + a) * KH + k] : 0.f;
            __syncthreads();
            for (int idx = t; idx < EV_CELLS * KH; idx += 256) {
                const int r = idx / KH, k = idx % KH;
                hl[r][k] = (nb + r < n1) ? Fb[((int64_t)half * rowsB_pad + nb + r) * KH + k] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < EV_CELLS; ++r) {
                float s = p[r];
#pragma unroll
                for (int k = 0; k < KH; ++k) s = fmaf(w[k], hl[r][k], s);
                p[r] = s;
            }
        }
        if (a < A) {
#pragma unroll
            for (int r = 0; r < EV_CELLS; ++r) {
                if (r < lim) {
                    const double d = (double)X[(int64_t)(nb + r) * ldX + a] - (double)p[r];
                    acc += d * d;
                }
            }
        }
    }
    red[t] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0];
}

}  // namespace alpine
