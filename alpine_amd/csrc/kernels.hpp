// Device kernels of libalpine_hip.so (gfx950 / CDNA4 only).
//
// One MU iteration of ALPINE (alpine/main.py:589-663, loss :726-753) is evaluated as
//
//   phase 1 (old H, old B; everything here is a SUM OVER CELLS -> the multi-GPU reduce block)
//     hstats_kernel      per-cell covariate terms: B-update numerators/denominators (:617-626),
//                        prediction-loss sums (:727-731, :745-748)
//     gram_kernel        HH^T  (the K x K factor of ((2W)H)H^T, :599, re-associated)
//     stream_gemm_kernel XH^T  (:596) over the cells x genes copy of X      <- HOT, MFMA f32
//   phase 2
//     w_update_mfma_kernel    W <- W * 2XH^T / max(W(2HH^T + orth + l2 I) + l1, eps)  (:596-605, :474-484)
//                        + float64 partials of <XH^T, W_old> for the trace-form loss
//     loss_finalize      ||X||^2 - 2<XH^T,W> + <W^TW,HH^T>  (== :736), total (:750-752)
//     b_update_kernel    (:615-628)
//     gram_kernel        W^T W (the K x K factor of (2W^T)(WH), :654, re-associated)
//     stream_gemm_kernel W^T X (:653) over the genes x cells copy of X      <- HOT, MFMA f32
//     h_update_mfma_kernel    H <- H * (guided_num + 2W^TX) / max(guided_den + 2W^TW H, eps) (:631-656)
//
// Layouts (all float32, zero padded): X_gn [Gp][Np], X_ng [Np][Gp] (Gp, Np multiples of 128);
// W [Gp][KP] gene-major, H [Np][KP] cell-major (KP = K rounded up to 32); Y [sum C_i][Np];
// partial slabs [split][rows][KP].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alpine {

// In-kernel time stamps (s_memrealtime, 10 ns ticks) for tools/: compiled only with -DALPINE_STAMPS (a throwaway build,
// python alpine_amd/build.py --stamps -> libalpine_hip_stamps.so); the product library carries none.
#ifdef ALPINE_STAMPS
__device__ unsigned long long g_sweep_stamps[8 * 2048];     // per sweep workgroup: start, first stage, last stage done, end, flush ticks, segments, XCC id
__device__ unsigned long long g_hu_stamps[16 * 8192];       // per H-update block: phase boundaries
__device__ unsigned int g_sweep_hist[4 + 4096];             // [0] = launches so far; then per sweep launch: XCC id of workgroup 0 | workgroup 1 << 8 | grid << 16
#define SG_STAMP_SET(i, v) do { if (threadIdx.x == 0) alpine::g_sweep_stamps[8 * (blockIdx.x & 2047) + (i)] = (v); } while (0)
#define SG_NOW() __builtin_amdgcn_s_memrealtime()
#define HU_STAMP(i) do { if (threadIdx.x == 0) alpine::g_hu_stamps[16 * (blockIdx.x & 8191) + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SG_STAMP_SET(i, v) do {} while (0)
#define SG_NOW() 0ull
#define HU_STAMP(i) do {} while (0)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int MAX_COV = 16;       // covariates per model
constexpr int MAX_COV_K = 64;     // guided components PER covariate (the statistics kernels size their LDS for it); their sum may reach K

struct CovMeta {
    int n_cov;
    int loss_type;                // 0 KL, 1 Frobenius
    int off[MAX_COV];             // first column of the covariate's block in W / H
    int k[MAX_COV];               // k_i
    int lev[MAX_COV];             // C_i
    int boff[MAX_COV];            // offset of B_i (C_i x k_i row-major) in the packed B buffer
    int yoff[MAX_COV];            // first row of Y_i in the packed Y buffer
    int soff[MAX_COV];            // offset of this covariate's statistics in the stats vector
    float lam[MAX_COV];           // float32(lambda_i)
    float lam2[MAX_COV];          // float32(2*lambda_i)
};
// statistics of covariate i: [bnum: C_i*k_i][bden: k_i][loss_hi][loss_lo]

// ----------------------------------------------------------------------------------------------
// stream_gemm: out[f][k] = sum_r S[r][f] * P[r][k]      S: R x ldS streamed once, P: R x KP panel
//
//   sweep XH^T : S = X_ng (cells x genes), P = H (cells x KP)  -> out[gene][k]
//   sweep W^TX : S = X_gn (genes x cells), P = W (genes x KP)  -> out[cell][k]
//
// The contraction index r is the ROW index of both operands, so S is read with 16 B/lane loads
// that are contiguous along f (1 KiB per wave-instruction = 2 rows x 512 B) straight into the MFMA
// B operand -- no LDS round trip for the 16 GB stream -- and the small panel is staged through LDS
// (double buffered, 16 rows per stage) and shared by the 4 waves of the workgroup.
// v_mfma_f32_32x32x2_f32: A[i=l&31][kk=l>>5] = P[r+kk][32m+i], B[kk=l>>5][j=l&31] = S[r+kk][f(j)],
// where the four B tiles of a wave interleave along f (tile t column j is f0 + 4j + t), so that one
// float4 load feeds four MFMAs.  Work item = (512-wide f tile, split of the r range); partial results
// go to per-workgroup pieces (stream-K, see SweepGeom) and are summed in a fixed order downstream.
constexpr int SG_THREADS = 256;
constexpr int SG_WAVES = 4;
constexpr int SG_WAVE_F = 128;
constexpr int SG_BLOCK_F = SG_WAVES * SG_WAVE_F;
constexpr int SG_ROW_ALIGN = 64;     // split boundaries are multiples of this (>= rows per panel stage of every variant)
constexpr int SG_MAX_CHAIN = 16384;  // longest run of contraction rows one float32 accumulator may cover (see SweepGeom::sub)

// One panel stage = SG_RING k-steps.  Software pipeline per k-step p (all indices static):
//   ds_read  A operands of step p+1          (LDS latency hidden behind the 8..16 MFMAs of step p)
//   8*KT/2.. MFMAs of step p on x[p]
//   global_load x[p] <- the same k-step of the NEXT stage (prefetch distance = SG_RING k-steps, constant)
// so the instruction stream is uniformly MFMA-dense and exactly SG_RING (+ panel) loads are always in flight.

template <int KT, int SG_RING, bool LAST>
__device__ __forceinline__ void sg_stage(f32x16 (&acc)[KT][4], f32x4 (&x)[SG_RING], const float* __restrict__ lrow,
                                         const float* __restrict__ xnext, int64_t ldS)
{
    constexpr int KP = 32 * KT;
    float a[2][KT];
#pragma unroll
    for (int m = 0; m < KT; ++m) a[0][m] = lrow[32 * m];
#pragma unroll
    for (int p = 0; p < SG_RING; ++p) {
        if (p + 1 < SG_RING) {
#pragma unroll
            for (int m = 0; m < KT; ++m) a[(p + 1) & 1][m] = lrow[(2 * (p + 1)) * KP + 32 * m];
        }
        __builtin_amdgcn_sched_barrier(0);       // LDS read of step p+1 issues BEFORE the MFMAs of step p
#pragma unroll
        for (int m = 0; m < KT; ++m) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p & 1][m], x[p][j], acc[m][j], 0, 0, 0);
        }
        if (!LAST) x[p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xnext + (int64_t)(2 * p) * ldS));
        __builtin_amdgcn_sched_barrier(0);       // keep one load per k-step where it is (see kernel comment)
    }
}

// Work division ("stream-K"): the (f tile, row) space of nft * R rows is cut into equal contiguous spans of L rows,
// one per workgroup of a fixed grid (2 x #CU at K <= 64), so every workgroup executes the same number of MFMAs and
// there is no ragged last round.  A span may cross tile boundaries; the workgroup writes one partial piece per tile
// it touches: piece(w, j) = pieces[((w * maxp + j) * bf + f_local) * KP + k], j = tile - first tile of w.  Consumers
// sum a tile's pieces in ascending w (sg_tile_pieces below): fixed order, no atomics, bitwise reproducible.
struct SweepGeom {
    int F;        // free extent, multiple of 128 (leading dimension of S)
    int R;        // contraction rows, multiple of SG_ROW_ALIGN
    int bf;       // f columns per workgroup tile: 128 per wave (512 with 4 waves; 1024 for the 8-wave bf16 sweeps)
    int nft;      // bf-wide f tiles
    int L;        // rows of (tile,row) space per workgroup, multiple of SG_ROW_ALIGN
    int nwg;      // spans (= piece owners); the launch grid is ceil(nwg / sub) workgroups
    int maxp;     // pieces per span
    int sub;      // consecutive spans per workgroup: a workgroup restarts its accumulators at every span boundary, so no
                  // float32 accumulator chain covers more than L <= SG_MAX_CHAIN contraction rows whatever the shard size
    int panel_fixed;  // timing-only ablation of the bf16 sweeps: every panel stage re-reads stage 0 (cache-resident) -> wrong results
    int dL;       // rows moved from every span of an odd workgroup to every span of an even one (multiple of SG_ROW_ALIGN, 0 = equal
                  // spans): workgroups are dispatched round-robin over the 8 XCDs and the XCDs do not stream this access pattern
                  // at the same rate; the division is static, so results never depend on where a workgroup ran
    int gw;       // TEAM width (x3 sweeps; 1 everywhere else).  The unit of the division above is then a team of gw workgroups that
                  // walk the SAME spans of contraction rows side by side, workgroup j of the team on columns [j, j + 1) * bf / gw of
                  // the bf-wide tile -- bf = gw x (columns of one workgroup) -- and the gw workgroups of a team have equal
                  // blockIdx % 8, i.e. sit on ONE XCD under the round-robin placement (sg_team_of_block): the panel rows a team
                  // needs are fetched into that XCD's L2 once instead of gw times.  At K = 105 the panel re-reads were 27 % of the
                  // sweep's fabric traffic with an L2 hit rate of 1.4 % (profiles/r04/cfg4_x3_share8_*): every workgroup was at a
                  // different row of the panel at any time.  Placement only decides the speed, never the result: "workgroup" in
                  // every comment of this division reads "team", and a piece is still bf columns x KP of one tile.
};

// blockIdx -> (team, member).  Blocks b and b + 8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md), so the gw members of a
// team are gw consecutive blocks OF ONE XCD: b = 8 i + x  ->  team = 8 (i / gw) + x, member = i % gw.  The team's parity is the
// XCD's parity, which is what SweepGeom::dL keys on.  A bijection of [0, 8 m gw) onto [0, 8 m) x [0, gw) (sg_grid pads the teams to
// a multiple of 8; teams past the last span find nothing to do).
__host__ __device__ inline void sg_team_of_block(const SweepGeom& g, int b, int& team, int& member)
{
    if (g.gw <= 1) { team = b; member = 0; return; }
    const int x = b & 7, i = b >> 3;
    team = (i / g.gw) * 8 + x;
    member = i % g.gw;
}
__host__ __device__ inline int sg_grid(const SweepGeom& g)
{
    const int teams = (g.nwg + g.sub - 1) / g.sub;
    return g.gw <= 1 ? teams : (teams + 7) / 8 * 8 * g.gw;
}

// span v (= piece owner) belongs to workgroup v / sub; its length is L + dL (even workgroup) or L - dL (odd workgroup); the
// spans of workgroups 2j and 2j + 1 together cover 2 * sub * L rows of the (tile, row) space
__host__ __device__ inline int sg_span_len(const SweepGeom& g, int v) { return ((v / g.sub) & 1) ? g.L - g.dL : g.L + g.dL; }
__host__ __device__ inline int64_t sg_span_start(const SweepGeom& g, int v)
{
    const int w = v / g.sub, s = v - w * g.sub;
    return (int64_t)(w >> 1) * 2 * g.sub * g.L + (w & 1) * (int64_t)g.sub * (g.L + g.dL) + (int64_t)s * ((w & 1) ? g.L - g.dL : g.L + g.dL);
}
__host__ __device__ inline int sg_span_of_row(const SweepGeom& g, int64_t row)
{
    const int64_t pair_rows = 2 * (int64_t)g.sub * g.L, even_rows = (int64_t)g.sub * (g.L + g.dL);
    const int64_t pb = row / pair_rows;
    const int64_t rem = row - pb * pair_rows;
    if (rem < even_rows) return (int)(2 * pb * g.sub + rem / (g.L + g.dL));
    return (int)((2 * pb + 1) * g.sub + (rem - even_rows) / (g.L - g.dL));
}

inline int64_t sg_round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
inline int64_t sg_max64(int64_t a, int64_t b) { return a > b ? a : b; }
inline int64_t sg_min64(int64_t a, int64_t b) { return a < b ? a : b; }
// Stream-K geometry of a sweep (host side): a fixed grid of `slots` workgroups (as many as the
// chip holds at once), each an equal share of the (tile,row) space.  forced > 0 asks for about `forced` workgroups per
// tile instead (tests use it to exercise shares that do / do not cross tiles).  A workgroup's share is cut into `sub`
// equal spans of L <= SG_MAX_CHAIN rows: it restarts its float32 accumulators at every span boundary and writes a piece
// per span and tile, so the length of an accumulator chain -- and with it the rounding error of a sweep, which has the
// same sign every iteration because X does not change -- is bounded independently of the shard size (the spans' pieces
// are summed in float64 by the consumers).  cfg3's shares are 15 4xx-15 6xx rows: sub = 1, nothing changes there.
// gw > 1: `bf` is the width of a TEAM's tile (gw x the columns of one workgroup) and `slots` the number of teams.
inline SweepGeom sg_make_geom(int64_t F, int64_t R, int slots, int forced, int bf, int bias_pm, int gw = 1)
{
    SweepGeom g{};
    g.F = (int)F; g.R = (int)R; g.bf = bf; g.gw = gw;
    g.nft = (int)((F + bf - 1) / bf);
    const int64_t total = (int64_t)g.nft * R;
    int64_t want = forced > 0 ? (int64_t)g.nft * forced : slots;
    want = sg_max64(1, sg_min64(want, total / SG_ROW_ALIGN));
    const int64_t share = sg_round_up((total + want - 1) / want, SG_ROW_ALIGN);         // rows per workgroup
    g.sub = (int)((share + SG_MAX_CHAIN - 1) / SG_MAX_CHAIN);
    g.L = (int)sg_round_up((share + g.sub - 1) / g.sub, SG_ROW_ALIGN);
    // uneven division between even and odd workgroups (SweepGeom::dL): as much of the asked bias as keeps the LONGER span
    // within the accumulation cap (never at the price of more spans) and the shorter one above 3/4 of the mean
    int64_t dL = (int64_t)g.L * bias_pm / 1000 / SG_ROW_ALIGN * SG_ROW_ALIGN;
    const int64_t dmax = sg_min64((SG_MAX_CHAIN - g.L) / SG_ROW_ALIGN * SG_ROW_ALIGN, g.L / 4 / SG_ROW_ALIGN * SG_ROW_ALIGN);
    dL = sg_max64(-dmax, sg_min64(dL, dmax));
    g.dL = (int)dL;
    int n_wg = (int)(total / ((int64_t)g.sub * g.L));                                   // first workgroup count whose shares cover `total`
    while (sg_span_start(g, n_wg * g.sub) < total) ++n_wg;
    while (n_wg > 1 && sg_span_start(g, (n_wg - 1) * g.sub) >= total) --n_wg;
    g.nwg = n_wg * g.sub;
    g.maxp = (int)((g.L + (dL < 0 ? -dL : dL) + R - 1) / R) + 1;
    return g;
}

// A workgroup's walk over its `sub` consecutive spans, cut at tile boundaries into segments (tile ft, rows [r_begin, r_end)
// of that tile); every segment owns one piece: slot = span * maxp + (ft - first tile of the span).  All wave-uniform.
struct SgWalk {
    int64_t pos, pos_end, span_end;
    int span, first_tile, len;
};
__host__ __device__ __forceinline__ void sg_walk_init(SgWalk& s, const SweepGeom& g, int wg)
{
    const int64_t total = (int64_t)g.nft * g.R;
    s.span = wg * g.sub;
    s.len = sg_span_len(g, s.span);
    s.pos = sg_span_start(g, s.span);
    s.pos_end = s.pos + (int64_t)g.sub * s.len;
    if (s.pos_end > total) s.pos_end = total;
    s.span_end = s.pos + s.len;
    if (s.span_end > s.pos_end) s.span_end = s.pos_end;
    s.first_tile = (int)(s.pos / g.R);
}
__host__ __device__ __forceinline__ bool sg_walk_next(SgWalk& s, const SweepGeom& g, int& ft, int& r_begin, int& r_end, int64_t& slot)
{
    if (s.pos >= s.pos_end) return false;
    if (s.pos == s.span_end) {                   // next span of this workgroup: fresh accumulators, fresh pieces
        ++s.span;
        s.span_end += s.len;
        if (s.span_end > s.pos_end) s.span_end = s.pos_end;
        s.first_tile = (int)(s.pos / g.R);
    }
    ft = (int)(s.pos / g.R);
    r_begin = (int)(s.pos - (int64_t)ft * g.R);
    const int64_t r_stop = r_begin + (s.span_end - s.pos);
    r_end = (int)(r_stop < (int64_t)g.R ? r_stop : (int64_t)g.R);
    s.pos += r_end - r_begin;
    slot = (int64_t)s.span * g.maxp + (ft - s.first_tile);
    return true;
}

// pieces that contribute to tile ft: workgroups w_lo..w_hi; piece index of w for this tile
__host__ __device__ __forceinline__ void sg_tile_pieces(const SweepGeom& g, int ft, int& w_lo, int& w_hi)
{
    w_lo = sg_span_of_row(g, (int64_t)ft * g.R);
    w_hi = sg_span_of_row(g, ((int64_t)ft + 1) * g.R - 1);
    if (w_hi > g.nwg - 1) w_hi = g.nwg - 1;
}
__host__ __device__ __forceinline__ int64_t sg_piece_offset(const SweepGeom& g, int w, int ft, int KP)
{
    const int first = (int)(sg_span_start(g, w) / g.R);
    return ((int64_t)w * g.maxp + (ft - first)) * g.bf * KP;
}

// Flush of one accumulator tile group (KT MFMA tiles = 32 f columns x KP components, C/D layout: lane (c, h) holds
// k = 32m + 8q + 4h + e of column c) to 32 rows of a piece with FULL-LINE stores.  Written straight from the C/D
// registers every store instruction scatters 64 x 16 B over 32 rows (32-byte fragments of 128-byte lines); those
// partial-line writes cost far more than their bytes (an iteration at cfg3 was 5 % faster with the flush ablated).
// So the tile goes through a wave-private LDS scratch (32 x (KP + 4) floats) and comes back row-major: one store
// instruction = 64 x 16 contiguous bytes per row group.  Row c of the tile lands at out_row0 + c * row_stride.
template <int KT>
__device__ __forceinline__ void sg_flush_tile(float* __restrict__ tr, const f32x16* const (&d)[KT], float* __restrict__ out_row0,
                                              int64_t row_stride, int lane)
{
    constexpr int KP = 32 * KT, LD = KP + 4, Q4 = KP / 4;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x16& t = *d[m];
            *reinterpret_cast<f32x4*>(&tr[c * LD + 32 * m + 8 * q + 4 * h]) = f32x4{t[4 * q + 0], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int i = 0; i < (32 * Q4) / 64; ++i) {
        const int idx = i * 64 + lane, r = idx / Q4, c4 = idx % Q4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tr[r * LD + 4 * c4]);
        *reinterpret_cast<f32x4*>(out_row0 + (int64_t)r * row_stride + 4 * c4) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SG_RING = X register ring depth in MFMA k-steps (= prefetch distance); SG_PASSES = ring passes per panel stage
// (one barrier per 2*SG_RING*SG_PASSES rows).
template <int KT, int SG_RING, int SG_PASSES>
__global__ __launch_bounds__(SG_THREADS, (KT <= 2 ? 2 : 1))
void stream_gemm_kernel(const float* __restrict__ S, const float* __restrict__ P, float* __restrict__ pieces,
                        int64_t ldS, SweepGeom g,
                        unsigned long long* __restrict__ clk = nullptr)   // diagnostics only (tools/sweep_bench): per-WG {cycles, 100 MHz ticks, start tick, XCC id}
{
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    constexpr int KP = 32 * KT;
    constexpr int SG_CH = 2 * SG_RING * SG_PASSES;                 // rows per panel stage
    static_assert(SG_ROW_ALIGN % SG_CH == 0, "stage rows must divide the span alignment");
    constexpr int PASS = 2 * SG_RING;                              // rows per ring pass
    constexpr int PV = SG_CH * KP / 4 / SG_THREADS;                // float4 per thread per panel stage
    static_assert(SG_CH * KP / 4 % SG_THREADS == 0, "panel stage must tile the workgroup exactly");
    __shared__ __attribute__((aligned(16))) float lds[2][SG_CH * KP];
    __shared__ __attribute__((aligned(16))) float flush_tr[SG_WAVES][32 * (KP + 4)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform -> scalar branches
    const int c = lane & 31, h = lane >> 5;
    SgWalk walk;
    sg_walk_init(walk, g, blockIdx.x);
    const int lds_lane = h * KP + c;

    f32x4 preg[PV];
    f32x4 x[SG_RING];

    int ft, r_begin, r_end;
    int64_t slot;
    while (sg_walk_next(walk, g, ft, r_begin, r_end, slot)) {
        const int nst = (r_end - r_begin) / SG_CH;                 // >= 1 (everything is a multiple of SG_ROW_ALIGN)
        const int f0 = (ft * SG_WAVES + wave) * SG_WAVE_F;
        const bool active = f0 < g.F;

        // panel staging through PV float4 registers per thread
        const float* pptr = P + (int64_t)r_begin * KP + 4 * tid;
        auto load_p = [&](int t) {
            const float* q = pptr + (int64_t)t * (SG_CH * KP);
#pragma unroll
            for (int v = 0; v < PV; ++v) preg[v] = *reinterpret_cast<const f32x4*>(q + 4 * SG_THREADS * v);
        };
        auto store_p = [&](int b) {
#pragma unroll
            for (int v = 0; v < PV; ++v) *reinterpret_cast<f32x4*>(&lds[b][4 * (tid + SG_THREADS * v)]) = preg[v];
        };

        __syncthreads();                      // previous segment's readers are done with both LDS buffers
        // prologue in the SAME issue order as the steady state (panel loads older than the X ring), so that the
        // vmcnt state at loop entry equals the state at the back-edge and the in-loop waits stay counted
        load_p(0);
        store_p(0);
        if (nst > 1) load_p(1);

        if (!active) {
            // a wave whose f range is past the matrix only helps staging the panel; same barrier sequence
            __syncthreads();
            for (int t = 0; t + 1 < nst; ++t) {
                store_p((t + 1) & 1);
                if (t + 2 < nst) load_p(t + 2);
                __syncthreads();
            }
            continue;
        }

        f32x16 acc[KT][4];
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[m][j][e] = 0.f;

        // X addressing: row base advances by whole ring passes; one per-lane offset
        const float* xrow = S + (int64_t)r_begin * ldS + (h * (int)ldS + f0 + 4 * c);   // this lane's element of row pair 0
        const int64_t x_pass = (int64_t)PASS * ldS;

        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < SG_RING; ++p)
            x[p] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xrow + (int64_t)(2 * p) * ldS));
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        // Steady state.  Loads in flight at any k-step: SG_RING X loads + PV panel loads, all unconditional, so the
        // compiler's vmcnt bookkeeping is exact (constant vmcnt(SG_RING + PV - 1)); sched_barrier(0) pins the issue
        // points -- left alone, the scheduler sinks the loads next to their first use to save registers, which
        // serialises the HBM latency.
        int t = 0;
        for (; t + 2 < nst; ++t) {
            const float* lb = &lds[t & 1][lds_lane];
            store_p((t + 1) & 1);             // panel t+1 (loaded during stage t-1); buffer last read in stage t-1
            load_p(t + 2);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < SG_PASSES; ++q)
                sg_stage<KT, SG_RING, false>(acc, x, lb + q * PASS * KP, xrow + (t * SG_PASSES + q + 1) * x_pass, ldS);
            __syncthreads();
        }
        if (t + 1 < nst) {                    // second-to-last stage: no panel left to prefetch
            const float* lb = &lds[t & 1][lds_lane];
            store_p((t + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < SG_PASSES; ++q)
                sg_stage<KT, SG_RING, false>(acc, x, lb + q * PASS * KP, xrow + (t * SG_PASSES + q + 1) * x_pass, ldS);
            __syncthreads();
            ++t;
        }
        {   // last stage: only its final pass has nothing left to prefetch
            const float* lb = &lds[t & 1][lds_lane];
#pragma unroll
            for (int q = 0; q < SG_PASSES - 1; ++q)
                sg_stage<KT, SG_RING, false>(acc, x, lb + q * PASS * KP, xrow + (t * SG_PASSES + q + 1) * x_pass, ldS);
            sg_stage<KT, SG_RING, true>(acc, x, lb + (SG_PASSES - 1) * PASS * KP, xrow, ldS);
        }

        // D row (k within tile m) = 8q + 4h + e, D column = lane & 31 = c -> f_local = 128*wave + 4c + j
        float* out = pieces + (slot * SG_BLOCK_F + wave * SG_WAVE_F) * KP;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x16* d[KT];
#pragma unroll
            for (int m = 0; m < KT; ++m) d[m] = &acc[m][j];
            sg_flush_tile<KT>(flush_tr[wave], d, out + (int64_t)j * KP, 4 * KP, lane);
        }
    }
    if (clk && tid == 0) {
        clk[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - clk_t0;
        clk[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - clk_r0;
        clk[4 * blockIdx.x + 2] = clk_r0;                                       // start, 100 MHz ticks
        clk[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID
    }
}

// Offset of the piece that workgroup w wrote for tile ft, for w_lo < w <= w_hi (w_lo = first workgroup that touches ft):
// such a workgroup's span STARTS inside ft, so ft is its piece 0 -- no division needed.
__host__ __device__ __forceinline__ int64_t sg_piece_offset_inner(const SweepGeom& g, int w, int KP)
{
    return (int64_t)w * g.maxp * g.bf * KP;
}

// out[f][k] = sum over the pieces of f's tile (ascending workgroup, float64 accumulation), f < rows.
// The piece loop is latency-bound when written one load per trip (a tile has ~13 pieces on the fixed stream-K grid,
// whatever the shard size): loads are issued nine, then eight pieces at a time, the adds stay in ascending order.
__device__ __forceinline__ void reduce_pieces_block(const float* __restrict__ pieces, float* __restrict__ out, int rows, int KP, const SweepGeom& g,
                                                    int block, int nblocks)
{
    const int kq = KP / 4;
    const int64_t n4 = (int64_t)rows * kq;
    for (int64_t i = (int64_t)block * 256 + threadIdx.x; i < n4; i += (int64_t)nblocks * 256) {
        const int f = (int)(i / kq), k4 = (int)(i % kq);
        const int ft = f / g.bf, fl = f % g.bf;
        int w_lo, w_hi;
        sg_tile_pieces(g, ft, w_lo, w_hi);
        const int64_t in_piece = (int64_t)fl * KP + 4 * k4;
        // the first piece together with the next eight (predicated), then eight at a time: a tile's ~13 pieces are two round trips
        // instead of four; the adds stay in ascending workgroup order (same sums as the sequential form)
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(pieces + sg_piece_offset(g, w_lo, ft, KP) + in_piece);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        for (int w = w_lo + 1; w == w_lo + 1 || w <= w_hi; w += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                v[u] = (w + u <= w_hi) ? *reinterpret_cast<const f32x4*>(pieces + sg_piece_offset_inner(g, w + u, KP) + in_piece) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (w == w_lo + 1) { a0 = v0[0]; a1 = v0[1]; a2 = v0[2]; a3 = v0[3]; }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (w + u <= w_hi) { a0 += v[u][0]; a1 += v[u][1]; a2 += v[u][2]; a3 += v[u][3]; }
        }
        f32x4 o = {(float)a0, (float)a1, (float)a2, (float)a3};
        *reinterpret_cast<f32x4*>(out + (int64_t)f * KP + 4 * k4) = o;
    }
}

__global__ __launch_bounds__(256)
void reduce_pieces_kernel(const float* __restrict__ pieces, float* __restrict__ out, int rows, int KP, SweepGeom g)
{
    reduce_pieces_block(pieces, out, rows, KP, g, blockIdx.x, gridDim.x);
}

// Placement report of a sweep launch (alpine_finalize_X's probe, see SweepGeom::dL): workgroup 0 writes the XCC id it runs on.
// (workgroup 1 reports too: the even/odd bias only makes sense when workgroups 0 and 1 sit on XCCs of different parity -- on a
// partitioned device whose workgroups all share one XCC it would only unbalance the grid)
__device__ __forceinline__ void sg_report_xcc(int* __restrict__ xcc_out)
{
    if (xcc_out != nullptr && blockIdx.x < 2 && threadIdx.x == 0)
        xcc_out[blockIdx.x] = (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 15u);        // HW_REG_XCC_ID
}

// ----------------------------------------------------------------------------------------------
// gram: part[b][k][k'] = sum_{r in rows of block b} A[r][k] * A[r][k']     (A: R x KP, row-major)
// Used for HH^T (A = H) and W^TW (A = W).  Each wave owns a contiguous row range; A and B MFMA
// operands are the same registers.
constexpr int GR_ROWS_PER_WAVE = 256;      // upper bound; small matrices use fewer rows per wave to fill the chip

template <int KT>
__device__ __forceinline__ void gram_block(const float* __restrict__ A, float* __restrict__ part, int R, int rows_per_wave, int block)
{
    constexpr int KP = 32 * KT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int gw = block * 4 + wave;
    const int r0 = gw * rows_per_wave;
    const int r1 = min(R, r0 + rows_per_wave);         // R and rows_per_wave are multiples of 16

    f32x16 acc[KT][KT];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    for (int r = r0; r < r1; r += 8) {
        float v[4][KT];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int m = 0; m < KT; ++m) v[p][m] = A[(int64_t)(r + 2 * p + h) * KP + 32 * m + c];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int b = 0; b < KT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[p][a], v[p][b], acc[a][b], 0, 0, 0);
    }
    // the block's 4 waves are summed in wave order through LDS -> one partial per block (fixed order)
    __shared__ float gl[KP * KP];
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int b = 0; b < KT; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = 32 * a + (e & 3) + 8 * (e >> 2) + 4 * h;
                        float* q = &gl[row * KP + 32 * b + c];
                        *q = (wv == 0 ? 0.f : *q) + acc[a][b][e];
                    }
        }
        __syncthreads();
    }
    float* out = part + (int64_t)block * KP * KP;
    for (int idx = threadIdx.x; idx < KP * KP; idx += 256) out[idx] = gl[idx];
}

template <int KT>
__global__ __launch_bounds__(256, (KT <= 2 ? 2 : 1))
void gram_kernel(const float* __restrict__ A, float* __restrict__ part, int R, int rows_per_wave)
{
    gram_block<KT>(A, part, R, rows_per_wave, blockIdx.x);
}

// out[j] = sum_s in[s][j], small n, many slabs: block = 64 outputs x 16 slab groups; groups are combined in group
// order (float64), so the sum order is fixed.
__global__ __launch_bounds__(1024)
void reduce_many_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int nslab)
{
    __shared__ double red[16][64];
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), sg = threadIdx.x >> 6;
    double a = 0.0;
    if (j < n) for (int s = sg; s < nslab; s += 16) a += (double)in[(int64_t)s * n + j];
    red[sg][threadIdx.x & 63] = a;
    __syncthreads();
    if (sg == 0 && j < n) {
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) t += red[u][threadIdx.x & 63];
        out[j] = (float)t;
    }
}

// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// Wave-wide sums on the DPP path (no LDS crossbar: __shfl_xor is a ds_bpermute with ~100 cycles of latency per step, six
// dependent steps per sum): quad_perm, row_shr:4, row_shr:8 inside each row of 16 lanes, then row_bcast:15 / row_bcast:31
// across rows; the total lands in lane 63 and is broadcast with v_readlane.  Fixed order -> reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov_f32(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum_f32_dpp(float v)
{
    v += dpp_mov_f32<0xb1, 0xf>(v);      // quad_perm [1,0,3,2]
    v += dpp_mov_f32<0x4e, 0xf>(v);      // quad_perm [2,3,0,1]
    v += dpp_mov_f32<0x114, 0xf>(v);     // row_shr:4
    v += dpp_mov_f32<0x118, 0xf>(v);     // row_shr:8
    v += dpp_mov_f32<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v += dpp_mov_f32<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double v)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_f64_dpp(double v)
{
    v += dpp_mov_f64<0xb1, 0xf>(v);
    v += dpp_mov_f64<0x4e, 0xf>(v);
    v += dpp_mov_f64<0x114, 0xf>(v);
    v += dpp_mov_f64<0x118, 0xf>(v);
    v += dpp_mov_f64<0x142, 0xa>(v);
    v += dpp_mov_f64<0x143, 0xc>(v);
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// num / den as num * v_rcp_f32(den) (<= 1 ulp reciprocal, so <= 2 ulp quotient) instead of the 11-instruction IEEE division
// sequence: the multiplicative updates divide every element of W and H once per iteration, and in the latency-bound update
// kernels (one wave per SIMD at shard sizes) the 32 divisions of a lane were 2.4 us of a 37 us block (in-kernel stamps).
// den is clamped from below by eps in every caller, so the reciprocal's range is harmless.
__device__ __forceinline__ float fast_div(float num, float den) { return num * __builtin_amdgcn_rcpf(den); }

__device__ __forceinline__ float lane_bcast(float v, int src)   // src must be a compile-time constant
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

// ----------------------------------------------------------------------------------------------
// hstats: per block of 128 cells, for every covariate i (old H_i, old B_i):
//   KL  (:617-623, :727-731): bnum[c][k] = sum_n (lam*(Y/max(BH,eps)))[c][n] * H[k][n],  bden[k] = sum_n lam*H[k][n]
//                             loss = sum (y*log(max(y/yhat,eps)) - y + yhat)
//   Fro (:625-626, :745-748): bnum[c][k] = sum_n Y[c][n]*H[k][n]  (the 2* and the B HH^T side come later)
//                             loss = sum (y - BH)^2
constexpr int HS_CELLS = 128;
constexpr int HS_CT = 16;

__host__ __device__ inline size_t hstats_group_bytes(int max_k, int max_ct)
{
    size_t b = sizeof(double) * HS_CELLS + sizeof(float) * ((size_t)(max_k + max_ct) * HS_CELLS + (size_t)max_ct * max_k);
    return (b + 15) & ~(size_t)15;
}

// One group = 128 threads = 128 cells; a block may hold several groups (`group` = index inside the block, `gblock` = the
// group's global index = row of `part`): every group has its own slice of the dynamic LDS, barriers are block-wide (all
// groups run the same trip counts).
// hrow(k) = H[this thread's cell][k]: a row of the cell-major H in global memory (stand-alone kernel) or of the updated tile
// that the H update still holds in LDS (fused tail).
template <typename HRow>
__device__ __forceinline__ void hstats_group(HRow hrow, const float* __restrict__ Y, const float* __restrict__ B,
                                             const CovMeta& meta, float* __restrict__ part, int N, int64_t Np, float eps, int nstat,
                                             int max_k, int max_ct, unsigned char* __restrict__ smem_base, int group, int64_t gblock)
{
    // dynamic LDS sized for THIS model (max_k = largest k_i, max_ct = min(HS_CT, largest C_i)): a few KB instead of the
    // 45 KB of worst-case static arrays, so that all blocks of a shard are resident at once and hide each other's latency
    const size_t group_bytes = hstats_group_bytes(max_k, max_ct);
    double* lred = reinterpret_cast<double*>(smem_base + (size_t)group * group_bytes);  // [HS_CELLS]
    float (*hbuf)[HS_CELLS] = reinterpret_cast<float (*)[HS_CELLS]>(lred + HS_CELLS);    // [max_k][HS_CELLS]
    float (*zbuf)[HS_CELLS] = hbuf + max_k;                                              // [max_ct][HS_CELLS]
    float* Blf = reinterpret_cast<float*>(zbuf + max_ct);                                // [max_ct][max_k]
    const int t = threadIdx.x & (HS_CELLS - 1), lane = t & 63, wave = t >> 6;
    const int64_t n = gblock * HS_CELLS + t;
    const bool valid = n < N;
    float* out = part + gblock * nstat;

    for (int i = 0; i < meta.n_cov; ++i) {
        const int ki = meta.k[i], Ci = meta.lev[i], off = meta.off[i];
        const float lam = meta.lam[i];
        float* so = out + meta.soff[i];
        __syncthreads();
        for (int k = 0; k < ki; ++k) hbuf[k][t] = valid ? hrow(off + k) : 0.f;
        double lacc = 0.0;
        for (int c0 = 0; c0 < Ci; c0 += HS_CT) {
            const int ct = min(HS_CT, Ci - c0);
            __syncthreads();
            for (int idx = t; idx < ct * ki; idx += HS_CELLS)
                Blf[(idx / ki) * max_k + idx % ki] = B[meta.boff[i] + (c0 + idx / ki) * ki + idx % ki];
            __syncthreads();
            for (int c = 0; c < ct; ++c) {
                float bh = 0.f;
                for (int k = 0; k < ki; ++k) bh = fmaf(Blf[c * max_k + k], hbuf[k][t], bh);
                const float y = valid ? Y[(int64_t)(meta.yoff[i] + c0 + c) * Np + n] : 0.f;
                float z;
                if (meta.loss_type == 0) {
                    const float yh = fmaxf(bh, eps);
                    z = lam * (y / yh);
                    if (valid) lacc += (double)(y * logf(fmaxf(y / yh, eps)) - y + yh);
                } else {
                    z = y;
                    const float d = y - bh;
                    if (valid) lacc += (double)(d * d);
                }
                zbuf[c][t] = valid ? z : 0.f;
            }
            __syncthreads();
            for (int p = wave; p < ct * ki; p += HS_CELLS / 64) {
                const int c = p / ki, k = p % ki;
                float v = zbuf[c][lane] * hbuf[k][lane];
                v = fmaf(zbuf[c][lane + 64], hbuf[k][lane + 64], v);
                v = wave_sum_f32(v);
                if (lane == 0) so[(c0 + c) * ki + k] = v;
            }
        }
        for (int k = wave; k < ki; k += HS_CELLS / 64) {
            float v = lam * hbuf[k][lane] + lam * hbuf[k][lane + 64];
            v = wave_sum_f32(v);
            if (lane == 0) so[Ci * ki + k] = v;
        }
        lred[t] = lacc;
        __syncthreads();
        for (int s = HS_CELLS / 2; s > 0; s >>= 1) {
            if (t < s) lred[t] += lred[t + s];
            __syncthreads();
        }
        if (t == 0) {
            const float hi = (float)lred[0];
            so[Ci * ki + ki] = hi;
            so[Ci * ki + ki + 1] = (float)(lred[0] - (double)hi);
        }
    }
}

__global__ __launch_bounds__(HS_CELLS)
void hstats_kernel(const float* __restrict__ H, const float* __restrict__ Y, const float* __restrict__ B,
                   CovMeta meta, float* __restrict__ part, int N, int64_t Np, int KP, float eps, int nstat, int max_k, int max_ct)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_smem[];
    const float* hr = H + ((int64_t)blockIdx.x * HS_CELLS + (threadIdx.x & (HS_CELLS - 1))) * KP;
    hstats_group([hr](int k) { return hr[k]; }, Y, B, meta, part, N, Np, eps, nstat, max_k, max_ct, hs_smem, 0, blockIdx.x);
}

// The two small kernels that open phase 1 (both read only the old H) in ONE launch: blocks [0, gram_blocks) compute the
// partial blocks of H H^T, the rest the per-covariate statistics, two 128-cell groups per 256-thread block.
template <int KT>
__global__ __launch_bounds__(256, (KT <= 2 ? 2 : 1))
void phase1_open_kernel(const float* __restrict__ H, float* __restrict__ gram_part, int R, int rows_per_wave, int gram_blocks,
                        const float* __restrict__ Y, const float* __restrict__ B, CovMeta meta, float* __restrict__ stat_part,
                        int N, int64_t Np, float eps, int nstat, int max_k, int max_ct, int stat_groups)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hs_smem[];
    constexpr int KP = 32 * KT;
    const int b = blockIdx.x;
    if (b < gram_blocks) { gram_block<KT>(H, gram_part, R, rows_per_wave, b); return; }
    const int group = threadIdx.x >> 7;
    int64_t gblock = (int64_t)(b - gram_blocks) * 2 + group;
    // a block whose second group lies past the last cell still walks the same barriers: it works on the last valid
    // group's cells again and writes the same values to the same row
    if (gblock >= stat_groups) gblock = stat_groups - 1;
    const float* hr = H + (gblock * HS_CELLS + (threadIdx.x & (HS_CELLS - 1))) * KP;
    hstats_group([hr](int k) { return hr[k]; }, Y, B, meta, stat_part, N, Np, eps, nstat, max_k, max_ct, hs_smem, group, gblock);
}

// stats[j] = sum over blocks of part[blk][j] in float64; kind[j]: 0 plain, 1 = hi word of a (hi,lo) pair
// whose lo word is j+1 (summed together, re-split), 2 = lo word (written by its hi's block).
// Block nstat writes (hi,lo) of ||X_local||^2 at stats[nstat], stats[nstat+1].
__device__ __forceinline__ void reduce_stats_block(const float* __restrict__ part, const int* __restrict__ kind, float* __restrict__ stats,
                                                   int nblk, int nstat, double xnorm2, int j)
{
    __shared__ double red[256];
    const int t = threadIdx.x;
    if (j == nstat) {
        if (t == 0) {
            const float hi = (float)xnorm2;
            stats[nstat] = hi;
            stats[nstat + 1] = (float)(xnorm2 - (double)hi);
        }
        return;
    }
    const int kd = kind[j];
    if (kd == 2) return;
    double a = 0.0;
    for (int b = t; b < nblk; b += 256) {
        a += (double)part[(int64_t)b * nstat + j];
        if (kd == 1) a += (double)part[(int64_t)b * nstat + j + 1];
    }
    red[t] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) {
        const float hi = (float)red[0];
        stats[j] = hi;
        if (kd == 1) stats[j + 1] = (float)(red[0] - (double)hi);
    }
}

__global__ __launch_bounds__(256)
void reduce_stats_kernel(const float* __restrict__ part, const int* __restrict__ kind, float* __restrict__ stats,
                         int nblk, int nstat, double xnorm2)
{
    reduce_stats_block(part, kind, stats, nblk, nstat, xnorm2, blockIdx.x);
}

// out[j] = sum_s in[s][j] for MANY slabs (one per 128-cell block of the fused H update: hundreds to thousands): a block owns
// 16 consecutive outputs (4 lanes x float4) and spreads the slabs over 64 slab lanes, 8 loads in flight per thread; the
// slab lanes are combined by a fixed tree in float64 -> the sum order depends only on (n, nslab): reproducible.
__device__ __forceinline__ void reduce_slabs_block(const float* __restrict__ in, float* __restrict__ out, int n, int nslab, int block)
{
    __shared__ double red[64][17];                       // [slab lane][16 outputs], padded
    const int q = threadIdx.x & 3, sl = threadIdx.x >> 2;
    const int j0 = block * 16 + 4 * q;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    if (j0 < n) {
        const float* p = in + j0;
        int sidx = sl;
        for (; sidx + 7 * 64 < nslab; sidx += 8 * 64) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (int64_t)(sidx + 64 * u) * n);
#pragma unroll
            for (int u = 0; u < 8; ++u) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
        }
        // the remainder (fewer than 8 slabs per lane: e.g. the 157 partial blocks of W^T W) as ONE batch of predicated loads instead of
        // one dependent round trip per slab; the additions keep their order, so the sums are bit-identical to the sequential form
        {
            f32x4 v[7];
#pragma unroll
            for (int u = 0; u < 7; ++u)
                v[u] = (sidx + 64 * u < nslab) ? *reinterpret_cast<const f32x4*>(p + (int64_t)(sidx + 64 * u) * n) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 7; ++u)
                if (sidx + 64 * u < nslab) { a[0] += v[u][0]; a[1] += v[u][1]; a[2] += v[u][2]; a[3] += v[u][3]; }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[sl][4 * q + e] = a[e];
    __syncthreads();
    for (int half = 32; half > 0; half >>= 1) {
        if (sl < half) {
#pragma unroll
            for (int e = 0; e < 4; ++e) red[sl][4 * q + e] += red[sl + half][4 * q + e];
        }
        __syncthreads();
    }
    if (sl == 0 && j0 < n) {
        f32x4 o = {(float)red[0][4 * q], (float)red[0][4 * q + 1], (float)red[0][4 * q + 2], (float)red[0][4 * q + 3]};
        *reinterpret_cast<f32x4*>(out + j0) = o;
    }
}

// The three reductions that close phase 1 in ONE launch (they are independent of each other): the pieces of the XH^T
// sweep, the partial blocks of H H^T and the per-block covariate statistics.  Blocks [0, nb_pieces) | [.., + nb_many) | rest.
struct Phase1Reduce {
    int nb_pieces, nb_many, nb_stats;
    const float* pieces; float* xht; int rows;
    const float* gram_part; float* hht; int n_hht, n_slab;
    const float* stat_part; const int* kind; float* stats; int stat_blocks, nstat; double xnorm2;
};

__global__ __launch_bounds__(256)
void phase1_reduce_kernel(Phase1Reduce a, int KP, SweepGeom g)
{
    int b = blockIdx.x;
    if (b < a.nb_pieces) { reduce_pieces_block(a.pieces, a.xht, a.rows, KP, g, b, a.nb_pieces); return; }
    b -= a.nb_pieces;
    if (b < a.nb_many) { reduce_slabs_block(a.gram_part, a.hht, a.n_hht, a.n_slab, b); return; }
    b -= a.nb_many;
    reduce_stats_block(a.stat_part, a.kind, a.stats, a.stat_blocks, a.nstat, a.xnorm2, b);
}

// ----------------------------------------------------------------------------------------------
// rows per wave of the (removed) lane-broadcast update kernels of round 1; still the unit in which the host sizes the float64 partials
// of <XH^T, W> (alpine_ctx::ndot): the MFMA W update fills one partial per 32 genes, the rest stay zero
constexpr int UPD_ROWS = 8;

template <int KT>
__device__ __forceinline__ void sg_sum_pieces(const float* __restrict__ pieces, const SweepGeom& g, int ft, int fl, int w_lo, int w_hi,
                                              int h, bool valid, f32x4 (&out)[KT][4])
{
    constexpr int KP = 32 * KT;
    double acc[KT][4][4];
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[m][q][e] = 0.0;
    if (valid) {
        for (int w = w_lo; w <= w_hi; ++w) {
            const float* base = pieces + sg_piece_offset(g, w, ft, KP) + (int64_t)fl * KP + 4 * h;
            f32x4 v[KT][4];
#pragma unroll
            for (int m = 0; m < KT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) v[m][q] = *reinterpret_cast<const f32x4*>(base + 32 * m + 8 * q);
#pragma unroll
            for (int m = 0; m < KT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][q][e] += (double)v[m][q][e];
        }
    }
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[m][q] = f32x4{(float)acc[m][q][0], (float)acc[m][q][1], (float)acc[m][q][2], (float)acc[m][q][3]};
}

// Row-major <-> C/D-layout exchange of one 32-row x KP tile through a wave-private LDS scratch tr[32][KP + 4].
// Global memory is touched with whole rows only (a 32-row tile of a row-major [.][KP] array is one contiguous block of
// 32*KP floats: lane L of instruction i reads / writes the 16 bytes at float offset 4*(64 i + L)); the C/D side is what
// the MFMA formulation of the H / W updates needs (lane (c, h) owns k = 32m + 8q + 4h + e of row c).
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int KT>
__device__ __forceinline__ void tile_lds_to_cd(const float* __restrict__ tr, int c, int h, f32x4 (&out)[KT][4])
{
    constexpr int LD = 32 * KT + 4;
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) out[m][q] = *reinterpret_cast<const f32x4*>(&tr[c * LD + 32 * m + 8 * q + 4 * h]);
}

// tile rows (contiguous at src) -> C/D registers
template <int KT>
__device__ __forceinline__ void tile_load_cd(const float* __restrict__ src, float* __restrict__ tr, int lane, f32x4 (&out)[KT][4])
{
    constexpr int KP = 32 * KT, LD = KP + 4, Q4 = KP / 4, NI = (32 * Q4) / 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = 64 * i + lane;
        *reinterpret_cast<f32x4*>(&tr[(idx / Q4) * LD + 4 * (idx % Q4)]) = *reinterpret_cast<const f32x4*>(src + 4 * idx);
    }
    wave_lds_fence();
    tile_lds_to_cd<KT>(tr, lane & 31, lane >> 5, out);
    wave_lds_fence();
}

// The same in two halves, so that the global loads of several tiles can be in flight before the first LDS exchange:
// tile_load_issue (rows -> raw registers: lane L of instruction i holds the 16 bytes at float offset 4*(64 i + L)) and
// tile_raw_to_cd (raw registers -> C/D registers through the wave's scratch).
template <int KT>
__device__ __forceinline__ void tile_load_issue(const float* __restrict__ src, int lane, f32x4 (&raw)[(32 * KT * 8) / 64])
{
    constexpr int NI = (32 * KT * 8) / 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) raw[i] = *reinterpret_cast<const f32x4*>(src + 4 * (64 * i + lane));
}

template <int KT>
__device__ __forceinline__ void tile_raw_to_cd(const f32x4 (&raw)[(32 * KT * 8) / 64], float* __restrict__ tr, int lane, f32x4 (&out)[KT][4])
{
    constexpr int KP = 32 * KT, LD = KP + 4, Q4 = KP / 4, NI = (32 * Q4) / 64;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = 64 * i + lane;
        *reinterpret_cast<f32x4*>(&tr[(idx / Q4) * LD + 4 * (idx % Q4)]) = raw[i];
    }
    wave_lds_fence();
    tile_lds_to_cd<KT>(tr, lane & 31, lane >> 5, out);
    wave_lds_fence();
}

// C/D registers -> tile rows (contiguous at dst), rows >= rows_valid are not written
template <int KT>
__device__ __forceinline__ void tile_store_cd(float* __restrict__ dst, float* __restrict__ tr, int lane, const f32x4 (&in)[KT][4], int rows_valid)
{
    constexpr int KP = 32 * KT, LD = KP + 4, Q4 = KP / 4, NI = (32 * Q4) / 64;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&tr[c * LD + 32 * m + 8 * q + 4 * h]) = in[m][q];
    wave_lds_fence();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = 64 * i + lane, r = idx / Q4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(&tr[r * LD + 4 * (idx % Q4)]);
        if (r < rows_valid) *reinterpret_cast<f32x4*>(dst + 4 * idx) = v;
    }
    wave_lds_fence();
}

// Sum of the pieces of tile ft for the 32 rows fl0 .. fl0+31 (ascending workgroup order: fixed, reproducible), read as
// whole rows and handed over in C/D layout.  float accumulation: a handful of partial sums that are themselves float
// sums over thousands of products.
template <int KT>
__device__ __forceinline__ void sg_sum_pieces_rows(const float* __restrict__ pieces, const SweepGeom& g, int ft, int fl0, int w_lo, int w_hi,
                                                   float* __restrict__ tr, int lane, f32x4 (&out)[KT][4])
{
    constexpr int KP = 32 * KT, LD = KP + 4, Q4 = KP / 4, NI = (32 * Q4) / 64;
    // UNR pieces (UNR * NI 1-KiB loads per wave) in flight per trip: on the fixed stream-K grid a tile has ~10 pieces at
    // every shard size and a trip per piece costs one full memory latency (measured with in-kernel stamps: 10.7 us of a
    // 40 us block at 25 000 cells with 1 + 4 + 4 + 1 + 1 trips); the adds stay in ascending workgroup order.
    constexpr int UNR = KT <= 2 ? 5 : 2;
    f32x4 acc[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int64_t off_lo = sg_piece_offset(g, w_lo, ft, KP);          // the first workgroup's piece index needs the division
    auto piece = [&](int w) { return pieces + (w == w_lo ? off_lo : sg_piece_offset_inner(g, w, KP)) + (int64_t)fl0 * KP; };
    int w = w_lo;
    for (; w + UNR - 1 <= w_hi; w += UNR) {
        f32x4 v[UNR][NI];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float* bu = piece(w + u);
#pragma unroll
            for (int i = 0; i < NI; ++i) v[u][i] = *reinterpret_cast<const f32x4*>(bu + 4 * (64 * i + lane));
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] += v[u][i];
    }
    if (w + 1 <= w_hi) {                                              // 2 .. UNR-1 pieces left: pairs
        for (; w + 1 <= w_hi; w += 2) {
            const float* b0 = piece(w);
            const float* b1 = piece(w + 1);
            f32x4 v0[NI], v1[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) v0[i] = *reinterpret_cast<const f32x4*>(b0 + 4 * (64 * i + lane));
#pragma unroll
            for (int i = 0; i < NI; ++i) v1[i] = *reinterpret_cast<const f32x4*>(b1 + 4 * (64 * i + lane));
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] += v0[i];
#pragma unroll
            for (int i = 0; i < NI; ++i) acc[i] += v1[i];
        }
    }
    if (w <= w_hi) {
        const float* base = piece(w);
        f32x4 v[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const f32x4*>(base + 4 * (64 * i + lane));
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[i] += v[i];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int idx = 64 * i + lane;
        *reinterpret_cast<f32x4*>(&tr[(idx / Q4) * LD + 4 * (idx % Q4)]) = acc[i];
    }
    wave_lds_fence();
    tile_lds_to_cd<KT>(tr, lane & 31, lane >> 5, out);
    wave_lds_fence();
}

// ----------------------------------------------------------------------------------------------
// h_update on MFMA: one wave = 32 cells, lane (c, h) = cell n0+c, k-half h.
//   den[k][cell] = sum_k' (2 W^TW)[k][k'] * H[cell][k']  as  D = A*B  with  A[k][k'] = M2 (symmetric, read transposed
//   from LDS: conflict-free), B[k'][cell] = H.  The contraction index is visited in the order
//   k'(m,q,e; h) = 32m + 8q + 4h + e, which is exactly the set of k a lane owns in the MFMA C/D layout
//   (row = 8q + 4h + e of tile m, column = cell): so a lane's float4 loads H[cell][32m+8q+4h .. +3] of the cell-major
//   H rows ARE its B operands, and den lands element-for-element on the same registers' k.  No LDS round trip for H.
// Guided terms (main.py:636-650) use the same registers: per covariate/class the lane forms its half of
// (B_i H_i)[c'][cell] over the k it owns and adds the partner half (lane ^ 32).
//
// Fused tail (MU branch on the whole shard): the updated H is the OLD H of the next iteration's phase 1, and every wave
// still holds its 32 x KP tile of it in LDS (the row-major image behind the global store).  So the block also emits what
// phase1_open_kernel would compute from a re-read of H: its partial block of H H^T (gram_part[block], the block's 128 cells
// as the contraction axis, each wave owning output tiles t = wave, wave + 4, ... of the KT x KT grid) and the covariate
// statistics of its 128-cell group (stat_part[block], same arithmetic as hstats_kernel).  The next phase 1 then starts
// directly with the XH^T sweep.  (One group per block, straight-line code: a loop over several groups makes the compiler
// keep the loop-invariant LDS operands in registers and spill.)
// A^T A over the 128 rows that the four waves of a block hold as row-major 32 x (KP + 4) tiles in LDS (trall), written to
// gout[k][k'] (KP x KP): v_mfma_f32_32x32x2_f32 with A[i = k][kk = row], B[kk = row][j = k'], two rows per step; wave w owns
// the output tiles t = w, w + 4, ... of the KT x KT grid, so no cross-wave reduction is needed.  Used by the tails of
// the H update (H H^T of the updated H) and of the W update (W^T W of the updated W).
template <int KT>
__device__ __forceinline__ void gram_of_lds_tiles(const float* __restrict__ trall, int wave, int lane, float* __restrict__ gout)
{
    constexpr int KP = 32 * KT, LD = KP + 4, TRSZ = 32 * LD, NTW = (KT * KT + 3) / 4;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int t = wave + 4 * u;                                        // wave-uniform
        if (KT * KT % 4 != 0 && t >= KT * KT) break;
        const int ta = t / KT, tb = t % KT;
        f32x16 gacc;
#pragma unroll
        for (int e = 0; e < 16; ++e) gacc[e] = 0.f;
        const float* pa = trall + h * LD + 32 * ta + c;                    // row 2p + h of tile (2p + h) >> 5
        const float* pb = trall + h * LD + 32 * tb + c;
#pragma unroll 8
        for (int p2 = 0; p2 < 64; ++p2) {
            const int o = (p2 >> 4) * TRSZ + ((2 * p2) & 31) * LD;
            gacc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[o], pb[o], gacc, 0, 0, 0);
        }
        // D layout: row (k) = 32 ta + (e & 3) + 8 (e >> 2) + 4 h, column (k') = 32 tb + c
#pragma unroll
        for (int e = 0; e < 16; ++e)
            gout[(32 * ta + (e & 3) + 8 * (e >> 2) + 4 * h) * KP + 32 * tb + c] = gacc[e];
    }
}

// Covariate statistics of one 128-cell group inside the fused H update (same quantities and the same per-(class, component)
// arithmetic as hstats_group): the updated H rows come from the waves' LDS tiles, Y from the block's LDS copy, B from the
// block's LDS copy -- no global load on the way, 2-3 barriers per covariate.  256 threads: threads 0..127 own one cell each
// for the per-cell part, all four waves share the reductions over cells.
//   smem: double lred[4]; float hbuf[max_k][128]; float zbuf[max_ct][128]
constexpr int HT_YROWS_MAX = 32;      // rows of Y (sum of the covariates' levels) that the block keeps in LDS (16 KB)

__host__ __device__ inline size_t hstats_tail_bytes(int max_k, int max_ct)
{
    return (sizeof(double) * 4 + sizeof(float) * (size_t)(max_k + max_ct) * HS_CELLS + 15) & ~(size_t)15;
}

__device__ __forceinline__ void hstats_tail(const float* __restrict__ trall, int LD, int TRSZ, const float* __restrict__ ybuf,
                                            const float* __restrict__ Bl, const CovMeta& meta, float* __restrict__ out,
                                            int64_t cell0, int N, float eps, int max_k, int max_ct, unsigned char* __restrict__ smem)
{
    double* lred = reinterpret_cast<double*>(smem);
    float (*hbuf)[HS_CELLS] = reinterpret_cast<float (*)[HS_CELLS]>(lred + 4);
    float (*zbuf)[HS_CELLS] = hbuf + max_k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool cellthread = tid < HS_CELLS;
    const bool valid = cellthread && cell0 + tid < N;
    const float* hrow = trall + (tid >> 5 & 3) * TRSZ + (tid & 31) * LD;
    for (int i = 0; i < meta.n_cov; ++i) {
        const int ki = meta.k[i], Ci = meta.lev[i], off = meta.off[i], bo = meta.boff[i];
        const float lam = meta.lam[i];
        float* so = out + meta.soff[i];
        __syncthreads();                                       // the previous covariate's readers are done with hbuf / zbuf
        if (cellthread)
            for (int k = 0; k < ki; ++k) hbuf[k][tid] = valid ? hrow[off + k] : 0.f;
        double lacc = 0.0;
        for (int c0 = 0; c0 < Ci; c0 += max_ct) {
            const int ct = min(max_ct, Ci - c0);
            if (c0 > 0) __syncthreads();                       // zbuf is reused by the next chunk of classes
            if (cellthread) {
                for (int c = 0; c < ct; ++c) {
                    float bh = 0.f;
                    for (int k = 0; k < ki; ++k) bh = fmaf(Bl[bo + (c0 + c) * ki + k], hbuf[k][tid], bh);
                    const float y = valid ? ybuf[(meta.yoff[i] + c0 + c) * HS_CELLS + tid] : 0.f;
                    float z;
                    if (meta.loss_type == 0) {
                        const float yh = fmaxf(bh, eps);
                        z = lam * (y / yh);
                        if (valid) lacc += (double)(y * logf(fmaxf(y / yh, eps)) - y + yh);
                    } else {
                        z = y;
                        const float d = y - bh;
                        if (valid) lacc += (double)(d * d);
                    }
                    zbuf[c][tid] = valid ? z : 0.f;
                }
            }
            __syncthreads();
            for (int p = wave; p < ct * ki; p += 4) {
                const int c = p / ki, k = p % ki;
                float v = zbuf[c][lane] * hbuf[k][lane];
                v = fmaf(zbuf[c][lane + 64], hbuf[k][lane + 64], v);
                v = wave_sum_f32_dpp(v);
                if (lane == 0) so[(c0 + c) * ki + k] = v;
            }
        }
        for (int k = wave; k < ki; k += 4) {
            float v = lam * hbuf[k][lane] + lam * hbuf[k][lane + 64];
            v = wave_sum_f32_dpp(v);
            if (lane == 0) so[Ci * ki + k] = v;
        }
        const double ws = wave_sum_f64_dpp(lacc);                  // waves 2, 3 hold zeros
        if (lane == 0) lred[wave] = ws;
        __syncthreads();
        if (tid == 0) {
            const double tot = lred[0] + lred[1];
            const float hi = (float)tot;
            so[Ci * ki + ki] = hi;
            so[Ci * ki + ki + 1] = (float)(tot - (double)hi);
        }
    }
}

// The same statistics for ALL covariates in one pass (two block barriers in all instead of two or three per covariate): used
// when the guided components and label levels of all covariates together fit HT_MERGED_ROWS rows of 128 floats of scratch.
// Same per-(class, component) arithmetic and summation order as hstats_tail -> bitwise the same statistics.
//   smem: double lred[MAX_COV][2]; float hbuf[sum k_i][128]; float zbuf[sum C_i][128]
constexpr int HT_MERGED_ROWS = 32;
__host__ __device__ inline size_t hstats_merged_bytes(int guided, int nY)
{
    return (sizeof(double) * MAX_COV * 2 + sizeof(float) * (size_t)(guided + nY) * HS_CELLS + 15) & ~(size_t)15;
}

__device__ __forceinline__ void hstats_tail_merged(const float* __restrict__ trall, int LD, int TRSZ, const float* __restrict__ ybuf,
                                                   const float* __restrict__ Bl, const CovMeta& meta, float* __restrict__ out,
                                                   int64_t cell0, int N, float eps, int guided, unsigned char* __restrict__ smem)
{
    double* lred = reinterpret_cast<double*>(smem);                                     // [MAX_COV][2]
    float (*hbuf)[HS_CELLS] = reinterpret_cast<float (*)[HS_CELLS]>(lred + MAX_COV * 2);   // [sum k_i][128], row off_i + k
    float (*zbuf)[HS_CELLS] = hbuf + guided;                                            // [sum C_i][128], row yoff_i + c
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool cellthread = tid < HS_CELLS;
    const bool valid = cellthread && cell0 + tid < N;
    const float* hrow = trall + (tid >> 5 & 3) * TRSZ + (tid & 31) * LD;
    if (cellthread) {
        for (int i = 0; i < meta.n_cov; ++i) {
            const int ki = meta.k[i], Ci = meta.lev[i], off = meta.off[i], bo = meta.boff[i], yo = meta.yoff[i];
            const float lam = meta.lam[i];
            for (int k = 0; k < ki; ++k) hbuf[off + k][tid] = valid ? hrow[off + k] : 0.f;
            double lacc = 0.0;
            for (int c = 0; c < Ci; ++c) {
                float bh = 0.f;
                for (int k = 0; k < ki; ++k) bh = fmaf(Bl[bo + c * ki + k], hbuf[off + k][tid], bh);
                const float y = valid ? ybuf[(yo + c) * HS_CELLS + tid] : 0.f;
                float z;
                if (meta.loss_type == 0) {
                    const float yh = fmaxf(bh, eps);
                    z = lam * (y / yh);
                    if (valid) lacc += (double)(y * logf(fmaxf(y / yh, eps)) - y + yh);
                } else {
                    z = y;
                    const float d = y - bh;
                    if (valid) lacc += (double)(d * d);
                }
                zbuf[yo + c][tid] = valid ? z : 0.f;
            }
            const double ws = wave_sum_f64_dpp(lacc);                  // waves 0 and 1 hold the block's 128 cells
            if (lane == 0) lred[2 * i + wave] = ws;
        }
    }
    __syncthreads();
    // every (covariate, class, component) product and every (covariate, component) sum is one wave-wide reduction over the 128
    // cells; the four waves take them round-robin
    int p0 = 0;
    for (int i = 0; i < meta.n_cov; ++i) {
        const int ki = meta.k[i], Ci = meta.lev[i], off = meta.off[i], yo = meta.yoff[i];
        const float lam = meta.lam[i];
        float* so = out + meta.soff[i];
        const int items = Ci * ki + ki;
        for (int p = (wave - p0 % 4 + 4) % 4; p < items; p += 4) {
            float v;
            if (p < Ci * ki) {
                const int c = p / ki, k = p - c * ki;
                v = zbuf[yo + c][lane] * hbuf[off + k][lane];
                v = fmaf(zbuf[yo + c][lane + 64], hbuf[off + k][lane + 64], v);
            } else {
                const int k = p - Ci * ki;
                v = lam * hbuf[off + k][lane] + lam * hbuf[off + k][lane + 64];
            }
            v = wave_sum_f32_dpp(v);
            if (lane == 0) so[p] = v;                                  // [bnum: C_i x k_i][bden: k_i] are contiguous
        }
        p0 += items;
        if (tid == i) {
            const double tot = lred[2 * i] + lred[2 * i + 1];
            const float hi = (float)tot;
            so[Ci * ki + ki] = hi;
            so[Ci * ki + ki + 1] = (float)(tot - (double)hi);
        }
    }
}

// stacked row r of Y: first column / columns of its covariate, offset of B_i[r - yoff_i][0] in the packed B, lam (Fro: 2 lam)
struct GuidedRow { int off, k, base; float lam; };

struct HTail {
    float* gram_part;       // [gridDim.x][KP*KP], nullptr = no tail
    float* stat_part;       // [groups][nstat]
    int nstat, max_k, max_ct;
    int ybuf_rows;          // rows of Y the block copies into LDS (all of them when they fit HT_YROWS_MAX, else 0 = read Y from global)
    int merged;             // 1: the tail's statistics for all covariates in one pass (hstats_tail_merged; needs the Y copy in LDS)
    int guided;             // sum of the k_i
    int gm;                 // 1: guided terms as two small matrix products on the MFMA (below); needs 0 < nY <= 32 and the Y copy in LDS
    int nY, kg;             // gm: rows of Y in total; guided columns rounded up to a multiple of 8
    const struct GuidedRow* rowtab;   // gm: [32] device table, stacked row of Y -> its covariate's columns / B rows / lam
};

// Guided terms of the H update as two small matrix products (gm path).  With every covariate's B_i stacked into ONE
// block-structured matrix Ball[r][k] (r = row of the stacked Y: covariate i's classes are rows yoff_i .., k = column of H:
// covariate i's components are columns off_i ..; zero elsewhere) main.py:636-650 reads
//     bh  = Ball H              (all classes of all covariates at once: [nY] x cells)
//     KL : num += (lam Ball)^T (Y / max(bh, eps)),  den += (lam Ball)^T 1        Fro: num += (2 lam Ball)^T Y,  den += (2 lam Ball)^T bh
// and the block structure makes "covariate i's terms touch only covariate i's columns" automatic (also for the block-coordinate
// branch, which updates one covariate's columns at a time).  Both products run on v_mfma_f32_32x32x2_f32 with the register
// trick of the 2 W^TW H product: the contraction index is visited in the order a lane holds it (C/D layout), so H (for bh) and
// z = Y / bh (for the numerator) are B operands straight from the registers they are in; the A operands are gathered from the
// block's LDS copy of the packed B through a 32-entry row table (no stacked copy is materialised).  12 + 4 MFMAs at cfg3 instead
// of a dependent scalar chain with a cross-lane exchange per (covariate, class): 7.7 -> 3.x us of a 32 us block (in-kernel stamps).

// bh[r][cell] = sum_k Ball[r][k] H[cell][k] over the guided columns, H in C/D registers (lane (c, h): cell c, k = 32m + 8q + 4h + e);
// result in C/D layout: lane (c, h) holds rows r = 8q' + 4h + e' of cell c.  A operand of lane (c, h): Ball[r = c][k].
template <int KT, int GT>
__device__ __forceinline__ void guided_bh(const float* __restrict__ Bl, const GuidedRow& mine, int kg, const f32x4 (&hreg)[KT][4], int h, f32x16& bh)
{
#pragma unroll
    for (int e = 0; e < 16; ++e) bh[e] = 0.f;
#pragma unroll
    for (int m = 0; m < GT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (32 * m + 8 * q >= kg) continue;                                     // wave-uniform
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = 32 * m + 8 * q + 4 * h + e - mine.off;
                const bool in = (unsigned)kk < (unsigned)mine.k;
                const float a = Bl[mine.base + (in ? kk : 0)];
                bh = __builtin_amdgcn_mfma_f32_32x32x2f32(in ? a : 0.f, hreg[m][q][e], bh, 0, 0, 0);
            }
        }
}

// LOSS: 0 = KL, 1 = Frobenius guided terms (a template parameter: as a run-time select every element of the guided loops
// computed both forms and picked one).
template <int KT, int LOSS>
__global__ __launch_bounds__(256, (KT <= 2 ? 2 : 1))
void h_update_mfma_kernel(float* __restrict__ H, const float* __restrict__ pieces, SweepGeom g,
                          const float* __restrict__ WtW, const float* __restrict__ Y, const float* __restrict__ B,
                          CovMeta meta, int N, int64_t Np, int K, float eps, int nB, int k_lo, int k_hi, int only_cov, HTail tail)
{
    // MU branch: k_lo = 0, k_hi = K, only_cov = -1.  Block-coordinate branch (main.py:564-588): only the rows
    // [k_lo, k_hi) of one component group are updated; guided terms only for that group's covariate (only_cov),
    // none for the unguided group (only_cov = n_cov).
    constexpr int KP = 32 * KT, LD = KP + 4, TRSZ = 32 * LD;
    constexpr int GT = KT;                         // k tiles that can hold guided columns: all of them (sum k_i up to K)
    extern __shared__ float smem[];
    float* M2l = smem;                             // [k'][k] = 2 * WtW
    float* Bl = smem + KP * KP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    float* trall = smem + KP * KP + ((nB + 3) & ~3);                 // 4 x [32][KP + 4]: the waves' tiles of one 128-cell group
    float* tr = trall + wave * TRSZ;
    float* ybuf = trall + 4 * TRSZ;                                                 // [ybuf_rows][128]: Y of the block's cells
    const bool gm = tail.gm != 0;                                                   // kernel-uniform
    const int ybuf_alloc = gm ? (tail.ybuf_rows + 7) / 8 * 8 : tail.ybuf_rows;      // gm: padded with zero rows to a multiple of 8
    GuidedRow* rowtab = reinterpret_cast<GuidedRow*>(ybuf + ybuf_alloc * HS_CELLS);  // gm: [32] stacked rows of Y -> their covariate
    unsigned char* hs_smem = reinterpret_cast<unsigned char*>(rowtab + (gm ? 32 : 0));   // tail only: statistics scratch
    const bool with_tail = tail.gram_part != nullptr;
    const bool y_lds = tail.ybuf_rows > 0;
    HU_STAMP(0);
    const int64_t grp = blockIdx.x;                                            // 128-cell group
    const int64_t n0 = (grp * 4 + wave) * 32;                                  // this wave's 32 cells
    const bool active = n0 < N;                                                // wave-uniform
    {
        // Y of this block's 128 cells: one load phase up front instead of a dependent global load per (covariate, class)
        const int64_t cell0 = (int64_t)blockIdx.x * HS_CELLS;
        const int ytotal = ybuf_alloc * HS_CELLS;
        for (int idx0 = tid; idx0 < ytotal; idx0 += 4 * 256) {               // four loads in flight per thread (run-time trip count: not unrolled otherwise)
            float yv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int idx = idx0 + 256 * u, row = idx >> 7, t = idx & (HS_CELLS - 1);
                yv[u] = (idx < ytotal && row < tail.ybuf_rows && cell0 + t < N) ? Y[(int64_t)row * Np + cell0 + t] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (idx0 + 256 * u < ytotal) ybuf[idx0 + 256 * u] = yv[u];
        }
    }
    for (int idx = tid; idx < KP * KP / 4; idx += 256)
        reinterpret_cast<f32x4*>(M2l)[idx] = 2.f * reinterpret_cast<const f32x4*>(WtW)[idx];
    for (int idx = tid; idx < nB; idx += 256) Bl[idx] = B[idx];
    if (gm && tid < 32) rowtab[tid] = tail.rowtab[tid];                        // built once on the host (launch_h_update)
    __syncthreads();
    HU_STAMP(1);

    {
        f32x4 hreg[KT][4];
        bool valid = false;
        if (active) {
            const int64_t n = n0 + c;
            valid = n < N;

            const int ft = (int)(n0 / g.bf), fl0 = (int)(n0 % g.bf);
            int w_lo, w_hi;
            sg_tile_pieces(g, ft, w_lo, w_hi);

            // whole-row global accesses, C/D layout in registers (rows n0 .. n0+31 exist: H and the pieces are padded to 128 rows).
            // (Issuing the H tile and the first pieces BEFORE the fills and their barrier does not pay: loads return in order, so
            // the fills then wait behind 24 KB per wave -- fills + pieces 8.4 -> 10.9 us at 200 000 cells in the in-kernel stamps.)
            f32x4 xreg[KT][4];
            {
                f32x4 hraw[(32 * KT * 8) / 64];
                tile_load_issue<KT>(H + n0 * KP, lane, hraw);          // in flight behind the pieces' loads
                sg_sum_pieces_rows<KT>(pieces, g, ft, fl0, w_lo, w_hi, tr, lane, xreg);
                tile_raw_to_cd<KT>(hraw, tr, lane, hreg);
            }
            HU_STAMP(2);

            f32x16 acc[KT];
#pragma unroll
            for (int mo = 0; mo < KT; ++mo)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mo][e] = 0.f;
#pragma unroll
            for (int m = 0; m < KT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float* mrow = M2l + (32 * m + 8 * q + 4 * h + e) * KP + c;
#pragma unroll
                        for (int mo = 0; mo < KT; ++mo)
                            acc[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(mrow[32 * mo], hreg[m][q][e], acc[mo], 0, 0, 0);
                    }
            HU_STAMP(3);

            // numerator = 2 W^TX (+ guided), denominator = (2 W^TW) H (+ guided): accumulated IN PLACE in xreg / acc so that the
            // kernel's live state stays at three tiles (H, numerator, denominator) -> 2 waves per SIMD
#pragma unroll
            for (int m = 0; m < KT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) xreg[m][q] = 2.f * xreg[m][q];
            if (gm) {
                // guided terms as two small products on the MFMA (see GuidedRow)
                f32x16 bh;
                guided_bh<KT, GT>(Bl, rowtab[c], tail.kg, hreg, h, bh);
                f32x16 zz;
#pragma unroll
                for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                    for (int ee = 0; ee < 4; ++ee) {
                        if (8 * qq >= tail.nY) { zz[4 * qq + ee] = 0.f; continue; }              // wave-uniform
                        const float y = ybuf[(8 * qq + 4 * h + ee) * HS_CELLS + wave * 32 + c];
                        zz[4 * qq + ee] = (LOSS == 0) ? y / fmaxf(bh[4 * qq + ee], eps) : y;
                    }
#pragma unroll
                for (int m = 0; m < GT; ++m) {
                    if (32 * m >= tail.kg) continue;                                              // wave-uniform
                    f32x16 nacc;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) nacc[4 * q + e] = xreg[m][q][e];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) {
                        if (8 * qq >= tail.nY) continue;                                          // wave-uniform
#pragma unroll
                        for (int ee = 0; ee < 4; ++ee) {
                            // A operand of lane (c, h): (lam B)[r][k] with r = 8q' + 4h + e', k = 32m + c
                            const GuidedRow gr = rowtab[8 * qq + 4 * h + ee];
                            const int kk = 32 * m + c - gr.off;
                            const bool in = (unsigned)kk < (unsigned)gr.k;
                            const float bv = Bl[gr.base + (in ? kk : 0)];
                            const float a2 = in ? gr.lam * bv : 0.f;
                            nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, zz[4 * qq + ee], nacc, 0, 0, 0);
                            // KL: den += sum_r lam B[r][k] (a product with ones); Fro: den += sum_r 2 lam B[r][k] bh[r]
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, (LOSS == 0) ? 1.f : bh[4 * qq + ee], acc[m], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) xreg[m][q][e] = nacc[4 * q + e];
                }
            } else {
            // A covariate's k_i components live in [off, off + k_i): of the 8 groups of 8 components that a lane pair (h = 0, 1)
            // holds in a k tile only those that overlap the range are touched (wave-uniform branches on scalars) -- with
            // k_i = 5 that is one group of 8 instead of all 64 components, per covariate and class (in-kernel stamps: the
            // guided terms were 10.6 us of a 40 us block when every group ran through selects).
            for (int i = 0; i < meta.n_cov; ++i) {
                if (only_cov >= 0 && i != only_cov) continue;
                const int off = meta.off[i], ki = meta.k[i], Ci = meta.lev[i], bo = meta.boff[i], yo = meta.yoff[i];
                const float lam = (LOSS == 0) ? meta.lam[i] : meta.lam2[i];
                for (int cl = 0; cl < Ci; ++cl) {
                    const float* brow = Bl + bo + cl * ki - off;            // brow[k] = B_i[cl][k - off]
                    float part = 0.f;
#pragma unroll
                    for (int m = 0; m < GT; ++m)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (32 * m + 8 * q + 8 <= off || 32 * m + 8 * q >= off + ki) continue;      // wave-uniform
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int k = 32 * m + 8 * q + 4 * h + e;
                                const float coef = (k >= off && k < off + ki) ? brow[k] : 0.f;
                                part = fmaf(coef, hreg[m][q][e], part);
                            }
                        }
                    const float bh = part + __shfl_xor(part, 32, 64);
                    const float y = y_lds ? ybuf[(yo + cl) * HS_CELLS + wave * 32 + c]
                                          : (valid ? Y[(int64_t)(yo + cl) * Np + n] : 0.f);
                    const float z = (LOSS == 0) ? y / fmaxf(bh, eps) : y;
#pragma unroll
                    for (int m = 0; m < GT; ++m)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (32 * m + 8 * q + 8 <= off || 32 * m + 8 * q >= off + ki) continue;      // wave-uniform
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int k = 32 * m + 8 * q + 4 * h + e;
                                const float lb = (k >= off && k < off + ki) ? lam * brow[k] : 0.f;
                                xreg[m][q][e] = fmaf(lb, z, xreg[m][q][e]);
                                acc[m][4 * q + e] = (LOSS == 0) ? acc[m][4 * q + e] + lb : fmaf(lb, bh, acc[m][4 * q + e]);
                            }
                        }
                }
            }
            }

#pragma unroll
            for (int m = 0; m < KT; ++m)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k4 = 32 * m + 8 * q + 4 * h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = hreg[m][q][e] * fast_div(xreg[m][q][e], fmaxf(acc[m][4 * q + e], eps));
                        const int k = k4 + e;
                        if (k >= k_lo && k < k_hi) hreg[m][q][e] = v;           // outside the range (and pads, which are 0): unchanged
                        if (with_tail && !valid) hreg[m][q][e] = 0.f;           // rows past the last cell must not enter the sums of the tail
                    }
                }
            (void)K;
            HU_STAMP(4);
            // leaves the updated tile row-major in tr (the tail reads it there)
            tile_store_cd<KT>(H + n0 * KP, tr, lane, hreg, (int)min((int64_t)32, (int64_t)N - n0));
        } else if (with_tail) {
            for (int idx = lane; idx < TRSZ / 4; idx += 64) reinterpret_cast<f32x4*>(tr)[idx] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (!with_tail) return;                                                // kernel-uniform

        HU_STAMP(5);
        __syncthreads();                                                       // the four tiles of this group are in LDS
        gram_of_lds_tiles<KT>(trall, wave, lane, tail.gram_part + (int64_t)blockIdx.x * KP * KP);      // H H^T over the block's 128 cells
        HU_STAMP(6);
        if (meta.n_cov > 0) {
            if (y_lds && tail.merged) {
                hstats_tail_merged(trall, LD, TRSZ, ybuf, Bl, meta, tail.stat_part + grp * tail.nstat, grp * HS_CELLS, N, eps, tail.guided, hs_smem);
            } else if (y_lds) {
                hstats_tail(trall, LD, TRSZ, ybuf, Bl, meta, tail.stat_part + grp * tail.nstat, grp * HS_CELLS, N, eps, tail.max_k, tail.max_ct, hs_smem);
            } else {
                // many label levels: the stand-alone arithmetic with Y from global memory; threads 128..255 walk the same
                // barriers on the same cells with their own scratch slice and write the same values
                const int t7 = tid & (HS_CELLS - 1);
                const float* hr = trall + (t7 >> 5) * TRSZ + (t7 & 31) * LD;
                hstats_group([hr](int k) { return hr[k]; }, Y, B, meta, tail.stat_part, N, Np, eps, tail.nstat, tail.max_k, tail.max_ct,
                             hs_smem, tid >> 7, grp);
            }
        }
        HU_STAMP(7);
    }
}

// ----------------------------------------------------------------------------------------------
// w_update on MFMA, same register trick as h_update_mfma_kernel with genes in place of cells:
//   den[k][gene] = sum_k' M[k'][k] * W[gene][k'],   M = 2 HH^T + orth * (coupled off-diagonal) + l2 * I  (LDS, [k'][k])
//   W[gene][k]  *= (2 XH^T[gene][k]) / max(den + l1, eps)   for k in [k_lo, k_hi)              main.py:596-605 / :533-545
// plus the float64 partial of <XH^T, W_old> over the wave's 32 genes (trace-form loss), dotpart[wave].
template <int KT>
__global__ __launch_bounds__(256)
void w_update_mfma_kernel(float* __restrict__ W, const float* __restrict__ XHt, const float* __restrict__ HHt,
                          double* __restrict__ dotpart, int G, int K, float orth, float l2, float l1, float eps,
                          int do_update, int k_lo, int k_hi, int block_orth, float* __restrict__ gram_part)
{
    // gram_part != nullptr (MU branch): the block also writes its partial block of W_new^T W_new (the 128 updated rows are
    // still in LDS behind the global store) -> no separate pass over W for W^T W.
    constexpr int KP = 32 * KT, LD = KP + 4, TRSZ = 32 * LD;
    extern __shared__ float Ml[];
    float* trall = Ml + KP * KP;                    // 4 x [32][KP + 4] row-major tiles of the updated rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    float* tr = trall + wave * TRSZ;
    if (do_update) {
        // float4 per thread and trip, four loads in flight (the scalar form of this loop was one L2 round trip per trip: 16 trips at
        // K <= 64, 64 at K <= 128, on the critical chain of every block)
#pragma unroll 4
        for (int idx4 = tid; idx4 < KP * KP / 4; idx4 += 256) {
            const int kp = idx4 / (KP / 4), k0 = 4 * (idx4 % (KP / 4));
            const f32x4 src = reinterpret_cast<const f32x4*>(HHt)[idx4];
            const bool coupled = !block_orth || (kp >= k_lo && kp < k_hi);
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k0 + e;
                v[e] = (kp < K && k < K) ? 2.f * src[e] + (kp == k ? l2 : (coupled ? orth : 0.f)) : 0.f;
            }
            reinterpret_cast<f32x4*>(Ml)[idx4] = v;
        }
        __syncthreads();
    }
    const int gw = blockIdx.x * 4 + wave;
    const int64_t g0 = (int64_t)gw * 32;
    const int64_t g = g0 + c;
    const bool valid = g < G;
    // W and XH^T are padded to 128 rows: the wave's 32 rows exist; whole-row loads, C/D layout in registers
    f32x4 wreg[KT][4], xreg[KT][4];
    {
        f32x4 wraw[(32 * KT * 8) / 64], xraw[(32 * KT * 8) / 64];
        tile_load_issue<KT>(W + g0 * KP, lane, wraw);
        tile_load_issue<KT>(XHt + g0 * KP, lane, xraw);
        tile_raw_to_cd<KT>(wraw, tr, lane, wreg);
        tile_raw_to_cd<KT>(xraw, tr, lane, xreg);
    }
    double dacc = 0.0;
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) dacc += (double)xreg[m][q][e] * (double)wreg[m][q][e];
    if (!valid) dacc = 0.0;
    dacc = wave_sum_f64_dpp(dacc);                       // (DPP path: six ds_bpermute round trips less on the block's critical chain)
    if (lane == 0) dotpart[gw] = dacc;
    if (!do_update) return;

    f32x16 acc[KT];
#pragma unroll
    for (int mo = 0; mo < KT; ++mo)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mo][e] = 0.f;
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* mrow = Ml + (32 * m + 8 * q + 4 * h + e) * KP + c;      // A[i = k][kk = k'] = M[k'][k]
#pragma unroll
                for (int mo = 0; mo < KT; ++mo)
                    acc[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(mrow[32 * mo], wreg[m][q][e], acc[mo], 0, 0, 0);
            }
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k4 = 32 * m + 8 * q + 4 * h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = k4 + e;
                const float d = fmaxf(acc[m][4 * q + e] + l1, eps);
                const float v = wreg[m][q][e] * fast_div(2.f * xreg[m][q][e], d);
                if (k >= k_lo && k < k_hi) wreg[m][q][e] = v;
                if (!valid) wreg[m][q][e] = 0.f;                                      // pad rows stay exactly zero
            }
        }
    // full-line stores through the wave's LDS tile, which stays behind for the W^T W partial
    tile_store_cd<KT>(W + g0 * KP, tr, lane, wreg, (int)min((int64_t)32, (int64_t)G - g0));
    if (gram_part == nullptr) return;                                                // kernel-uniform
    __syncthreads();
    gram_of_lds_tiles<KT>(trall, wave, lane, gram_part + (int64_t)blockIdx.x * KP * KP);
}

// ----------------------------------------------------------------------------------------------
// transform (alpine/main.py:705-709): n_iter times  H *= (2 W^T X) / max((2 W^T W) H, eps)  with W frozen.
// The numerator is loop-invariant (one W^TX sweep) and every cell is independent, so all iterations of a
// 32-cell tile run in registers: the updated H in the MFMA C/D layout is directly the next iteration's B operand
// (same k'(m,q,e;h) ordering as h_update_mfma_kernel).  One read of the pieces and H, one write of H.
template <int KT>
__global__ __launch_bounds__(256)
void h_iterate_mfma_kernel(float* __restrict__ H, const float* __restrict__ pieces, SweepGeom g,
                           const float* __restrict__ WtW, int N, int K, float eps, int n_iter)
{
    constexpr int KP = 32 * KT;
    extern __shared__ float smem[];
    float* M2l = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    for (int idx = tid; idx < KP * KP / 4; idx += 256)
        reinterpret_cast<f32x4*>(M2l)[idx] = 2.f * reinterpret_cast<const f32x4*>(WtW)[idx];
    __syncthreads();
    const int64_t n0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (n0 >= N) return;
    const int64_t n = n0 + c;
    const bool valid = n < N;
    const int ft = (int)(n0 / g.bf), fl = (int)(n0 % g.bf) + c;
    int w_lo, w_hi;
    sg_tile_pieces(g, ft, w_lo, w_hi);

    f32x4 hreg[KT][4], num[KT][4];
    sg_sum_pieces<KT>(pieces, g, ft, fl, w_lo, w_hi, h, valid, num);
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            hreg[m][q] = valid ? *reinterpret_cast<const f32x4*>(H + n * KP + 32 * m + 8 * q + 4 * h) : f32x4{0.f, 0.f, 0.f, 0.f};
            num[m][q] = 2.f * num[m][q];
        }

    for (int it = 0; it < n_iter; ++it) {
        f32x16 acc[KT];
#pragma unroll
        for (int mo = 0; mo < KT; ++mo)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mo][e] = 0.f;
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* mrow = M2l + (32 * m + 8 * q + 4 * h + e) * KP + c;
#pragma unroll
                    for (int mo = 0; mo < KT; ++mo)
                        acc[mo] = __builtin_amdgcn_mfma_f32_32x32x2f32(mrow[32 * mo], hreg[m][q][e], acc[mo], 0, 0, 0);
                }
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    hreg[m][q][e] = hreg[m][q][e] * (num[m][q][e] / fmaxf(acc[m][4 * q + e], eps));
    }

    if (!valid) return;
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k4 = 32 * m + 8 * q + 4 * h;
            if (k4 >= K) continue;
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (k4 + e < K) ? hreg[m][q][e] : 0.f;
            *reinterpret_cast<f32x4*>(H + n * KP + k4) = o;
        }
}

// ----------------------------------------------------------------------------------------------
// b_update (one block): B_i <- B_i * num / max(den, eps), main.py:615-628, from the reduced statistics.
__device__ __forceinline__ void b_update_block(const float* __restrict__ Bold, float* __restrict__ Bnew, const float* __restrict__ stats,
                     const float* __restrict__ HHt, CovMeta meta, int KP, float eps)
{
    for (int i = 0; i < meta.n_cov; ++i) {
        const int ki = meta.k[i], Ci = meta.lev[i], off = meta.off[i];
        const float* st = stats + meta.soff[i];
        for (int idx = threadIdx.x; idx < Ci * ki; idx += blockDim.x) {
            const int c = idx / ki, k = idx % ki;
            const float b = Bold[meta.boff[i] + idx];
            float num, den;
            if (meta.loss_type == 0) {
                num = st[idx];
                den = st[Ci * ki + k];
            } else {
                num = 2.f * st[idx];
                den = 0.f;
                for (int kk = 0; kk < ki; ++kk)
                    den = fmaf(2.f * Bold[meta.boff[i] + c * ki + kk], HHt[(off + kk) * KP + off + k], den);
            }
            Bnew[meta.boff[i] + idx] = b * (num / fmaxf(den, eps));
        }
    }
}

// loss_finalize (one block): row = [total, recon, pred_1..pred_C] in float64, main.py:726-753 with
// recon = ||X||^2 - 2 <XH^T, W> + <W^TW, HH^T>  (all three terms belong to the same (W, H)).
__device__ __forceinline__ void loss_finalize_block(const double* __restrict__ dotpart, int ndot, const float* __restrict__ WtW,
                          const float* __restrict__ HHt, const float* __restrict__ stats, CovMeta meta,
                          int nstat, int KP, const double* __restrict__ lam64, double* __restrict__ row)
{
    __shared__ double red[256];
    const int t = threadIdx.x;
    // batches of four independent loads per thread (as scalar loops these were one L2 round trip per trip -- 10 + 16 trips at K <= 64,
    // 10 + 64 at K <= 128 -- and this block is the longest of its launch)
    double a = 0.0;
    int i = t;
    for (; i + 3 * 256 < ndot; i += 4 * 256) {
        const double d0 = dotpart[i], d1 = dotpart[i + 256], d2 = dotpart[i + 512], d3 = dotpart[i + 768];
        a -= 2.0 * d0; a -= 2.0 * d1; a -= 2.0 * d2; a -= 2.0 * d3;
    }
    for (; i < ndot; i += 256) a -= 2.0 * dotpart[i];
    const int n4 = KP * KP / 4;                                   // KP is a multiple of 32
    const f32x4* w4 = reinterpret_cast<const f32x4*>(WtW);
    const f32x4* h4 = reinterpret_cast<const f32x4*>(HHt);
    for (i = t; i < n4; i += 4 * 256) {
        f32x4 w[4], hh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = i + 256 * u < n4;
            w[u] = in ? w4[i + 256 * u] : f32x4{0.f, 0.f, 0.f, 0.f};
            hh[u] = in ? h4[i + 256 * u] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) a += (double)w[u][e] * (double)hh[u][e];
    }
    red[t] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) {
        const double xn = (double)stats[nstat] + (double)stats[nstat + 1];
        const double recon = xn + red[0];
        double total = recon;
        for (int i = 0; i < meta.n_cov; ++i) {
            const int o = meta.soff[i] + meta.lev[i] * meta.k[i] + meta.k[i];
            const double pl = (double)stats[o] + (double)stats[o + 1];
            row[2 + i] = pl;
            total += lam64[i] * pl;
        }
        row[0] = total;
        row[1] = recon;
    }
}

__global__ __launch_bounds__(256)
void b_update_kernel(const float* __restrict__ Bold, float* __restrict__ Bnew, const float* __restrict__ stats,
                     const float* __restrict__ HHt, CovMeta meta, int KP, float eps)
{
    b_update_block(Bold, Bnew, stats, HHt, meta, KP, eps);
}

__global__ __launch_bounds__(256)
void loss_finalize_kernel(const double* __restrict__ dotpart, int ndot, const float* __restrict__ WtW,
                          const float* __restrict__ HHt, const float* __restrict__ stats, CovMeta meta,
                          int nstat, int KP, const double* __restrict__ lam64, double* __restrict__ row)
{
    loss_finalize_block(dotpart, ndot, WtW, HHt, stats, meta, nstat, KP, lam64, row);
}

// The middle of phase 2 in ONE launch: after the W update three things depend on its result and on nothing else of each
// other -- the partial blocks of W^TW (gram), the pending loss row (needs the <XH^T, W_old> partials and the OLD W^TW, which
// reduce_many only overwrites afterwards) and the B updates.  Blocks [0, gram_blocks) run the gram, the next two the tails.
struct Phase2Mid {
    int gram_blocks, do_loss, do_b;
    const double* dotpart; int ndot; const float* WtW; const float* HHt; const float* stats; int nstat; const double* lam64; double* row;
    const float* Bold; float* Bnew; float eps;
};

// After the fused W update (w_update_mfma_kernel with gram_part): ONE launch for the reduction of its W^T W partial blocks
// into the OTHER W^T W buffer (blocks [0, nb_slabs)), the pending loss row (reads the CURRENT buffer = W^T W of the W that
// produced the reduce block) and the B updates.  The host then swaps the two W^T W buffers.
struct Phase2Tail {
    int nb_slabs, n_slab, do_loss, do_b;
    const float* gram_part; float* WtW_new;
    const double* dotpart; int ndot; const float* WtW_old; const float* HHt; const float* stats; int nstat; const double* lam64; double* row;
    const float* Bold; float* Bnew; float eps;
};

__global__ __launch_bounds__(256)
void phase2_tail_kernel(Phase2Tail a, CovMeta meta, int KP)
{
    const int b = blockIdx.x;
    if (b < a.nb_slabs) { reduce_slabs_block(a.gram_part, a.WtW_new, KP * KP, a.n_slab, b); return; }
    if (b == a.nb_slabs) {
        if (a.do_loss) loss_finalize_block(a.dotpart, a.ndot, a.WtW_old, a.HHt, a.stats, meta, a.nstat, KP, a.lam64, a.row);
        return;
    }
    if (a.do_b) b_update_block(a.Bold, a.Bnew, a.stats, a.HHt, meta, KP, a.eps);
}

template <int KT>
__global__ __launch_bounds__(256, (KT <= 2 ? 2 : 1))
void phase2_mid_kernel(const float* __restrict__ W, float* __restrict__ part, int R, int rows_per_wave, Phase2Mid a, CovMeta meta)
{
    constexpr int KP = 32 * KT;
    const int b = blockIdx.x;
    if (b < a.gram_blocks) { gram_block<KT>(W, part, R, rows_per_wave, b); return; }
    if (b == a.gram_blocks) {
        if (a.do_loss) loss_finalize_block(a.dotpart, a.ndot, a.WtW, a.HHt, a.stats, meta, a.nstat, KP, a.lam64, a.row);
        return;
    }
    if (a.do_b) b_update_block(a.Bold, a.Bnew, a.stats, a.HHt, meta, KP, a.eps);
}

// ----------------------------------------------------------------------------------------------
// ingest helpers
__global__ __launch_bounds__(256)
void transpose_kernel(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int64_t ld_dst,
                      int rows, int cols)                       // dst[c][r] = src[r][c]
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? src[(int64_t)r * ld_src + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < rows) dst[(int64_t)c * ld_dst + r] = tile[tx][j];
    }
}

__global__ __launch_bounds__(256)
void sqnorm_kernel(const float* __restrict__ x, int64_t n4, double* __restrict__ part)
{
    // part[block] = sum of squares; part[gridDim.x + block] = number of elements that are NOT exactly one bf16 plane (low 16
    // bits of the float32 non-zero): the census behind the choice of the x3 sweep's matrix instruction
    __shared__ double red[256];
    __shared__ double red2[256];
    double a = 0.0;
    unsigned long long multi = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        a += (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2] + (double)v[3] * v[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) multi += (__float_as_uint(v[e]) & 0xffffu) != 0u;
    }
    red[threadIdx.x] = a;
    red2[threadIdx.x] = (double)multi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { red[threadIdx.x] += red[threadIdx.x + s]; red2[threadIdx.x] += red2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { part[blockIdx.x] = red[0]; part[gridDim.x + blockIdx.x] = red2[0]; }
}

// pack / unpack between the caller's (row-major, unpadded) factors and the padded device layouts
__global__ void pad_rows_kernel(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int KP,
                                int64_t rows, int K)            // dst[r][k<K] = src[r][k]
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const int64_t r = i / K;
    const int k = (int)(i % K);
    dst[r * KP + k] = src[r * ld_src + k];
}
__global__ void unpad_rows_kernel(const float* __restrict__ src, int KP, float* __restrict__ dst, int64_t ld_dst,
                                  int64_t rows, int K)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const int64_t r = i / K;
    const int k = (int)(i % K);
    dst[r * ld_dst + k] = src[r * KP + k];
}

// ----------------------------------------------------------------------------------------------
// mini-batch views (main.py:509-521): dst[j][0:width] = j < n ? src[idx[j]][0:width] : 0 for j < n_pad
// (width multiple of 4), the column gather of Y, and the row scatter of the updated H (main.py:659-663; duplicate
// indices of a batch carry identical rows, so the write order does not matter).
__global__ __launch_bounds__(256)
void gather_rows_kernel(const float* __restrict__ src, int64_t ld_src, const int* __restrict__ idx, int n, int n_pad,
                        float* __restrict__ dst, int64_t ld_dst, int width)
{
    const int w4 = width / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)n_pad * w4; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i / w4), q = (int)(i % w4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (j < n) v = *reinterpret_cast<const f32x4*>(src + (int64_t)idx[j] * ld_src + 4 * q);
        *reinterpret_cast<f32x4*>(dst + (int64_t)j * ld_dst + 4 * q) = v;
    }
}
__global__ __launch_bounds__(256)
void gather_cols_kernel(const float* __restrict__ src, int64_t ld_src, const int* __restrict__ idx, int n, int n_pad,
                        float* __restrict__ dst, int64_t ld_dst, int rows)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)rows * n_pad; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / n_pad), j = (int)(i % n_pad);
        dst[(int64_t)r * ld_dst + j] = j < n ? src[(int64_t)r * ld_src + idx[j]] : 0.f;
    }
}
__global__ __launch_bounds__(256)
void scatter_rows_kernel(const float* __restrict__ src, int64_t ld_src, const int* __restrict__ idx, int n,
                         float* __restrict__ dst, int64_t ld_dst, int width)
{
    const int w4 = width / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)n * w4; i += (int64_t)gridDim.x * 256) {
        const int j = (int)(i / w4), q = (int)(i % w4);
        *reinterpret_cast<f32x4*>(dst + (int64_t)idx[j] * ld_dst + 4 * q) = *reinterpret_cast<const f32x4*>(src + (int64_t)j * ld_src + 4 * q);
    }
}

// ----------------------------------------------------------------------------------------------
// scaling, main.py:772-781
__global__ __launch_bounds__(256)
void colsum_part_kernel(const float* __restrict__ W, int KP, int G, int rows_per_block, double* __restrict__ part)
{
    const int k = threadIdx.x;
    if (k >= KP) return;
    const int g0 = blockIdx.x * rows_per_block, g1 = min(G, g0 + rows_per_block);
    double a = 0.0;
    for (int g = g0; g < g1; ++g) a += (double)W[(int64_t)g * KP + k];
    part[(int64_t)blockIdx.x * KP + k] = a;
}
__global__ __launch_bounds__(256)
void colsum_final_kernel(const double* __restrict__ part, int nblk, int KP, float* __restrict__ scale)
{
    const int k = threadIdx.x;
    if (k >= KP) return;
    double a = 0.0;
    for (int b = 0; b < nblk; ++b) a += part[(int64_t)b * KP + k];
    scale[k] = (float)a;
}
__global__ void scale_rows_kernel(float* __restrict__ A, int KP, int64_t rows, int K, const float* __restrict__ scale,
                                  int divide)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * K) return;
    const int64_t r = i / K;
    const int k = (int)(i % K);
    const float s = scale[k];
    float v = A[r * KP + k];
    A[r * KP + k] = divide ? v / s : v * s;
}
__global__ void scale_b_kernel(float* __restrict__ B, CovMeta meta, const float* __restrict__ scale)
{
    for (int i = 0; i < meta.n_cov; ++i)
        for (int idx = threadIdx.x; idx < meta.lev[i] * meta.k[i]; idx += blockDim.x)
            B[meta.boff[i] + idx] /= scale[meta.off[i] + idx % meta.k[i]];
}

// ----------------------------------------------------------------------------------------------
// direct-form ||X - W H||^2 in float64 (validation only).  Block = 256 genes x a range of cells;
// thread = gene, its W row lives in registers, H rows are broadcast from LDS.
constexpr int EV_CELLS = 32;

template <int KT>
__global__ __launch_bounds__(256)
void eval_recon_kernel(const float* __restrict__ Xng, int64_t ldG, const float* __restrict__ W,
                       const float* __restrict__ H, int G, int N, int cells_per_block, double* __restrict__ part)
{
    constexpr int KP = 32 * KT;
    __shared__ float hl[EV_CELLS][KP];
    __shared__ double red[256];
    const int t = threadIdx.x;
    const int g = blockIdx.x * 256 + t;
    float w[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) w[k] = (g < G) ? W[(int64_t)g * KP + k] : 0.f;
    const int n0 = blockIdx.y * cells_per_block, n1 = min(N, n0 + cells_per_block);
    double acc = 0.0;
    for (int nb = n0; nb < n1; nb += EV_CELLS) {
        __syncthreads();
        for (int idx = t; idx < EV_CELLS * KP; idx += 256) {
            const int r = idx / KP, k = idx % KP;
            hl[r][k] = (nb + r < n1) ? H[(int64_t)(nb + r) * KP + k] : 0.f;
        }
        __syncthreads();
        const int lim = min(EV_CELLS, n1 - nb);
        for (int r = 0; r < lim; ++r) {
            float p = 0.f;
#pragma unroll
            for (int k = 0; k < KP; ++k) p = fmaf(w[k], hl[r][k], p);
            if (g < G) {
                const double d = (double)Xng[(int64_t)(nb + r) * ldG + g] - (double)p;
                acc += d * d;
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
