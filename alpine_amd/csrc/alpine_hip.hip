// libalpine_hip.so -- C ABI (include/alpine_hip.h) over the gfx950 kernels in kernels.hpp.
// Host side of one shard (one GPU) of ALPINE's full-batch MU fit loop (alpine/main.py:486-676).
#include "../../include/alpine_hip.h"
#include "kernels.hpp"
#include "kernels_bf16.hpp"
#include "kernels_x3.hpp"
#include "kernels_wide.hpp"

#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace alpine;

static_assert(SG_MAX_CHAIN == ALPINE_MAX_ACCUMULATION_ROWS, "include/alpine_hip.h documents the accumulation cap");

static thread_local std::string g_create_error;

// The cells an iteration works on: the whole shard, or a gathered mini-batch (main.py:509-521).
struct CellView {
    float *Xgn = nullptr, *Xng = nullptr, *H = nullptr, *Y = nullptr;
    int N = 0;                // cells in the view
    int64_t Np = 0;           // padded cells = leading dimension of Xgn / Y, rows of Xng / H
    SweepGeom gA{}, gB{};
    int statBlocks = 0, gramBlocksH = 0;
};

struct alpine_ctx {
    // geometry
    int G = 0, N = 0, K = 0, KP = 0, KT = 0, n_cov = 0;
    // wide model (128 < K <= 1024, kernels_wide.hpp): NH = ceil(K / 128) halves, KP = 128 NH, KT = 4 = tiles per HALF, factors in the blocked layout [NH][rows][128]
    bool wide = false;
    int NH = 1;
    float *wide_den = nullptr, *wide_num = nullptr;     // [NH][wide_den_rows][128] product A.M of the updates; [NH][Np][128] transform numerator
    bool wide_one_pass = false;                         // x3 sweeps of a wide model: stream_gemm_x3w2_kernel (X read once per sweep) instead of one x3w launch per component half
    u32x4* wide_panel3 = nullptr;                       // x3: the panel of the sweep about to run as three exact bf16 planes, k-packed [3][rows / 8][256][8] (stream_gemm_x3w2_kernel)
    int64_t wide_den_rows = 0;
    int64_t Gp = 0, Np = 0;
    std::vector<int> cov_k, cov_lev;
    std::vector<double> lam;
    double orth = 0, alpha = 0, l1r = 0, eps = 0;
    int loss_type = 0;
    CovMeta meta{};
    int nstat = 0, nB = 0, nYrows = 0;
    int device = 0, n_cu = 256;
    struct GuardRec { void* user; size_t bytes; const char* name; };
    std::vector<GuardRec> guards;     // ALPINE_HIP_GUARD=1 (diagnostics): every device buffer sits between two pattern-filled zones, checked when it is freed
    size_t lds_max = 160 * 1024;      // LDS a workgroup may use (hipDeviceAttributeMaxSharedMemoryPerBlock); ALPINE_HIP_LDS_LIMIT lowers it for the H update's optional parts
    size_t lds_dev = 160 * 1024;      // ... the device's own figure (every other kernel's fixed LDS need is checked against this one)
    // stream
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // device buffers
    float *Xgn = nullptr, *Xng = nullptr, *W = nullptr, *H = nullptr, *Y = nullptr, *B[2] = {nullptr, nullptr};
    int bcur = 0;
    unsigned short *Xgn16 = nullptr, *Xng16 = nullptr, *Wp16 = nullptr, *Hp16 = nullptr;   // bf16 path (k-packed)
    unsigned short *Xgn16b = nullptr, *Xng16b = nullptr;     // second exact plane of X (split mode; freed if all zero)
    int64_t x_plane_gn = 0, x_plane_ng = 0;                  // element offsets plane 2 - plane 1 of the two copies
    bool bf16 = false;                // any bf16-pipe mode (rounded operands or exact split)
    bool split = false;               // exact-split mode
    int npx = 1;                      // bf16 planes of X in use (split mode: 1 or 2)
    int* xflags = nullptr;            // device: [0] some element of X not exact in the stored planes, [1] second plane in use
    float *piecesA = nullptr, *piecesB = nullptr;     // stream-K partial results of the two sweeps
    SweepGeom geomA{}, geomB{};
    CellView full{};                                  // the whole shard
    CellView batch_view{};                            // view of the batch opened by alpine_batch_begin
    bool batch_open = false;
    int64_t batch_n = 0;
    // mini-batch view buffers (alpine_config.batch_capacity > 0)
    int64_t batch_cap = 0;
    float *Xb_gn = nullptr, *Xb_ng = nullptr, *Hb = nullptr, *Yb = nullptr;
    int* idx_dev = nullptr;
    int slots = 512;
    int team_force = 0;               // alpine_debug_set_team_width: 0 = the library's own choice (team_width)
    bool x3 = false;                  // ALPINE_FLAG_X3_PRODUCTS in effect (float32 X, exact bf16 plane products)
    int64_t piecesA_cap = 0, piecesB_cap = 0;   // capacity of the pieces buffers, in floats (covers both tile widths of the x3 sweeps)
    int split_a_hint = 0, split_b_hint = 0;     // alpine_config.split_a / split_b
    int sweep_bf = SG_BLOCK_F;        // f columns per sweep workgroup tile (SweepGeom::bf of every geometry of this ctx)
    int sweep_waves = 4;              // waves per sweep workgroup (8 for the bf16 sweeps with K <= 64)
    float* red = nullptr;
    bool own_red = false;
    int64_t red_floats = 0, red_hht = 0, red_stats = 0;
    float *WtW = nullptr, *gramPart = nullptr;          // WtW = the CURRENT one of WtWbuf[2] (the fused phase 2 writes the other, then swaps)
    float* WtWbuf[2] = {nullptr, nullptr};
    bool fused_w = true;                               // env ALPINE_HIP_FUSED_W=0: W update and W^T W in separate launches (A/B)
    int gramBlocksH = 0, gramBlocksW = 0;
    float* statPart = nullptr;
    int statBlocks = 0;
    int64_t statPart_cap = 0, gramPart_cap = 0;        // capacities in blocks (a mini-batch view may be larger than the shard)
    // fused tail of the H update (MU branch, whole shard): partial blocks of H H^T and covariate statistics of the UPDATED H,
    // i.e. what the next phase 1 needs -> that phase 1 skips phase1_open_kernel while tail_valid
    float *gramPartH = nullptr, *statPartH = nullptr;
    int tail_blocks = 0;                               // grid of the fused H update (128 cells per block)
    bool tail_valid = false;
    bool no_tail = false;                              // env ALPINE_HIP_NO_TAIL=1: separate phase1_open_kernel every iteration (A/B)
    int* kind = nullptr;
    double *dotpart = nullptr, *lam_dev = nullptr, *loss_dev = nullptr, *f64part = nullptr;
    int ndot = 0;
    int64_t loss_cap = 0, loss_rows = 0, f64part_n = 0;
    float* scale = nullptr;
    float* stage = nullptr;           // staging for host uploads / factor packing
    int64_t stage_floats = 0;
    // state
    double xnorm2 = 0;
    std::vector<std::pair<int64_t, int64_t>> x_cover;   // disjoint, sorted cell intervals [a, b) uploaded so far
    bool x_plane2_dropped = false;    // split mode: alpine_finalize_X found the second plane all zero and freed it
    bool x_final = false, factors_set = false, pending_loss = false, loss_enabled = true;
    std::vector<bool> y_set;
    size_t bytes = 0;
    std::string err;
    // A/B knobs that change the launch structure, never the results; read from the environment ONCE, in alpine_create
    bool unfused_mid = false;         // env ALPINE_HIP_UNFUSED_MID=1: separate loss_finalize / b_update / gram launches (A/B)
    // timing-only ablations that produce WRONG results: compiled only into the diagnostics build (-DALPINE_DIAGNOSTICS,
    // libalpine_hip_diag.so, used by tools/); the production library ignores these environment variables
    bool ablate_stride0 = false;      // ALPINE_HIP_ABLATE_STRIDE0=1: the sweeps re-read row 0 of X (prices the HBM stream)
    bool ablate_panel = false;        // ALPINE_HIP_ABLATE_PANEL=1: bf16 sweeps re-read panel stage 0
    bool ablate_flush = false;        // ALPINE_HIP_ABLATE_FLUSH=1: x3 sweeps skip the accumulator flush
    int x3_ablate = 0;                // ALPINE_HIP_X3_ABLATE=1|2|3
    // multi-GPU: communicator over the shards of the cell axis (alpine_comm_init_rank); collectives run on `stream`
    ncclComm_t comm = nullptr;
    int comm_ranks = 1, comm_rank = 0;
    bool transform_only = false;
    bool use_als = false;
    bool tail_stats_per_cov = false;  // env ALPINE_HIP_TAIL_STATS=per_covariate: the fused tail's statistics one covariate at a time (A/B)
    bool no_guided_mfma = false;      // env ALPINE_HIP_GUIDED=scalar: the per-(covariate, class) scalar form of the guided terms instead of the MFMA products (A/B)
    GuidedRow* rowtab = nullptr;      // [32] stacked rows of Y -> covariate (guided terms on the MFMA, see GuidedRow)
    bool xcd_bias_auto = true;        // no ALPINE_HIP_XCD_BIAS in the environment: decided by the placement probe (create_impl)
    int xcc_of_wg0 = -1;              // placement probe (alpine_finalize_X): XCC id that workgroup 0 of a sweep launch ran on
    int* xcc_dev = nullptr;           // ... written here by that launch
    bool probe_placement = false;
    int xcd_bias_pm = 0;              // env ALPINE_HIP_XCD_BIAS (per mille): span length of even workgroups +bias, odd -bias (see SweepGeom::dL)
    int sg_variant = 0;               // env ALPINE_HIP_SG_VARIANT: pipeline shape of the sweep kernel (A/B experiments)
    int x3_variant = -1;              // env ALPINE_HIP_X3_VARIANT: 0 = 32x32x16 MFMA, 2 = 16x16x32 (x3w), unset = chosen from the data
    bool x3_wide = false;             // the sweeps use stream_gemm_x3w_kernel (decided in alpine_finalize_X)
    bool x3_narrow = false;           // 512-column workgroup tiles at K <= 64 (in effect)
    bool x3_narrow_pref = false;      // ... wanted for this shard size; alpine_finalize_X confirms it once the matrix instruction is known
    bool x3_narrow_forced = false;    // ... asked for explicitly (alpine_debug_set_option "x3_narrow"): kept whatever the matrix instruction
    double x_multi_plane_frac = 0;    // fraction of the elements of X that are not exactly one bf16 plane
    int x3_two_wave_opt = -1;         // alpine_debug_set_option "x3_two_wave": -1 = the library's choice, 0 = never, 1 = wherever the kernel exists
    bool x3_two_wave = false;         // the K in (64, 128] sweeps run stream_gemm_x3v_kernel (two waves per SIMD; decided in alpine_finalize_X)
    bool wide_two_wave = false;       // the one-pass sweep of a wide model runs its 8-wave form (one-plane data, K <= 160; decided in alpine_finalize_X)
    bool x_one_plane = false;         // ... and NONE is (census of alpine_finalize_X): the K > 64 sweeps then run the form without split and zero-plane test
    bool team_ok = true;              // teams pay for this ctx's data and model size (decided in alpine_finalize_X; see team_width)
    // profiling
    bool prof = false;
    int prof_every = 1;               // events bracket the launches of every prof_every-th phase 1 / iteration only
    int64_t prof_tick = 0;            // phase-1 counter; prof_now = prof && prof_tick % prof_every == 0
    bool prof_now = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[ALPINE_KERNEL_COUNT];
    size_t ev_used[ALPINE_KERNEL_COUNT] = {};
};

static int fail(alpine_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, expr)                                                                           \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail((c), e_ == hipErrorOutOfMemory ? ALPINE_ERR_OOM : ALPINE_ERR_HIP,         \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define NCCLCHK(c, expr)                                                                          \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail((c), ALPINE_ERR_RCCL, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// THE rule for every copy between the device and PAGEABLE host memory (caller arrays, std::vector, the stack) -- in BOTH directions:
// drain the ctx stream, then the BLOCKING hipMemcpy / hipMemcpy2D.  Never hipMemcpyAsync + hipStreamSynchronize: for pageable memory the
// runtime stages the bytes through a pinned buffer of its own, and the hop between that buffer and the user's memory is the runtime's
// business, not a stream operation.  What is ESTABLISHED (round 3, tests/fuzz_model_gpu.py seed 40885 repeated, gpurun_out/r04k-r04o):
// with device -> host copies issued as Async + synchronise the process died of heap corruption about once per 400 compute_loss calls
// (8-byte stray writes into freed chunks: the doubles of sum_f64_partials landing in a std::vector freed two lines later; numpy shape
// fields changing under the test); guard zones around every device buffer stayed clean, the device alone and the oracle alone survived
// thousands of repeats; replacing ONLY those copies by this rule ended it (8 x 400 + 4 065 + 2 919 cases without a crash).  What is
// INFERRED: that the late writer is the staging hop (no trace of the runtime's copy thread was taken).  The host -> device direction has
// shown no symptom -- there the hazard would be the runtime READING freed or reused host memory (wrong factors, not a crash) -- but it
// rests on the same assumption, so it follows the same rule.  All call sites go through these two helpers.
static int host_copy(alpine_ctx* c, void* dst, const void* src, size_t bytes, hipMemcpyKind kind)
{
    if (bytes == 0) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(dst, src, bytes, kind));
    return 0;
}
static int host_copy_2d(alpine_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind kind)
{
    if (width == 0 || height == 0) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy2D(dst, dpitch, src, spitch, width, height, kind));
    return 0;
}
#define HOSTCOPY(c, ...)    do { int rc_ = host_copy((c), __VA_ARGS__); if (rc_) return rc_; } while (0)
#define HOSTCOPY2D(c, ...)  do { int rc_ = host_copy_2d((c), __VA_ARGS__); if (rc_) return rc_; } while (0)

#define DISPATCH_KT(kt, CALL)                  \
    switch (kt) {                              \
        case 1: { constexpr int KT_ = 1; CALL; } break; \
        case 2: { constexpr int KT_ = 2; CALL; } break; \
        case 3: { constexpr int KT_ = 3; CALL; } break; \
        default: { constexpr int KT_ = 4; CALL; } break; \
    }

static int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }
// Environment variables.  The PRODUCTION library reads three, all at alpine_create / load time: ALPINE_HIP_GUARD (diagnostics allocator),
// ALPINE_HIP_LDS_LIMIT (lowers the LDS budget of the H update's optional parts: exercises its fall-backs) and ALPINE_HIP_XCD_BIAS
// (0 = equal spans = the bit-reproducible setting, see alpine_finalize_X).  Every knob that switches a kernel or a launch structure
// is an explicit call, alpine_debug_set_option(ctx, name, value) -- a stray variable in a user's environment cannot change which
// kernel runs.  Only the diagnostics build (-DALPINE_DIAGNOSTICS, tools/) also takes them from ALPINE_HIP_<NAME> variables.
#ifdef ALPINE_DIAGNOSTICS
static const char* knob_env(const char* name) { return std::getenv(name); }
#else
static const char* knob_env(const char*) { return nullptr; }
#endif
static bool knob_is(const char* name, char v) { const char* e = knob_env(name); return e && e[0] == v; }

// Diagnostics, ALPINE_HIP_GUARD=1: a device-side write past either end of a buffer does not fault when the neighbouring addresses are
// mapped (another buffer, or host memory the runtime keeps pinned) -- it corrupts silently.  In guard mode every buffer is allocated with
// a 4 KiB zone of 0xA5 on both sides; dev_free / alpine_destroy compare the zones and abort with the buffer's name when one was touched.
static const bool g_guard = std::getenv("ALPINE_HIP_GUARD") != nullptr && std::getenv("ALPINE_HIP_GUARD")[0] == '1';
constexpr size_t GUARD_BYTES = 4096;

static int dev_alloc(alpine_ctx* c, void** p, size_t bytes, bool zero = true, const char* name = "")
{
    if (bytes == 0) bytes = 16;
    const size_t want = g_guard ? ((bytes + 255) & ~(size_t)255) + 2 * GUARD_BYTES : bytes;
    {
        const hipError_t e = hipMalloc(p, want);
        if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation) {
            (void)hipGetLastError();                      // the failed allocation must not surface again at the next check
            size_t free_b = 0, total_b = 0;
            (void)hipMemGetInfo(&free_b, &total_b);
            *p = nullptr;
            return fail(c, ALPINE_ERR_OOM, "device %d is out of memory: %.2f GiB requested on top of the %.2f GiB this ctx already holds (%.2f of %.2f GiB free); "
                        "the float32 storage keeps TWO copies of X (genes x cells and cells x genes): shard the cell axis over more GPUs, or use the "
                        "bf16-plane storage (half the bytes) where X allows it", c->device, (double)bytes / 1073741824.0, (double)c->bytes / 1073741824.0,
                        (double)free_b / 1073741824.0, (double)total_b / 1073741824.0);
        }
        HIPCHK(c, e);
    }
    c->bytes += bytes;
    if (g_guard) {
        char* base = static_cast<char*>(*p);
        HIPCHK(c, hipMemsetAsync(base, 0xA5, want, c->stream));
        *p = base + GUARD_BYTES;
        c->guards.push_back({*p, bytes, name});
    }
    if (zero) HIPCHK(c, hipMemsetAsync(*p, 0, bytes, c->stream));
    return 0;
}

// guard mode: compare the two zones of the buffer `user` (abort with its name if one was written); returns the pointer hipFree wants
static void* guard_check(alpine_ctx* c, void* user, bool forget)
{
    if (!g_guard || !user) return user;
    for (size_t i = 0; i < c->guards.size(); ++i) {
        if (c->guards[i].user != user) continue;
        const alpine_ctx::GuardRec r = c->guards[i];
        char* base = static_cast<char*>(user) - GUARD_BYTES;
        const size_t tail = ((r.bytes + 255) & ~(size_t)255) - r.bytes + GUARD_BYTES;
        std::vector<unsigned char> lo(GUARD_BYTES), hi(tail);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(lo.data(), base, GUARD_BYTES, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hi.data(), static_cast<char*>(user) + r.bytes, tail, hipMemcpyDeviceToHost);
        for (size_t k = 0; k < GUARD_BYTES; ++k)
            if (lo[k] != 0xA5) { std::fprintf(stderr, "alpine guard: buffer %s (%zu bytes) was written %zu bytes BEFORE its start\n", r.name, r.bytes, GUARD_BYTES - k); std::abort(); }
        for (size_t k = 0; k < tail; ++k)
            if (hi[k] != 0xA5) { std::fprintf(stderr, "alpine guard: buffer %s (%zu bytes) was written %zu bytes PAST its end\n", r.name, r.bytes, k); std::abort(); }
        if (forget) c->guards.erase(c->guards.begin() + (long)i);
        return base;
    }
    return user;
}

static hipError_t dev_free(alpine_ctx* c, void* user) { return hipFree(guard_check(c, user, true)); }

#define ALLOC(c, ptr, T, count) \
    do { int rc_ = dev_alloc((c), reinterpret_cast<void**>(&(ptr)), sizeof(T) * (size_t)(count), true, #ptr); if (rc_) return rc_; } while (0)

// ---------------------------------------------------------------------------------- geometry
struct Geometry {
    int K, KP, KT, nstat, nB, nYrows;
    int64_t Gp, Np, red_floats, red_hht, red_stats;
};

static int geometry(const alpine_config* cfg, Geometry* g, std::string* why)
{
    if (!cfg || cfg->struct_size != (int32_t)sizeof(alpine_config)) { *why = "alpine_config.struct_size mismatch"; return -1; }
    if (cfg->n_genes <= 0 || cfg->n_cells <= 0) { *why = "n_genes and n_cells must be positive"; return -1; }
    if (cfg->batch_capacity < 0) { *why = "batch_capacity must be >= 0"; return -1; }
    if (cfg->n_genes > (1 << 30) || cfg->n_cells > (1 << 30)) { *why = "n_genes / n_cells too large for this build"; return -1; }
    if (cfg->n_components <= 0) { *why = "n_components must be greater than 0."; return -1; }
    if (cfg->n_covariates < 0 || cfg->n_covariates > MAX_COV) { *why = "n_covariates out of range (0..16)"; return -1; }
    if (cfg->n_covariates > 0 && (!cfg->cov_components || !cfg->cov_levels || !cfg->lam)) { *why = "covariate arrays are NULL"; return -1; }
    int K = cfg->n_components, nstat = 0, nB = 0, nY = 0;
    for (int i = 0; i < cfg->n_covariates; ++i) {
        const int k = cfg->cov_components[i], C = cfg->cov_levels[i];
        // k = 0 is the reference's "covariate without guided components" (main.py:335 rejects only n < 0): it only adds its loss column
        if (k < 0 || k > MAX_COV_K) { *why = "a covariate may have 0..64 guided components in this build"; return -1; }
        if (C <= 0) { *why = "each covariate needs at least one level"; return -1; }
        if (!(cfg->lam[i] >= 0)) { *why = "Each element in lam must be a non-negative float."; return -1; }
        K += k;
        nstat += C * k + k + 2;
        nB += C * k;
        nY += C;
    }
    if (K > WIDE_MAX_NH * WIDE_KH) { *why = "total components > 1024 not supported by this build"; return -1; }
    if (K > 128) {
        int guided = 0;
        for (int i = 0; i < cfg->n_covariates; ++i) guided += cfg->cov_components[i];
        if (guided > 128) { *why = "with more than 128 components in total the guided ones must fit in the first 128 columns (sum k_i <= 128)"; return -1; }
        if (cfg->flags & (ALPINE_FLAG_X_BF16 | ALPINE_FLAG_X_SPLIT)) { *why = "more than 128 components need the float32 storage (x3 or f32 sweeps)"; return -1; }
    }
    if (cfg->loss_type != ALPINE_LOSS_KL && cfg->loss_type != ALPINE_LOSS_FROBENIUS) { *why = "loss_type must be one of ['kl-divergence', 'frobenius']."; return -1; }
    if (!(cfg->eps >= 0) || !(cfg->alpha_W >= 0) || !(cfg->orth_W >= 0) || !(cfg->l1_ratio_W >= 0 && cfg->l1_ratio_W <= 1)) { *why = "eps/alpha_W/orth_W must be >= 0 and l1_ratio_W in [0,1]"; return -1; }
    g->K = K;
    g->KT = K > 128 ? WIDE_KT : (K + 31) / 32;          // wide: tiles per half
    g->KP = K > 128 ? (int)round_up(K, WIDE_KH) : 32 * g->KT;         // wide: NH = KP / 128 halves
    g->nstat = nstat; g->nB = nB; g->nYrows = nY;
    g->Gp = round_up(cfg->n_genes, 128);
    g->Np = round_up(cfg->n_cells, 128);
    g->red_hht = g->Gp * g->KP;
    g->red_stats = g->red_hht + (int64_t)g->KP * g->KP;
    g->red_floats = round_up(g->red_stats + nstat + 2, 4);
    return 0;
}

extern "C" int64_t alpine_reduce_block_floats(const alpine_config* cfg)
{
    Geometry g; std::string why;
    if (geometry(cfg, &g, &why)) { g_create_error = why; return ALPINE_ERR_BAD_ARG; }
    return g.red_floats;
}

constexpr int XCD_BIAS_MAG = 40;      // per mille: what the placement probe applies (alpine_finalize_X); more stops paying, see DESIGN.md 4.2c
static inline int sweep_grid(const SweepGeom& g) { return sg_grid(g); }

// Team width of an x3 sweep over F columns with workgroup tiles of bf_wg columns (SweepGeom::gw): the widest team of {8, 4, 2} whose
// tiles leave at most 2.5 % of the team members without columns (the last team tile is padded to gw workgroup tiles); 1 for every
// other sweep kernel, for forced divisions (a test knob that counts WORKGROUPS per tile) and when the grid cannot be dealt to the
// 8 XCDs in whole teams.  team_force > 0 (alpine_debug_set_team_width): that width whenever it is admissible.
// Measured with tools/x3w_bench (variants interleaved, profiles/r04/x3w_bench_*.txt): K = 105 on one-plane data, teams of 8 together with
// the one-plane kernel 1.74 -> 1.54 ms per sweep (either alone: 0 - 2 %: the sweep sat on the fabric limit AND on its issue limit);
// K = 60 on count data -3 % (W^TX, 8) / -4 % (XH^T, 4); K <= 64 on full significands teams LOSE 2 - 15 % (there the chip is at its
// power limit and lock-step neighbours make it worse), K = 105 on full significands +-1 %.  Hence team_ok (alpine_finalize_X).
static int team_width(const alpine_ctx* c, int64_t F, int bf_wg, int forced);
static SweepGeom make_geom(const alpine_ctx* c, int64_t F, int64_t R, int forced, int bf_wg, int bias_pm = 0, int gw = 0);

static int gram_rows_per_wave(int64_t R, int n_cu)
{
    // enough blocks to occupy the chip for small matrices (W: 20k rows), at most GR_ROWS_PER_WAVE rows per wave
    int64_t rpw = round_up(std::max<int64_t>(16, R / (4 * (int64_t)n_cu)), 16);
    return (int)std::min<int64_t>(GR_ROWS_PER_WAVE, rpw);
}

// (re)computes the stream-K geometry of the two sweeps over the whole shard for workgroup tiles of bf columns
static void apply_sweep_geometry(alpine_ctx* c, int bf)
{
    c->sweep_bf = bf;
    c->geomA = make_geom(c, c->Gp, c->Np, c->split_a_hint, bf, c->xcd_bias_pm);       // XH^T: f = genes, r = cells
    c->geomB = make_geom(c, c->Np, c->Gp, c->split_b_hint, bf, c->xcd_bias_pm);       // W^TX: f = cells, r = genes
    c->full.gA = c->geomA; c->full.gB = c->geomB;
}

static int team_width(const alpine_ctx* c, int64_t F, int bf_wg, int forced)
{
    if (!(c->x3 || c->bf16) || forced > 0) return 1;
    if (c->team_force == 0 && !c->team_ok) return 1;
    const int64_t tiles = (F + bf_wg - 1) / bf_wg;
    auto admissible = [&](int gw) { return gw >= 1 && c->slots % (8 * gw) == 0; };
    if (c->team_force > 0) return admissible(c->team_force) ? c->team_force : 1;
    for (int gw : {8, 4, 2}) {
        if (!admissible(gw)) continue;
        const int64_t padded = (tiles + gw - 1) / gw * gw;
        if ((padded - tiles) * 40 <= padded) return gw;
    }
    return 1;
}

static SweepGeom make_geom(const alpine_ctx* c, int64_t F, int64_t R, int forced, int bf_wg, int bias_pm, int gw)
{
    if (gw <= 0) gw = team_width(c, F, bf_wg, forced);
    else if (!(c->x3 || c->bf16) || forced > 0 || c->slots % (8 * gw) != 0) gw = 1;
    // K > 64 in teams: the even/odd bias the placement probe chose for the team-less division costs 2 % there (tools/team_ab.py, one
    // engine, interleaved: K = 105 at 125 000 cells 3.630 ms per iteration with -40 per mille, 3.547 without; at K = 60 the bias still
    // pays with teams: 5.284 vs 5.340 ms) -- an explicitly requested bias (environment, alpine_debug_set_xcd_bias) is left alone
    if (gw > 1 && c->KT >= 3 && c->xcd_bias_auto) bias_pm = 0;
    return sg_make_geom(F, R, std::max(1, c->slots / gw), forced, bf_wg * gw, bias_pm, gw);
}

// ---------------------------------------------------------------------------------- create
static int create_impl(alpine_ctx* c, const alpine_config* cfg, const Geometry& g)
{
    c->G = (int)cfg->n_genes; c->N = (int)cfg->n_cells;
    c->K = g.K; c->KP = g.KP; c->KT = g.KT; c->Gp = g.Gp; c->Np = g.Np;
    c->wide = g.K > 128;
    c->NH = c->wide ? g.KP / WIDE_KH : 1;
    c->n_cov = cfg->n_covariates;
    c->nstat = g.nstat; c->nB = g.nB; c->nYrows = g.nYrows;
    c->orth = cfg->orth_W; c->alpha = cfg->alpha_W; c->l1r = cfg->l1_ratio_W; c->eps = cfg->eps;
    c->loss_type = cfg->loss_type;
    c->transform_only = (cfg->flags & ALPINE_FLAG_TRANSFORM_ONLY) != 0;
    c->split = (cfg->flags & ALPINE_FLAG_X_SPLIT) != 0;
    c->bf16 = (cfg->flags & ALPINE_FLAG_X_BF16) != 0 || c->split;
    c->npx = c->split ? 2 : 1;        // two planes until alpine_finalize_X knows whether the second is needed
    c->use_als = (cfg->flags & ALPINE_FLAG_USE_ALS) != 0;
    c->x3 = (cfg->flags & ALPINE_FLAG_X3_PRODUCTS) != 0 && !c->bf16;
    c->device = cfg->device_id;
    HIPCHK(c, hipSetDevice(c->device));
    hipDeviceProp_t prop;
    HIPCHK(c, hipGetDeviceProperties(&prop, c->device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(c, ALPINE_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 (MI355X) only", c->device, prop.gcnArchName);
    c->n_cu = prop.multiProcessorCount;
    {
        int lds = 0;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) == hipSuccess && (size_t)lds > c->lds_max) c->lds_max = (size_t)lds;   // gfx950: 160 KiB per CU
        c->lds_dev = c->lds_max;
        if (const char* e = std::getenv("ALPINE_HIP_LDS_LIMIT")) { const long v = std::atol(e); if (v > 0 && (size_t)v < c->lds_max) c->lds_max = (size_t)v; }   // tests: exercise the fall-backs
    }
    c->unfused_mid = knob_is("ALPINE_HIP_UNFUSED_MID", '1');
    c->no_tail = knob_is("ALPINE_HIP_NO_TAIL", '1') || c->unfused_mid;
    c->fused_w = !knob_is("ALPINE_HIP_FUSED_W", '0') && !c->unfused_mid;
    c->ablate_stride0 = knob_is("ALPINE_HIP_ABLATE_STRIDE0", '1');
    c->ablate_panel = knob_is("ALPINE_HIP_ABLATE_PANEL", '1');
    c->ablate_flush = knob_is("ALPINE_HIP_ABLATE_FLUSH", '1');
    if (const char* e = knob_env("ALPINE_HIP_X3_ABLATE")) c->x3_ablate = std::atoi(e);
    c->tail_stats_per_cov = knob_is("ALPINE_HIP_TAIL_STATS", 'p');
    if (const char* e = knob_env("ALPINE_HIP_GUIDED")) c->no_guided_mfma = (std::strcmp(e, "scalar") == 0);
    if (const char* e = std::getenv("ALPINE_HIP_XCD_BIAS")) { c->xcd_bias_pm = std::max(-200, std::min(200, std::atoi(e))); c->xcd_bias_auto = false; }
    if (const char* e = knob_env("ALPINE_HIP_SG_VARIANT")) c->sg_variant = std::atoi(e);
    if (const char* e = knob_env("ALPINE_HIP_X3_VARIANT")) c->x3_variant = std::atoi(e);
    if (cfg->stream) { c->stream = (hipStream_t)cfg->stream; c->own_stream = false; }
    else { HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }

    // covariate metadata
    c->meta.n_cov = c->n_cov; c->meta.loss_type = c->loss_type;
    int off = 0, boff = 0, yoff = 0, soff = 0;
    std::vector<int> kind(std::max(1, c->nstat), 0);
    for (int i = 0; i < c->n_cov; ++i) {
        const int k = cfg->cov_components[i], C = cfg->cov_levels[i];
        c->cov_k.push_back(k); c->cov_lev.push_back(C); c->lam.push_back(cfg->lam[i]);
        c->meta.off[i] = off; c->meta.k[i] = k; c->meta.lev[i] = C;
        c->meta.boff[i] = boff; c->meta.yoff[i] = yoff; c->meta.soff[i] = soff;
        c->meta.lam[i] = (float)cfg->lam[i]; c->meta.lam2[i] = (float)(2.0 * cfg->lam[i]);
        kind[soff + C * k + k] = 1; kind[soff + C * k + k + 1] = 2;
        off += k; boff += C * k; yoff += C; soff += C * k + k + 2;
    }
    c->y_set.assign(c->n_cov, false);
    {
        GuidedRow rows[32];
        for (int r = 0; r < 32; ++r) {
            rows[r] = GuidedRow{0, 0, 0, 0.f};                          // rows past the last class: no columns
            for (int i = 0; i < c->n_cov; ++i) {
                const int rr = r - c->meta.yoff[i];
                if (rr >= 0 && rr < c->meta.lev[i])
                    rows[r] = GuidedRow{c->meta.off[i], c->meta.k[i], c->meta.boff[i] + rr * c->meta.k[i],
                                        c->loss_type == ALPINE_LOSS_KL ? c->meta.lam[i] : c->meta.lam2[i]};
            }
        }
        ALLOC(c, c->xcc_dev, int, 4);
        ALLOC(c, c->rowtab, GuidedRow, 32);
        HOSTCOPY(c, c->rowtab, rows, sizeof rows, hipMemcpyHostToDevice);
    }

    const int64_t Gp = c->Gp, Np = c->Np; const int KP = c->KP;
    // sweeps
    // bf16 sweeps with K <= 64: 8-wave workgroups (1024-column tiles, one per CU); everything else 4 waves x 512 columns
    c->sweep_waves = (c->bf16 && c->KT <= 2 && !knob_is("ALPINE_HIP_BF16_WAVES", '4')) ? 8 : 4;   // (diagnostics build: A/B of the workgroup shape, same results)
    int slots = c->n_cu * (c->KT <= 2 && c->sweep_waves == 4 && !c->x3 ? 2 : 1);   // resident workgroups: x3 and 8-wave bf16 run one per CU
    // experiment knob (same results): K <= 32 x3 kernels fit two waves per SIMD (<= 256 registers): ALPINE_HIP_X3_SLOTS=2 gives them
    // two workgroups per CU -- the one data point available for "would a second wave per SIMD help the memory-bound sweep?"
    if (c->x3 && c->KT == 1 && knob_is("ALPINE_HIP_X3_SLOTS", '2')) slots = 2 * c->n_cu;
    c->slots = slots;
    // K <= 64: 1024-column workgroup tiles (a wave owns 256 columns) -- except for small shards on the 32x32x16 form, where the
    // piece traffic (every workgroup flushes bf x KP accumulators whatever the shard size: 67 MB per sweep at 1024 columns,
    // written and read back) outweighs the doubled panel re-reads of 512-column tiles.  Interleaved A/B, ms per iteration, 512 vs
    // 1024 columns: 25 000 cells 0.7436 vs 0.7510, 50 000 cells 1.360 vs 1.399, 100 000 cells 2.680 vs 2.640, 200 000 cells
    // +0.8 %; on the 16x16x32 form (x3w) 512 columns LOSE at 25 000 cells (0.794 vs 0.779).  So: 512 columns for small shards
    // unless alpine_finalize_X selects x3w.  Round 4, with teams (the panel of a 512-column tile is shared by its team through L2, which
    // takes most of the narrow tiles' extra panel traffic away; tools/option_ab.py, two engines per form, profiles/r04/narrow_tiles_ab.txt):
    // 25 000 cells 0.707 / 0.720 vs 0.757 / 0.757 ms, 50 000 1.330 / 1.361 vs 1.422 / 1.383, 100 000 2.605 / 2.596 vs 2.687 / 2.650,
    // 200 000 5.220 / 5.224 vs 5.103 / 5.183: the limit moved from 65 536 to 131 072 cells.  Option "x3_narrow" forces one form.
    if (const char* e = knob_env("ALPINE_HIP_X3_NARROW")) { c->x3_narrow_pref = (e[0] == '1'); c->x3_narrow_forced = true; }
    else c->x3_narrow_pref = cfg->n_cells <= 131072;
    if (c->x3_ablate || !c->x3 || c->KT > 2) c->x3_narrow_pref = false;   // (the diagnostics build's ablated kernels are 1024-column only)
    c->batch_cap = cfg->batch_capacity;
    c->split_a_hint = cfg->split_a; c->split_b_hint = cfg->split_b;
    // pieces: nwg * maxp tiles of bf x KP floats; mini-batch views have their own geometry (alpine_batch_begin), one per
    // view size.  Sized for every tile width and every view size this ctx may use.
    auto piece_floats = [&](int bf, int64_t* capA, int64_t* capB) {
        // every even/odd bias the ctx may end up with (the placement probe of alpine_finalize_X picks -+XCD_BIAS_MAG; a bias moves the
        // last share's boundary AND can add a piece per span: maxp depends on the longer span)
        // ... and every team width (the library's own choice = 0, and the widths alpine_debug_set_team_width may ask for)
        int64_t ta = 0, tb = 0;
        auto tiles = [](const SweepGeom& g) { return (int64_t)g.nwg * g.maxp * g.bf; };
        for (int gw : {0, 1, 2, 4, 8}) {
            for (int bias : {c->xcd_bias_pm, 0, -XCD_BIAS_MAG, XCD_BIAS_MAG}) {
                ta = std::max(ta, tiles(make_geom(c, Gp, Np, cfg->split_a, bf, bias, gw)));
                tb = std::max(tb, tiles(make_geom(c, Np, Gp, cfg->split_b, bf, bias, gw)));
            }
            if (c->batch_cap > 0) {
                const int64_t Bp_max = round_up(std::min<int64_t>(c->batch_cap, (int64_t)1 << 30), 128);
                for (int64_t Bp = 128; Bp <= Bp_max; Bp += 128) {
                    ta = std::max(ta, tiles(make_geom(c, Gp, Bp, 0, bf, 0, gw)));
                    tb = std::max(tb, tiles(make_geom(c, Bp, Gp, 0, bf, 0, gw)));
                }
            }
        }
        *capA = std::max(*capA, ta * KP); *capB = std::max(*capB, tb * KP);          // (wide: NH buffers of [.][128], cap / NH each)
    };
    // (wide models on the x3 sweeps: the one-pass kernel stream_gemm_x3w2_kernel, a wave owns 64 columns x all 256 components -- up to
    // 224 components: with all 16 component tiles its 256 accumulators + X ring + planes no longer fit the register file (68 - 248 B of
    // scratch per lane in the hot loop: 41 - 70 it/s at K = 256 against 69 on two passes), so K > 224 stays on the two-pass form)
    // (the kernels that address X as a scalar row base + a 32-bit lane offset -- x3w2, x3v -- need 24 rows of the longer axis to fit 2^32 bytes:
    // shards of more than 2^25 cells (or genes) stay on the kernels with 64-bit lane addresses)
    const bool lane32_ok = std::max(Gp, Np) <= ((int64_t)1 << 25);
    c->wide_one_pass = c->wide && c->x3 && !c->x3_ablate && c->K <= 224 && lane32_ok;        // (K > 256: one launch per half)
    const int bf_default = c->x3 ? (c->wide_one_pass ? 256 : (c->KT <= 2 ? 1024 : 512)) : c->sweep_waves * SG_WAVE_F;
    piece_floats(bf_default, &c->piecesA_cap, &c->piecesB_cap);
    if (c->wide && c->x3 && !c->x3_ablate) piece_floats(c->wide_one_pass ? 512 : 256, &c->piecesA_cap, &c->piecesB_cap);    // (alpine_debug_set_option "wide_one_pass")
    if (c->x3 && c->KT <= 2 && !c->x3_ablate) piece_floats(512, &c->piecesA_cap, &c->piecesB_cap);     // (alpine_debug_set_option "x3_narrow" may ask for it later)
    c->x3_narrow = c->x3_narrow_pref;                 // until alpine_finalize_X knows the data
    apply_sweep_geometry(c, c->x3_narrow ? 512 : bf_default);

    ALLOC(c, c->Xgn, float, c->bf16 ? 4 : Gp * Np);
    ALLOC(c, c->Xng, float, (c->transform_only || c->bf16) ? 4 : Np * Gp);
    if (c->bf16) {
        ALLOC(c, c->Xgn16, unsigned short, Gp * Np);
        ALLOC(c, c->Xng16, unsigned short, c->transform_only ? 8 : Np * Gp);
        if (c->split) {                                      // second plane in its own allocation so that it can be dropped
            ALLOC(c, c->Xgn16b, unsigned short, Gp * Np);
            ALLOC(c, c->Xng16b, unsigned short, c->transform_only ? 8 : Np * Gp);
            c->x_plane_gn = c->Xgn16b - c->Xgn16;
            c->x_plane_ng = c->Xng16b - c->Xng16;
        }
        // rounded mode: bf16 operand copies of the panels (the split forms read the float32 masters)
        ALLOC(c, c->Wp16, unsigned short, c->split ? 8 : Gp * KP);
        ALLOC(c, c->Hp16, unsigned short, c->split ? 8 : Np * KP);
        ALLOC(c, c->xflags, int, 4);
    }
    ALLOC(c, c->W, float, Gp * KP);
    ALLOC(c, c->H, float, Np * KP);
    if (c->wide) {
        // den = A . M for the rows of W, of H, or of a mini-batch view of H -- which may hold MORE cells than the shard (draws with
        // replacement, or a global batch whose cells all fall into this rank's block): sized for the largest of the three
        c->wide_den_rows = std::max(std::max(Gp, Np), round_up(std::max<int64_t>(c->N, cfg->batch_capacity), 128));
        ALLOC(c, c->wide_den, float, c->wide_den_rows * KP);
        ALLOC(c, c->wide_num, float, Np * KP);
        if (c->x3) ALLOC(c, c->wide_panel3, u32x4, 3 * (c->wide_den_rows / 8) * KP);       // 6 bytes per element of the largest panel (W, H or a view of H)
    }
    ALLOC(c, c->Y, float, (int64_t)std::max(1, c->nYrows) * Np);
    ALLOC(c, c->B[0], float, std::max(1, c->nB));
    ALLOC(c, c->B[1], float, std::max(1, c->nB));
    ALLOC(c, c->piecesA, float, c->transform_only ? 4 : c->piecesA_cap);
    ALLOC(c, c->piecesB, float, c->piecesB_cap);
    c->red_floats = g.red_floats; c->red_hht = g.red_hht; c->red_stats = g.red_stats;
    if (cfg->reduce_block) { c->red = (float*)cfg->reduce_block; c->own_red = false; HIPCHK(c, hipMemsetAsync(c->red, 0, sizeof(float) * c->red_floats, c->stream)); }
    else { ALLOC(c, c->red, float, c->red_floats); c->own_red = true; }
    ALLOC(c, c->WtWbuf[0], float, KP * KP);
    ALLOC(c, c->WtWbuf[1], float, KP * KP);
    c->WtW = c->WtWbuf[0];
    const int rows_per_gram_block = 4 * GR_ROWS_PER_WAVE;
    c->gramBlocksH = (int)((Np + rows_per_gram_block - 1) / rows_per_gram_block);
    c->gramBlocksW = (int)((Gp + rows_per_gram_block - 1) / rows_per_gram_block);
    // a mini-batch view holds up to batch_capacity cells, which may exceed the shard (draws with replacement, or a
    // global batch whose cells all fall into this rank's block): the per-block partial buffers are sized for both
    const int64_t view_cells_max = std::max<int64_t>(c->N, cfg->batch_capacity);
    {
        auto nblk = [&](int64_t R) { const int rpw = gram_rows_per_wave(R, c->n_cu); return (R + 4 * rpw - 1) / (4 * rpw); };
        c->gramPart_cap = std::max(std::max(nblk(Np), nblk(Gp)), nblk(round_up(view_cells_max, 128)));
        c->gramPart_cap = std::max<int64_t>(c->gramPart_cap, Gp / 128);       // the fused W update writes one block per 128 genes
        // nblk() is not monotone in R (rows per wave grow with R): cover every view size up to the maximum
        for (int64_t R = 128; R <= round_up(view_cells_max, 128) && R <= (int64_t)4 * GR_ROWS_PER_WAVE * 4 * c->n_cu; R += 128)
            c->gramPart_cap = std::max(c->gramPart_cap, nblk(R));
        ALLOC(c, c->gramPart, float, c->gramPart_cap * KP * KP);
    }
    c->statBlocks = (int)((c->N + HS_CELLS - 1) / HS_CELLS);
    c->statPart_cap = (view_cells_max + HS_CELLS - 1) / HS_CELLS;
    ALLOC(c, c->statPart, float, c->statPart_cap * std::max(1, c->nstat));
    {
        // fused tail: one partial H H^T block and one statistics row per 128-cell block of the H update (N x 2 KP bytes)
        c->tail_blocks = (int)((c->N + HS_CELLS - 1) / HS_CELLS);
        // (the blocked path of wide models has no fused tail: 4 MB per 128 cells at K = 1024 would be the largest allocation of the ctx)
        ALLOC(c, c->gramPartH, float, c->wide ? 4 : (int64_t)c->tail_blocks * KP * KP);
        ALLOC(c, c->statPartH, float, c->wide ? 4 : (int64_t)c->tail_blocks * std::max(1, c->nstat));
    }
    ALLOC(c, c->kind, int, kind.size());
    HOSTCOPY(c, c->kind, kind.data(), sizeof(int) * kind.size(), hipMemcpyHostToDevice);
    c->ndot = (int)((c->G + UPD_ROWS * 4 - 1) / (UPD_ROWS * 4)) * 4;
    ALLOC(c, c->dotpart, double, c->ndot);
    ALLOC(c, c->lam_dev, double, std::max(1, c->n_cov));
    if (c->n_cov) HOSTCOPY(c, c->lam_dev, c->lam.data(), sizeof(double) * c->n_cov, hipMemcpyHostToDevice);
    c->loss_cap = 1024;
    ALLOC(c, c->loss_dev, double, c->loss_cap * (c->n_cov + 2));
    c->f64part_n = 65536;
    ALLOC(c, c->f64part, double, c->f64part_n);
    ALLOC(c, c->scale, float, KP);
    c->stage_floats = std::max<int64_t>((int64_t)1 << 26, std::max<int64_t>((int64_t)c->K * Np, Gp * (int64_t)KP));   // >= 256 MiB
    ALLOC(c, c->stage, float, c->stage_floats);
    c->full.Xgn = c->Xgn; c->full.Xng = c->Xng; c->full.H = c->H; c->full.Y = c->Y;
    c->full.N = c->N; c->full.Np = Np; c->full.gA = c->geomA; c->full.gB = c->geomB;
    c->full.statBlocks = c->statBlocks; c->full.gramBlocksH = c->gramBlocksH;
    if (c->batch_cap > 0) {
        if (c->bf16 || c->transform_only) return fail(c, ALPINE_ERR_UNSUPPORTED, "batch_capacity needs the float32 two-copy layout");
        const int64_t Bp = round_up(std::min<int64_t>(c->batch_cap, (int64_t)1 << 30), 128);
        ALLOC(c, c->Xb_gn, float, Gp * Bp);
        ALLOC(c, c->Xb_ng, float, Bp * Gp);
        ALLOC(c, c->Hb, float, Bp * KP);
        ALLOC(c, c->Yb, float, (int64_t)std::max(1, c->nYrows) * Bp);
        ALLOC(c, c->idx_dev, int, Bp);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int alpine_create(const alpine_config* cfg, alpine_ctx** out)
{
    if (!out) return fail(nullptr, ALPINE_ERR_BAD_ARG, "out is NULL");
    *out = nullptr;
    Geometry g; std::string why;
    if (geometry(cfg, &g, &why)) return fail(nullptr, ALPINE_ERR_BAD_ARG, "%s", why.c_str());
    alpine_ctx* c = new (std::nothrow) alpine_ctx();
    if (!c) return fail(nullptr, ALPINE_ERR_OOM, "host allocation failed");
    const int rc = create_impl(c, cfg, g);
    if (rc) { g_create_error = c->err; alpine_destroy(c); return rc; }
    *out = c;
    return 0;
}

extern "C" int alpine_destroy(alpine_ctx* c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    void* ptrs[] = {c->Xgn, c->Xng, c->W, c->H, c->Y, c->B[0], c->B[1], c->piecesA, c->piecesB, c->own_red ? c->red : nullptr,
                    c->WtWbuf[0], c->WtWbuf[1], c->Xgn16, c->Xng16, c->Xgn16b, c->Xng16b, c->Wp16, c->Hp16, c->xflags, c->Xb_gn, c->Xb_ng, c->Hb, c->Yb, c->idx_dev, c->gramPart, c->statPart, c->gramPartH, c->statPartH, c->rowtab, c->xcc_dev, c->wide_den, c->wide_num, c->wide_panel3, c->kind, c->dotpart, c->lam_dev, c->loss_dev, c->f64part, c->scale, c->stage};
    for (void* p : ptrs) if (p) (void)dev_free(c, p);
    for (auto& v : c->ev) for (auto& pr : v) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" const char* alpine_last_error(const alpine_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

extern "C" int alpine_get_info(alpine_ctx* c, alpine_info* info)
{
    if (!c || !info) return fail(c, ALPINE_ERR_BAD_ARG, "NULL argument");
    info->abi_version = ALPINE_HIP_ABI_VERSION;
    info->k_total = c->K; info->k_padded = c->KP;
    info->split_a = c->geomA.maxp; info->split_b = c->geomB.maxp; info->grid_a = sweep_grid(c->geomA); info->grid_b = sweep_grid(c->geomB);
    info->genes_padded = c->Gp; info->cells_padded = c->Np;
    info->reduce_block_floats = c->red_floats;
    info->device_bytes = (int64_t)c->bytes;
    info->x_sqnorm = c->xnorm2;
    info->x_multi_plane_fraction = c->x_multi_plane_frac;
    info->x3_wide = c->x3 && c->x3_wide ? 1 : 0;
    info->sweep_waves_per_simd = !c->x3 ? 0 : ((c->x3_two_wave || c->wide_two_wave) ? 2 : 1);
    info->span_rows_a = c->geomA.L; info->span_rows_b = c->geomB.L;
    info->spans_per_workgroup_a = c->geomA.sub; info->spans_per_workgroup_b = c->geomB.sub;
    info->xcd_bias_per_mille = c->xcd_bias_pm; info->xcc_of_workgroup0 = c->xcc_of_wg0;
    info->team_width_a = c->geomA.gw; info->team_width_b = c->geomB.gw;
    return 0;
}

// ---------------------------------------------------------------------------------- ingest
// planes == 0: round to one bf16 plane; planes >= 1: exact split into `planes` planes (plane_stride elements apart)
static int launch_pack(alpine_ctx* c, const float* src, int64_t ld, int rows, int cols, unsigned short* dst, int64_t dst_cols,
                       int64_t k0, int64_t f0, int rows_are_k, int planes = 0, int64_t plane_stride = 0, int* flags = nullptr)
{
    dim3 grid((cols + 63) / 64, (rows + 63) / 64);
    if (planes > 0)
        hipLaunchKernelGGL(pack_split_kernel, grid, dim3(256), 0, c->stream, src, ld, rows, cols, dst, plane_stride, planes, dst_cols, k0, f0,
                           rows_are_k, flags);
    else
        hipLaunchKernelGGL(pack_bf16_kernel, grid, dim3(256), 0, c->stream, src, ld, rows, cols, dst, dst_cols, k0, f0, rows_are_k);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int upload_x_dev(alpine_ctx* c, const float* dev, int layout, int64_t ld, int64_t cell0, int64_t n)
{
    const int G = c->G;
    if (c->bf16) {
        if (cell0 % 8) return fail(c, ALPINE_ERR_BAD_ARG, "bf16 path: X chunks must start at a multiple of 8 cells (got %lld)", (long long)cell0);
        int rc;
        const int pl = c->split ? 2 : 0;
        if (layout == ALPINE_X_CELLS_BY_GENES) {            // chunk[cell][gene]
            if (!c->transform_only && (rc = launch_pack(c, dev, ld, (int)n, G, c->Xng16, c->Gp, cell0, 0, 1, pl, c->x_plane_ng, nullptr))) return rc;   // k = cell
            if ((rc = launch_pack(c, dev, ld, (int)n, G, c->Xgn16, c->Np, 0, cell0, 0, pl, c->x_plane_gn, c->xflags))) return rc;                        // k = gene
        } else {                                            // chunk[gene][cell]
            if ((rc = launch_pack(c, dev, ld, G, (int)n, c->Xgn16, c->Np, 0, cell0, 1, pl, c->x_plane_gn, c->xflags))) return rc;                        // k = gene
            if (!c->transform_only && (rc = launch_pack(c, dev, ld, G, (int)n, c->Xng16, c->Gp, cell0, 0, 0, pl, c->x_plane_ng, nullptr))) return rc;   // k = cell
        }
        return 0;
    }
    if (layout == ALPINE_X_CELLS_BY_GENES) {
        if (!c->transform_only)
            HIPCHK(c, hipMemcpy2DAsync(c->Xng + cell0 * c->Gp, sizeof(float) * c->Gp, dev, sizeof(float) * ld,
                                       sizeof(float) * G, (size_t)n, hipMemcpyDeviceToDevice, c->stream));
        dim3 grid((G + 31) / 32, (unsigned)((n + 31) / 32));
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, dev, ld, c->Xgn + cell0, c->Np, (int)n, G);
    } else {
        HIPCHK(c, hipMemcpy2DAsync(c->Xgn + cell0, sizeof(float) * c->Np, dev, sizeof(float) * ld,
                                   sizeof(float) * (size_t)n, (size_t)G, hipMemcpyDeviceToDevice, c->stream));
        dim3 grid((unsigned)((n + 31) / 32), (G + 31) / 32);
        if (!c->transform_only)
            hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, dev, ld, c->Xng + cell0 * c->Gp, c->Gp, G, (int)n);
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int check_x_args(alpine_ctx* c, const float* p, int layout, int64_t ld, int64_t cell0, int64_t n)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!p) return fail(c, ALPINE_ERR_BAD_ARG, "X pointer is NULL");
    if (layout != ALPINE_X_CELLS_BY_GENES && layout != ALPINE_X_GENES_BY_CELLS) return fail(c, ALPINE_ERR_BAD_ARG, "unknown X layout %d", layout);
    if (cell0 < 0 || n <= 0 || cell0 + n > c->N) return fail(c, ALPINE_ERR_BAD_ARG, "cell range [%lld, %lld) outside the shard's %d cells", (long long)cell0, (long long)(cell0 + n), c->N);
    if (c->x_plane2_dropped)
        return fail(c, ALPINE_ERR_STATE, "X was finalised as ONE exact bf16 plane and the second plane's memory was released: create a new ctx to upload other data");
    const int64_t min_ld = layout == ALPINE_X_CELLS_BY_GENES ? c->G : n;
    if (ld < min_ld) return fail(c, ALPINE_ERR_BAD_ARG, "ld %lld smaller than the row length %lld", (long long)ld, (long long)min_ld);
    return 0;
}

// coverage of the shard's cells by the uploaded chunks: union of intervals (re-uploading a chunk overwrites it and
// counts once); alpine_finalize_X requires the union to be exactly [0, N)
static void mark_x(alpine_ctx* c, int64_t cell0, int64_t n)
{
    auto& v = c->x_cover;
    v.emplace_back(cell0, cell0 + n);
    std::sort(v.begin(), v.end());
    size_t o = 0;
    for (size_t i = 1; i < v.size(); ++i) {
        if (v[i].first <= v[o].second) v[o].second = std::max(v[o].second, v[i].second);
        else v[++o] = v[i];
    }
    v.resize(o + 1);
    c->x_final = false;
}

extern "C" int alpine_upload_X_device(alpine_ctx* c, const float* dev, int layout, int64_t ld, int64_t cell0, int64_t n)
{
    int rc = check_x_args(c, dev, layout, ld, cell0, n);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    rc = upload_x_dev(c, dev, layout, ld, cell0, n);
    if (rc) return rc;
    mark_x(c, cell0, n);
    return 0;
}

extern "C" int alpine_upload_X_host(alpine_ctx* c, const float* host, int layout, int64_t ld, int64_t cell0, int64_t n)
{
    int rc = check_x_args(c, host, layout, ld, cell0, n);
    if (rc) return rc;
    HIPCHK(c, hipSetDevice(c->device));
    const int G = c->G;
    if (layout == ALPINE_X_CELLS_BY_GENES) {
        const int64_t step = std::max<int64_t>(8, c->stage_floats / G / 8 * 8);   // multiple of 8 cells: the bf16 paths pack 8 cells per granule
        for (int64_t r = 0; r < n; r += step) {
            const int64_t m = std::min(step, n - r);
            HOSTCOPY2D(c, c->stage, sizeof(float) * G, host + r * ld, sizeof(float) * ld, sizeof(float) * G, (size_t)m, hipMemcpyHostToDevice);   // (the drain also frees the staging buffer for reuse)
            rc = upload_x_dev(c, c->stage, layout, G, cell0 + r, m);
            if (rc) return rc;
        }
    } else {
        const int64_t step = std::max<int64_t>(8, c->stage_floats / G / 8 * 8);      // cells per piece (multiple of 8)
        for (int64_t r = 0; r < n; r += step) {
            const int64_t m = std::min(step, n - r);
            HOSTCOPY2D(c, c->stage, sizeof(float) * m, host + r, sizeof(float) * ld, sizeof(float) * m, (size_t)G, hipMemcpyHostToDevice);
            rc = upload_x_dev(c, c->stage, layout, m, cell0 + r, m);
            if (rc) return rc;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    mark_x(c, cell0, n);
    return 0;
}

static int sum_f64_partials(alpine_ctx* c, int n, double* out)
{
    std::vector<double> h(n);
    HOSTCOPY(c, h.data(), c->f64part, sizeof(double) * n, hipMemcpyDeviceToHost);          // (the rule: see host_copy)
    double s = 0;
    for (double v : h) s += v;
    *out = s;
    return 0;
}

static int launch_sweep(alpine_ctx* c, const SweepGeom& g, const float* S, const float* P, float* pieces, int which, int active16 = 0);
static int launch_sweeps_wide(alpine_ctx* c, const SweepGeom& g, const float* S, float* P, int64_t rows_pad, float* pieces, int64_t cap, int which);

extern "C" int alpine_finalize_X(alpine_ctx* c)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (c->x_cover.size() != 1 || c->x_cover[0].first != 0 || c->x_cover[0].second != c->N) {
        int64_t have = 0;
        for (auto& iv : c->x_cover) have += iv.second - iv.first;
        int64_t gap = 0;                                   // first cell that no chunk covered
        for (auto& iv : c->x_cover) { if (iv.first > gap) break; gap = std::max(gap, iv.second); }
        return fail(c, ALPINE_ERR_STATE, "only %lld of %d cells of X were uploaded (first missing cell: %lld)", (long long)have, c->N, (long long)gap);
    }
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t n4 = c->Gp * c->Np / (c->bf16 ? 8 : 4);
    const int blocks = (int)std::min<int64_t>(4096, (n4 + 255) / 256);
    if (c->split) {
        int h[2] = {0, 0};
        HOSTCOPY(c, h, c->xflags, sizeof h, hipMemcpyDeviceToHost);
        if (h[0]) return fail(c, ALPINE_ERR_UNSUPPORTED, "X is not exactly representable as the sum of two bf16 planes (more than 16 significant bits): use the float32 layout");
        c->npx = h[1] ? 2 : 1;        // small integer counts: the second plane is all zero and is never read
        if (c->npx == 1) {            // ... so give its memory back
            if (c->Xgn16b) { HIPCHK(c, dev_free(c, c->Xgn16b)); c->Xgn16b = nullptr; c->bytes -= sizeof(unsigned short) * (size_t)(c->Gp * c->Np); }
            if (c->Xng16b) { HIPCHK(c, dev_free(c, c->Xng16b)); c->Xng16b = nullptr; if (!c->transform_only) c->bytes -= sizeof(unsigned short) * (size_t)(c->Gp * c->Np); }
            c->x_plane2_dropped = true;                      // further uploads would write plane 2 into freed memory: refused (check_x_args)
        }
    }
    if (c->bf16) hipLaunchKernelGGL(sqnorm_bf16_kernel, dim3(blocks), dim3(256), 0, c->stream, c->Xgn16,
                                    (c->split && c->npx == 2) ? c->Xgn16b : (const unsigned short*)nullptr, n4, c->f64part);
    else hipLaunchKernelGGL(sqnorm_kernel, dim3(blocks), dim3(256), 0, c->stream, c->Xgn, n4, c->f64part);
    HIPCHK(c, hipGetLastError());
    int rc = sum_f64_partials(c, blocks, &c->xnorm2);
    if (rc) return rc;
    if (!c->bf16) {
        // Which matrix instruction the x3 sweeps use is decided by the data: on count-like X (every element one bf16 plane:
        // the mid / lo products are skipped) the sweep is memory-bound and the 32x32x16 form is ~1.5 % faster; on X with full
        // significands all six products run, the chip lowers its clock under the matrix load, and the 16x16x32 form (x3w)
        // holds a higher clock: ~5 % faster.  Both are float32-grade; they differ in summation order only.
        std::vector<double> h(2 * (size_t)blocks);
        HOSTCOPY(c, h.data(), c->f64part, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
        double multi = 0;
        for (int b = 0; b < blocks; ++b) multi += h[(size_t)blocks + b];
        c->x_multi_plane_frac = multi / ((double)c->G * (double)c->N);
        c->x_one_plane = multi == 0.0;                     // the census counts elements: an exact zero means EVERY element is one bf16 plane
        // (wide models with an all-padding 16-component tile -- K = 105 -> 7 of 8 tiles -- are matrix-pipe-bound in both data
        // regimes and x3w never multiplies that tile: 11 % faster on full significands, 5 % on counts at K = 105)
        const bool pad_tile = c->K <= c->KP - 16;
        // (K > 64 on one-plane data: x3w's one-plane form -- no split, no zero-plane test -- with or without a padding tile)
        c->x3_wide = c->x3_variant == 2 || (c->x3_variant < 0 && (c->x_multi_plane_frac > 0.01 || (pad_tile && c->KT >= 3) || (c->x_one_plane && c->KT >= 3)));
        // K in (64, 128] on data that is NOT one-plane throughout: the two-waves-per-SIMD form of the 16x16x32 sweep (stream_gemm_x3v_kernel,
        // kernels_x3.hpp).  tools/x3w_bench 60 (profiles/r04/x3v_bench_*.txt): full significands 3 - 10 % faster than x3w; in the
        // library's iteration (tools/option_ab.py, engines interleaved, profiles/r04/option_ab_*.txt) cfg4's share on full significands
        // 5.54 -> 5.25 ms, on one-plane counts level with x3w's one-plane form (3.50 / 3.53 vs 3.51 / 3.58 ms: inside the spread between
        // two engines of the SAME kind), which therefore keeps its kernel.  x3_two_wave = 1 (option): the one-plane form of x3v as well.
        c->x3_two_wave = c->x3 && !c->wide && c->KT >= 3 && !c->x3_ablate && c->x3_variant < 0 && std::max(c->Gp, c->Np) <= ((int64_t)1 << 25) &&
                         (c->x3_two_wave_opt == 1 || (c->x3_two_wave_opt < 0 && !c->x_one_plane));
        if (c->x3_two_wave) c->x3_wide = true;
        // 128 < K <= 160 on one-plane data: the 8-wave form of the one-pass sweep (two waves per SIMD, 512-column workgroup tiles)
        {
            const bool w2 = c->wide_one_pass && c->x_one_plane && c->x3_variant < 0 && c->K <= 160 && c->x3_two_wave_opt != 0;
            if (w2 != c->wide_two_wave) { c->wide_two_wave = w2; apply_sweep_geometry(c, w2 ? 512 : 256); c->tail_valid = false; }
        }
        const bool team_ok = c->KT >= 3 || c->x_multi_plane_frac <= 0.01;
        if (c->x3 && team_ok != c->team_ok) { c->team_ok = team_ok; apply_sweep_geometry(c, c->sweep_bf); c->tail_valid = false; }
        if (c->x3 && c->KT <= 2) {
            // tile width (see alpine_create): 512 columns only for small shards on the 32x32x16 form
            const bool narrow = c->x3_narrow_pref && (!c->x3_wide || c->x3_narrow_forced);
            if (narrow != c->x3_narrow) { c->x3_narrow = narrow; apply_sweep_geometry(c, narrow ? 512 : 1024); c->tail_valid = false; }
        }
    }
    if (c->xcd_bias_auto && (c->x3 || c->bf16)) {
        // Placement probe.  The XCDs do not stream the sweeps' access pattern at the same rate: on MI355X the odd XCCs finish a
        // sweep 6-7 % after the even ones (tools/stamps.py sweep, every box so far), and giving the workgroups that land on them
        // 4 % shorter spans takes 0.7-1.4 % off an iteration (tools/xcd_bias_sweep.py: one engine, interleaved; the XCDs share
        // one memory-side limit, so most of the imbalance is NOT recoverable).  Sweep workgroups go to the XCDs round-robin,
        // workgroup 0 always to the same XCC for this kernel (XCC 7 on every box so far; a grid of small workgroups starts on
        // XCC 0, so the question is put to the sweep kernel itself): one launch of the W^TX sweep over the resident X with the
        // (still zero) panel reports where its workgroup 0 ran.  The division itself stays a static function of blockIdx:
        // results never depend on placement, only this 1 % does.
        c->probe_placement = true;
        HIPCHK(c, hipMemsetAsync(c->xcc_dev, 0xff, sizeof(int) * 2, c->stream));        // -1: "no report" (a one-workgroup grid has no workgroup 1)
        rc = c->wide ? launch_sweeps_wide(c, c->geomB, c->Xgn, c->W, c->Gp, c->piecesB, c->piecesB_cap, 1)
                     : launch_sweep(c, c->geomB, c->Xgn, c->W, c->piecesB, 1);
        c->probe_placement = false;
        if (rc) return rc;
        int hh[2] = {-1, -1};
        HOSTCOPY(c, hh, c->xcc_dev, sizeof hh, hipMemcpyDeviceToHost);
        const int h = hh[0];
        c->xcc_of_wg0 = h;
        // workgroup 0 on an odd XCC: the even workgroups are the slow ones.  No bias unless workgroup 1 sits on an XCC of the OTHER parity
        // (a partitioned device with every workgroup on one XCC, a one-workgroup grid): the even/odd split would only unbalance the grid.
        const int bias = (h < 0 || hh[1] < 0 || ((h ^ hh[1]) & 1) == 0) ? 0 : ((h & 1) ? -XCD_BIAS_MAG : XCD_BIAS_MAG);
        if (bias != c->xcd_bias_pm) {
            const int old_bias = c->xcd_bias_pm;
            c->xcd_bias_pm = bias;
            apply_sweep_geometry(c, c->sweep_bf);
            c->tail_valid = false;
            // the pieces buffers were sized for this division (create_impl); never launch a geometry they do not hold
            if ((int64_t)c->geomA.nwg * c->geomA.maxp * c->geomA.bf * c->KP > (c->transform_only ? INT64_MAX : c->piecesA_cap) ||
                (int64_t)c->geomB.nwg * c->geomB.maxp * c->geomB.bf * c->KP > c->piecesB_cap) {
                c->xcd_bias_pm = old_bias;
                apply_sweep_geometry(c, c->sweep_bf);
            }
        }
    }
    c->x_final = true;
    return 0;
}

extern "C" int alpine_upload_Y(alpine_ctx* c, int cov, const float* host, int64_t ld)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (cov < 0 || cov >= c->n_cov) return fail(c, ALPINE_ERR_BAD_ARG, "covariate index %d out of range", cov);
    if (!host || ld < c->N) return fail(c, ALPINE_ERR_BAD_ARG, "bad Y pointer / ld");
    HIPCHK(c, hipSetDevice(c->device));
    HOSTCOPY2D(c, c->Y + (int64_t)c->meta.yoff[cov] * c->Np, sizeof(float) * c->Np, host, sizeof(float) * ld,
               sizeof(float) * c->N, (size_t)c->cov_lev[cov], hipMemcpyHostToDevice);
    c->y_set[cov] = true;
    c->tail_valid = false;
    return 0;
}

extern "C" int alpine_set_factors(alpine_ctx* c, const float* W, const float* H, int64_t ldH, const float* const* B)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!W || !H || ldH < c->N || (c->n_cov > 0 && !B)) return fail(c, ALPINE_ERR_BAD_ARG, "bad factor pointers / ldH");
    HIPCHK(c, hipSetDevice(c->device));
    const int K = c->K, KP = c->KP;
    // W: G x K -> [Gp][KP]   (wide: column half h -> [h][Gp][128])
    const int halves = c->wide ? c->NH : 1, kph = c->wide ? WIDE_KH : KP;
    HIPCHK(c, hipMemsetAsync(c->W, 0, sizeof(float) * c->Gp * KP, c->stream));
    HOSTCOPY(c, c->stage, W, sizeof(float) * (size_t)c->G * K, hipMemcpyHostToDevice);
    for (int h = 0; h < halves; ++h) {
        const int kh = std::min(kph, K - h * kph);
        const int64_t n = (int64_t)c->G * kh;
        hipLaunchKernelGGL(pad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->stage + h * kph, (int64_t)K,
                           c->W + (int64_t)h * c->Gp * kph, kph, (int64_t)c->G, kh);
    }
    // H: K x N (ldH) -> [Np][KP] cell-major   (wide: row block h of H -> [h][Np][128]); the helper's drain also ends the kernels that read the staged W
    HIPCHK(c, hipMemsetAsync(c->H, 0, sizeof(float) * c->Np * KP, c->stream));
    HOSTCOPY2D(c, c->stage, sizeof(float) * c->N, H, sizeof(float) * ldH, sizeof(float) * c->N, (size_t)K, hipMemcpyHostToDevice);
    for (int h = 0; h < halves; ++h) {
        const int kh = std::min(kph, K - h * kph);
        dim3 grid((unsigned)((c->N + 31) / 32), (kh + 31) / 32);
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, c->stage + (int64_t)h * kph * c->N, (int64_t)c->N,
                           c->H + (int64_t)h * c->Np * kph, (int64_t)kph, kh, c->N);
    }
    c->bcur = 0;
    for (int i = 0; i < c->n_cov; ++i) {
        if (c->cov_lev[i] * c->cov_k[i] == 0) continue;                       // k_i = 0: B_i is C_i x 0
        if (!B[i]) return fail(c, ALPINE_ERR_BAD_ARG, "B[%d] is NULL", i);
        HOSTCOPY(c, c->B[0] + c->meta.boff[i], B[i], sizeof(float) * c->cov_lev[i] * c->cov_k[i], hipMemcpyHostToDevice);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipGetLastError());
    c->factors_set = true;
    c->pending_loss = false;
    c->tail_valid = false;
    return 0;
}

extern "C" int alpine_get_factors(alpine_ctx* c, float* W, float* H, int64_t ldH, float* const* B)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!c->factors_set) return fail(c, ALPINE_ERR_STATE, "factors were never set");
    if ((H && ldH < c->N)) return fail(c, ALPINE_ERR_BAD_ARG, "ldH too small");
    HIPCHK(c, hipSetDevice(c->device));
    const int K = c->K, KP = c->KP;
    const int halves = c->wide ? c->NH : 1, kph = c->wide ? WIDE_KH : KP;
    if (W) {
        for (int h = 0; h < halves; ++h) {
            const int kh = std::min(kph, K - h * kph);
            const int64_t n = (int64_t)c->G * kh;
            hipLaunchKernelGGL(unpad_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->W + (int64_t)h * c->Gp * kph, kph,
                               c->stage + h * kph, (int64_t)K, (int64_t)c->G, kh);
        }
        HOSTCOPY(c, W, c->stage, sizeof(float) * (size_t)c->G * K, hipMemcpyDeviceToHost);
    }
    if (H) {
        for (int h = 0; h < halves; ++h) {
            const int kh = std::min(kph, K - h * kph);
            dim3 grid((kh + 31) / 32, (unsigned)((c->N + 31) / 32));
            hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, c->H + (int64_t)h * c->Np * kph, (int64_t)kph,
                               c->stage + (int64_t)h * kph * c->N, (int64_t)c->N, c->N, kh);
        }
        HOSTCOPY2D(c, H, sizeof(float) * ldH, c->stage, sizeof(float) * c->N, sizeof(float) * c->N, (size_t)K, hipMemcpyDeviceToHost);
    }
    if (B) for (int i = 0; i < c->n_cov; ++i) if (B[i] && c->cov_lev[i] * c->cov_k[i] > 0)
        HOSTCOPY(c, B[i], c->B[c->bcur] + c->meta.boff[i], sizeof(float) * c->cov_lev[i] * c->cov_k[i], hipMemcpyDeviceToHost);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------- iteration
static int prof_begin(alpine_ctx* c, int which)
{
    if (!c->prof_now) return 0;
    auto& v = c->ev[which];
    if (c->ev_used[which] == v.size()) {
        hipEvent_t a, b;
        HIPCHK(c, hipEventCreate(&a));
        HIPCHK(c, hipEventCreate(&b));
        v.emplace_back(a, b);
    }
    HIPCHK(c, hipEventRecord(v[c->ev_used[which]].first, c->stream));
    return 0;
}
static int prof_end(alpine_ctx* c, int which)
{
    if (!c->prof_now) return 0;
    HIPCHK(c, hipEventRecord(c->ev[which][c->ev_used[which]].second, c->stream));
    c->ev_used[which]++;
    return 0;
}

static int launch_gram(alpine_ctx* c, const float* A, int64_t R, int /*blocks_hint*/, float* out)
{
    const int rpw = gram_rows_per_wave(R, c->n_cu);
    const int blocks = (int)((R + 4 * rpw - 1) / (4 * rpw));
    // (the partial-block buffers are sized at alpine_create from the shard AND the largest mini-batch view; a launch that would write
    // past them is refused here instead of faulting on the device -- the common shape of the two faults the round-3 fuzz found)
    if (blocks > c->gramPart_cap) return fail(c, ALPINE_ERR_STATE, "internal: the Gram product of %lld rows needs %d partial blocks, the buffer holds %lld", (long long)R, blocks, (long long)c->gramPart_cap);
    DISPATCH_KT(c->KT, hipLaunchKernelGGL(gram_kernel<KT_>, dim3(blocks), dim3(256), 0, c->stream, A, c->gramPart, (int)R, rpw));
    HIPCHK(c, hipGetLastError());
    const int n = c->KP * c->KP;
    hipLaunchKernelGGL(reduce_many_kernel, dim3((n + 63) / 64), dim3(1024), 0, c->stream, c->gramPart, out, n, blocks);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// the two streaming sweeps share one launcher; the variant only changes the pipeline shape, never the result
// which: 0 = XH^T (S = cells x genes copy, panel H), 1 = W^TX (S = genes x cells copy, panel W)
static int launch_sweep_bf16(alpine_ctx* c, int which, const SweepGeom& g_in)
{
    SweepGeom g = g_in;
    g.panel_fixed = c->ablate_panel ? 1 : 0;
    const float* master = which == 0 ? c->H : c->W;
    unsigned short* panel = which == 0 ? c->Hp16 : c->Wp16;
    const int64_t p_plane = (int64_t)g.R * c->KP;
    // rounded mode: float32 master -> k-packed bf16 operand copy once per sweep; the split forms read the float32 master
    // and split it into exact planes inside the sweep
    if (!c->split) {
        int rc = launch_pack(c, master, c->KP, g.R, c->KP, panel, c->KP, 0, 0, 1, 0, p_plane, nullptr);
        if (rc) return rc;
    }
    const unsigned short* S = which == 0 ? c->Xng16 : c->Xgn16;
    float* pieces = which == 0 ? c->piecesA : c->piecesB;
    int* xcc_out = c->probe_placement ? c->xcc_dev : nullptr;
#define BF_ARGS S, (which == 0 ? c->x_plane_ng : c->x_plane_gn), panel, p_plane, master, pieces, g, xcc_out
#define BF_LAUNCH(NPX, NPP) do { \
        if (g.bf == 8 * SG_WAVE_F * g.gw) {   /* 8 waves: K <= 64 only (create_impl); g.bf = the TEAM tile */ \
            if (c->KT == 1) hipLaunchKernelGGL((stream_gemm_bf16_kernel<1, NPX, NPP, 8>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, BF_ARGS); \
            else            hipLaunchKernelGGL((stream_gemm_bf16_kernel<2, NPX, NPP, 8>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, BF_ARGS); \
        } else if (c->KT >= 3 && c->x3_two_wave_opt != 0 && std::max(c->Gp, c->Np) <= ((int64_t)1 << 25)) {   /* K > 64: eight 64-column waves, two per SIMD, 32-bit lane offsets (option "x3_two_wave" 0: the 4-wave form) */ \
            if (c->KT == 3) hipLaunchKernelGGL((stream_gemm_bf16_kernel<3, NPX, NPP, 8, 2>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, BF_ARGS); \
            else            hipLaunchKernelGGL((stream_gemm_bf16_kernel<4, NPX, NPP, 8, 2>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, BF_ARGS); \
        } else { \
            DISPATCH_KT(c->KT, hipLaunchKernelGGL((stream_gemm_bf16_kernel<KT_, NPX, NPP, 4>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, BF_ARGS)); \
        } } while (0)
    // exact-plane storage at K > 64: the two-wave sweep on the 16x16x32 instruction (stream_gemm_bf16v_kernel, kernels_x3.hpp; 32-bit lane offsets:
    // axes up to 2^25; option "x3_two_wave" 0: the 32x32x16 kernel)
    if (c->split && c->KT >= 3 && c->x3_two_wave_opt != 0 && std::max(c->Gp, c->Np) <= ((int64_t)1 << 25) && g.bf == 512 * g.gw) {
        const bool pad_tile = c->K <= c->KP - 16;
        const int64_t xpl = which == 0 ? c->x_plane_ng : c->x_plane_gn;
#define BFV(KT_, M_) do { \
            if (c->npx == 1) hipLaunchKernelGGL((stream_gemm_bf16v_kernel<KT_, M_, 1>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, S, xpl, master, pieces, g, xcc_out); \
            else hipLaunchKernelGGL((stream_gemm_bf16v_kernel<KT_, M_, 2>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, S, xpl, master, pieces, g, xcc_out); } while (0)
        if (c->KT == 3) { if (pad_tile) BFV(3, 5); else BFV(3, 6); }
        else { if (pad_tile) BFV(4, 7); else BFV(4, 8); }
#undef BFV
        HIPCHK(c, hipGetLastError());
        return 0;
    }
    if (!c->split) { BF_LAUNCH(1, 1); }
    else if (c->npx == 1) { BF_LAUNCH(1, 3); }
    else { BF_LAUNCH(2, 3); }
#undef BF_LAUNCH
#undef BF_ARGS
    HIPCHK(c, hipGetLastError());
    return 0;
}

// active16 > 0 (wide models, second component half): only that many 16-component tiles of the panel hold real components; the x3w form
// leaves the others unmultiplied (zero accumulators, same pieces layout), the other forms ignore the hint
static int launch_sweep(alpine_ctx* c, const SweepGeom& g, const float* S, const float* P, float* pieces, int which, int active16)
{
    // the division about to be launched must fit the pieces buffer it writes (sized in create_impl for every division this ctx
    // may use): an error here instead of an out-of-bounds write on the device
    const int64_t need = (int64_t)g.nwg * g.maxp * g.bf * (c->wide ? WIDE_KH : c->KP);        // (wide: one pass = one half of the buffer)
    if (need > (which == 0 ? c->piecesA_cap : c->piecesB_cap) / (c->wide ? c->NH : 1) || (which == 0 && c->transform_only))
        return fail(c, ALPINE_ERR_STATE, "internal: sweep %d needs %lld floats of pieces, the buffer holds %lld", which, (long long)need,
                    (long long)(which == 0 ? c->piecesA_cap : c->piecesB_cap));
    if (c->bf16) return launch_sweep_bf16(c, which, g);
    const int64_t ldS = c->ablate_stride0 ? 0 : g.F;
    if (c->x3) {
        // the division must be for THIS kernel's workgroup tile: a wave reads the columns its tile index names
        const int bf_wg = (c->KT <= 2 && !c->wide && !c->x3_narrow) ? 1024 : 512;
        if (c->wide_one_pass) return fail(c, ALPINE_ERR_STATE, "internal: a one-pass wide ctx launched a per-half sweep");
        if (g.bf != bf_wg * g.gw)
            return fail(c, ALPINE_ERR_STATE, "internal: sweep %d was divided for %d-column workgroup tiles, its kernel works on %d", which, g.bf / std::max(1, g.gw), bf_wg);
        SweepGeom gx = g;
        int* xcc_out = c->probe_placement ? c->xcc_dev : nullptr;
        if (c->ablate_flush) gx.panel_fixed = 2;
        else if (c->ablate_panel) gx.panel_fixed = 1;
        if (c->x3_wide && !c->x3_ablate && c->wide && active16 > 0 && active16 < 7) {
#define X3W_PART(M_) case M_: hipLaunchKernelGGL((stream_gemm_x3w_kernel<4, 1, M_>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); break
            switch (active16) { X3W_PART(1); X3W_PART(2); X3W_PART(3); X3W_PART(4); X3W_PART(5); default: X3W_PART(6); }
#undef X3W_PART
            HIPCHK(c, hipGetLastError());
            return 0;
        }
        if (c->x3_two_wave) {
            const bool pad_tile = c->K <= c->KP - 16;                 // the last 16-component tile is all padding: not multiplied
            const bool one_plane = c->x_one_plane;
#define X3V_LAUNCH(KT_, M_) do { \
                if (one_plane) hipLaunchKernelGGL((stream_gemm_x3v_kernel<KT_, 1, M_, true>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); \
                else hipLaunchKernelGGL((stream_gemm_x3v_kernel<KT_, 1, M_, false>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); } while (0)
            if (c->KT == 3) { if (pad_tile) X3V_LAUNCH(3, 5); else X3V_LAUNCH(3, 6); }
            else if (c->KT == 4) { if (pad_tile) X3V_LAUNCH(4, 7); else X3V_LAUNCH(4, 8); }
            else return fail(c, ALPINE_ERR_STATE, "internal: the two-wave sweep exists for 64 < K <= 128 only");
#undef X3V_LAUNCH
            HIPCHK(c, hipGetLastError());
            return 0;
        }
        if (c->x3_wide && !c->x3_ablate) {
            const bool pad_tile = (!c->wide && c->K <= c->KP - 16) || (c->wide && active16 == 7);          // the last 16-component tile is all padding: not multiplied
#define X3W_LAUNCH(KT_, NH_) do { \
                if (pad_tile) hipLaunchKernelGGL((stream_gemm_x3w_kernel<KT_, NH_, 2 * KT_ - 1>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); \
                else hipLaunchKernelGGL((stream_gemm_x3w_kernel<KT_, NH_>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); } while (0)
#define X3W_LAUNCH1(KT_) do {     /* every element of X is exactly one bf16 plane (census): the form without split and test */ \
                if (pad_tile) hipLaunchKernelGGL((stream_gemm_x3w_kernel<KT_, 1, 2 * KT_ - 1, true>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); \
                else hipLaunchKernelGGL((stream_gemm_x3w_kernel<KT_, 1, 2 * KT_, true>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); } while (0)
            const bool one_plane = c->x_one_plane && c->x3_variant < 0;        // (wide models: the full and the 7-tile component blocks; partly filled last blocks take X3W_PART above)
            switch (c->KT) {
                case 1: if (c->x3_narrow) X3W_LAUNCH(1, 1); else X3W_LAUNCH(1, 2); break;
                case 2: if (c->x3_narrow) X3W_LAUNCH(2, 1); else X3W_LAUNCH(2, 2); break;
                case 3: if (one_plane) X3W_LAUNCH1(3); else X3W_LAUNCH(3, 1); break;
                default: if (one_plane) X3W_LAUNCH1(4); else X3W_LAUNCH(4, 1); break;
            }
#undef X3W_LAUNCH1
#undef X3W_LAUNCH
            HIPCHK(c, hipGetLastError());
            return 0;
        }
        switch (c->KT) {
            case 1:
                if (c->x3_narrow) hipLaunchKernelGGL((stream_gemm_x3_kernel<1, 1>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out);
                else hipLaunchKernelGGL((stream_gemm_x3_kernel<1, 2>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out);
                break;
            case 2:
#ifdef ALPINE_DIAGNOSTICS
                if (c->x3_ablate == 1) { hipLaunchKernelGGL((stream_gemm_x3_kernel<2, 2, 1>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); break; }
                if (c->x3_ablate == 2) { hipLaunchKernelGGL((stream_gemm_x3_kernel<2, 2, 2>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); break; }
                if (c->x3_ablate == 3) { hipLaunchKernelGGL((stream_gemm_x3_kernel<2, 2, 3>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); break; }
#endif
                if (c->x3_narrow) hipLaunchKernelGGL((stream_gemm_x3_kernel<2, 1>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out);
                else hipLaunchKernelGGL((stream_gemm_x3_kernel<2, 2>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out);
                break;
            case 3: hipLaunchKernelGGL((stream_gemm_x3_kernel<3, 1>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); break;
            default: hipLaunchKernelGGL((stream_gemm_x3_kernel<4, 1>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, P, pieces, ldS, gx, xcc_out); break;
        }
        HIPCHK(c, hipGetLastError());
        return 0;
    }
#define SG_LAUNCH(RING, PASSES) \
    DISPATCH_KT(c->KT, hipLaunchKernelGGL((stream_gemm_kernel<KT_, RING, PASSES>), dim3(sweep_grid(g)), dim3(SG_THREADS), 0, c->stream, S, P, pieces, ldS, g, (unsigned long long*)nullptr))
    switch (c->sg_variant) {
        case 1: SG_LAUNCH(8, 4); break;
        case 2: SG_LAUNCH(8, 2); break;
        default: SG_LAUNCH(16, 1); break;
    }
#undef SG_LAUNCH
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int launch_reduce_pieces(alpine_ctx* c, const float* pieces, float* out, int rows, const SweepGeom& g, int kp = 0)
{
    if (kp == 0) kp = c->KP;
    const int64_t n4 = (int64_t)rows * kp / 4;
    const int blocks = (int)std::min<int64_t>(c->n_cu * 8, (n4 + 255) / 256);
    hipLaunchKernelGGL(reduce_pieces_kernel, dim3(std::max(1, blocks)), dim3(256), 0, c->stream, pieces, out, rows, kp, g);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int ready(alpine_ctx* c)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!c->x_final) return fail(c, ALPINE_ERR_STATE, "X not finalised (alpine_upload_X_* then alpine_finalize_X)");
    if (!c->factors_set) return fail(c, ALPINE_ERR_STATE, "factors not set (alpine_set_factors)");
    for (int i = 0; i < c->n_cov; ++i) if (!c->y_set[i]) return fail(c, ALPINE_ERR_STATE, "Y[%d] not uploaded", i);
    HIPCHK(c, hipSetDevice(c->device));
    return 0;
}

// phase 1 on a view: local sums of everything that depends on the old H of the view's cells -> reduce block
static int phase1(alpine_ctx* c, const CellView& v)
{
    int rc;
    c->prof_now = c->prof && (c->prof_tick++ % c->prof_every) == 0;      // holds until the next phase 1 (covers the all-reduce and the W^TX sweep)
    const int KP = c->KP;
    int max_k = 1, max_c = 1;
    for (int i = 0; i < c->n_cov; ++i) { max_k = std::max(max_k, c->cov_k[i]); max_c = std::max(max_c, c->cov_lev[i]); }
    const int max_ct = std::min(HS_CT, max_c);
    const size_t hs_bytes = hstats_group_bytes(max_k, max_ct);
    if (v.statBlocks > c->statPart_cap) return fail(c, ALPINE_ERR_STATE, "internal: %d statistics blocks, the buffer holds %lld", v.statBlocks, (long long)c->statPart_cap);
    if (c->unfused_mid) {            // A/B: every small kernel and every reduction in its own launch
        if (c->n_cov > 0) {
            hipLaunchKernelGGL(hstats_kernel, dim3(v.statBlocks), dim3(HS_CELLS), hs_bytes, c->stream, v.H, v.Y, c->B[c->bcur], c->meta,
                               c->statPart, v.N, v.Np, KP, (float)c->eps, c->nstat, max_k, max_ct);
            HIPCHK(c, hipGetLastError());
        }
        hipLaunchKernelGGL(reduce_stats_kernel, dim3(c->nstat + 1), dim3(256), 0, c->stream, c->statPart, c->kind, c->red + c->red_stats,
                           v.statBlocks, c->nstat, c->xnorm2);
        HIPCHK(c, hipGetLastError());
        if ((rc = launch_gram(c, v.H, v.Np, v.gramBlocksH, c->red + c->red_hht))) return rc;
        if ((rc = prof_begin(c, ALPINE_KERNEL_SWEEP_XHT))) return rc;
        if ((rc = launch_sweep(c, v.gA, v.Xng, v.H, c->piecesA, 0))) return rc;
        if ((rc = prof_end(c, ALPINE_KERNEL_SWEEP_XHT))) return rc;
        return launch_reduce_pieces(c, c->piecesA, c->red, (int)c->Gp, v.gA);
    }
    // ONE launch for the two small kernels that read only the old H (H H^T partial blocks, covariate statistics) -- or none
    // at all when the previous H update already produced them in its tail -- then the sweep, then ONE launch for the
    // three reductions that close phase 1
    const bool from_tail = c->tail_valid && v.H == c->H;
    const int rpw = gram_rows_per_wave(v.Np, c->n_cu);
    const int gblocks = (int)((v.Np + 4 * rpw - 1) / (4 * rpw));
    if (v.statBlocks > c->statPart_cap || (!from_tail && gblocks > c->gramPart_cap) || (from_tail && v.statBlocks > c->tail_blocks))
        return fail(c, ALPINE_ERR_STATE, "internal: phase 1 of %d cells needs %d statistics / %d Gram partial blocks, the buffers hold %lld / %lld", v.N, v.statBlocks, gblocks,
                    (long long)c->statPart_cap, (long long)c->gramPart_cap);
    if (!from_tail) {
        const int stat_blocks2 = c->n_cov > 0 ? (v.statBlocks + 1) / 2 : 0;
        DISPATCH_KT(c->KT, hipLaunchKernelGGL(phase1_open_kernel<KT_>, dim3(gblocks + stat_blocks2), dim3(256), 2 * hs_bytes, c->stream, v.H, c->gramPart,
                                               (int)v.Np, rpw, gblocks, v.Y, c->B[c->bcur], c->meta, c->statPart, v.N, v.Np, (float)c->eps, c->nstat,
                                               max_k, max_ct, v.statBlocks));
        HIPCHK(c, hipGetLastError());
    }
    if ((rc = prof_begin(c, ALPINE_KERNEL_SWEEP_XHT))) return rc;
    if ((rc = launch_sweep(c, v.gA, v.Xng, v.H, c->piecesA, 0))) return rc;
    if ((rc = prof_end(c, ALPINE_KERNEL_SWEEP_XHT))) return rc;
    Phase1Reduce a{};
    const int64_t n4 = (int64_t)c->Gp * KP / 4;
    a.nb_pieces = (int)std::max<int64_t>(1, std::min<int64_t>(c->n_cu * 8, (n4 + 255) / 256));
    a.nb_many = (KP * KP + 15) / 16;
    a.nb_stats = c->nstat + 1;
    a.pieces = c->piecesA; a.xht = c->red; a.rows = (int)c->Gp;
    a.gram_part = from_tail ? c->gramPartH : c->gramPart; a.hht = c->red + c->red_hht; a.n_hht = KP * KP; a.n_slab = from_tail ? c->tail_blocks : gblocks;
    a.stat_part = from_tail ? c->statPartH : c->statPart; a.kind = c->kind; a.stats = c->red + c->red_stats; a.stat_blocks = v.statBlocks; a.nstat = c->nstat; a.xnorm2 = c->xnorm2;
    hipLaunchKernelGGL(phase1_reduce_kernel, dim3(a.nb_pieces + a.nb_many + a.nb_stats), dim3(256), 0, c->stream, a, KP, v.gA);
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int grow_losses(alpine_ctx* c)
{
    const int w = c->n_cov + 2;
    double* bigger = nullptr;
    int rc = dev_alloc(c, reinterpret_cast<void**>(&bigger), sizeof(double) * c->loss_cap * 2 * w, false, "loss_dev");
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(bigger, c->loss_dev, sizeof(double) * c->loss_rows * w, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->bytes -= sizeof(double) * c->loss_cap * w;
    HIPCHK(c, dev_free(c, c->loss_dev));
    c->loss_dev = bigger;
    c->loss_cap *= 2;
    return 0;
}

// gram_part != nullptr: the kernel also writes one partial block of W_new^T W_new per 128 genes (MFMA form only)
static int launch_w_update(alpine_ctx* c, const float* HHt, bool update, int k_lo, int k_hi, bool block_orth, float* gram_part = nullptr)
{
    const int KP = c->KP;
    const float l2 = (float)((1.0 - c->l1r) * c->alpha), l1 = (float)(c->l1r * c->alpha);
    {
        // 32 genes per wave: ceil(G/128) blocks; the remaining dotpart entries stay at their initial zero
        const size_t bytes = sizeof(float) * (KP * KP + 4 * 32 * (KP + 4));          // M + the waves' row-major tiles
        if (bytes > c->lds_dev) return fail(c, ALPINE_ERR_UNSUPPORTED, "internal: the W update needs %zu bytes of LDS, the device has %zu", bytes, c->lds_dev);
        DISPATCH_KT(c->KT, hipLaunchKernelGGL(w_update_mfma_kernel<KT_>, dim3((c->G + 127) / 128), dim3(256), bytes, c->stream, c->W, c->red,
                                               HHt, c->dotpart, c->G, c->K, (float)c->orth, l2, l1, (float)c->eps, update ? 1 : 0, k_lo, k_hi,
                                               block_orth ? 1 : 0, gram_part));
    }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static void cov_maxima(const alpine_ctx* c, int* max_k, int* max_ct)
{
    int mk = 1, mc = 1;
    for (int i = 0; i < c->n_cov; ++i) { mk = std::max(mk, c->cov_k[i]); mc = std::max(mc, c->cov_lev[i]); }
    *max_k = mk; *max_ct = std::min(HS_CT, mc);
}

// with_tail: MU branch on the whole shard -> the kernel also emits the next phase 1's H H^T partial blocks and covariate
// statistics (HTail in kernels.hpp)
static int launch_h_update(alpine_ctx* c, const CellView& v, int k_lo, int k_hi, int only_cov, bool with_tail = false)
{
    const int KP = c->KP, K = c->K;
    size_t h_bytes_mfma = sizeof(float) * (KP * KP + ((std::max(1, c->nB) + 3) & ~3) + 4 * 32 * (KP + 4));   // + per-wave transpose scratch
    {
        HTail tail{};
        const int hblocks = (int)((v.N + 127) / 128);
        // Optional LDS on top of the update's own (2 W^T W, B, the waves' tiles: ~134 KB at K > 96): the Y copy of the block's
        // cells and the fused tail's statistics scratch.  Both are dropped when the total would exceed the CU's LDS -- Y is
        // then read from global memory and the next phase 1 runs phase1_open_kernel instead of finding its inputs in the tail
        // (same results, one more launch) -- e.g. K = 100 with a covariate of k = 10 and 30 levels plus a second covariate.
        const size_t base = h_bytes_mfma;
        int max_k = 1, max_ct = 1;
        cov_maxima(c, &max_k, &max_ct);
        int guided = 0;
        for (int i = 0; i < c->n_cov; ++i) guided += c->cov_k[i];
        const int kg = (guided + 7) / 8 * 8;
        const bool merged_ok = guided + c->nYrows <= HT_MERGED_ROWS && !c->tail_stats_per_cov;
        // LDS on top of the update's own, in order of preference (same results in every form, fewer launches / less latency first):
        //   gm  = guided terms as two small matrix products on the MFMA: needs all of Y's rows in LDS (<= 32)
        //   y   = the Y copy of the block's cells in LDS (else Y is read from global memory per covariate and class)
        //   tail = the fused tail (else the next phase 1 runs phase1_open_kernel)
        auto total_bytes = [&](bool gm, int ybuf_rows, bool tail_on) {
            size_t b = base + sizeof(float) * (size_t)(gm ? (ybuf_rows + 7) / 8 * 8 : ybuf_rows) * HS_CELLS;
            if (gm) b += sizeof(GuidedRow) * 32;
            if (tail_on) b += ybuf_rows ? (merged_ok ? hstats_merged_bytes(guided, c->nYrows) : hstats_tail_bytes(max_k, max_ct))
                                        : 2 * hstats_group_bytes(max_k, max_ct);
            return b;
        };
        const bool y_fits = c->nYrows > 0 && c->nYrows <= HT_YROWS_MAX;
        const bool gm_ok = y_fits && !c->no_guided_mfma;
        struct Form { bool gm; int yrows; bool tail; };
        const Form forms[] = {{gm_ok, c->nYrows, with_tail}, {false, y_fits ? c->nYrows : 0, with_tail}, {false, 0, with_tail},
                              {gm_ok, c->nYrows, false}, {false, y_fits ? c->nYrows : 0, false}, {false, 0, false}};
        const Form* pick = nullptr;
        for (const Form& f : forms) {
            if (f.gm && !gm_ok) continue;
            if (total_bytes(f.gm, f.yrows, f.tail) <= c->lds_max) { pick = &f; break; }
        }
        if (!pick)
            return fail(c, ALPINE_ERR_UNSUPPORTED, "internal: the H update needs %zu bytes of LDS, the device has %zu", total_bytes(false, 0, false), c->lds_max);
        with_tail = pick->tail;
        tail.ybuf_rows = pick->yrows;
        tail.gm = pick->gm ? 1 : 0;
        tail.nY = c->nYrows; tail.kg = kg; tail.rowtab = c->rowtab;
        tail.merged = (merged_ok && pick->yrows > 0) ? 1 : 0; tail.guided = guided;
        tail.nstat = c->nstat; tail.max_k = max_k; tail.max_ct = max_ct;
        h_bytes_mfma = total_bytes(pick->gm, pick->yrows, pick->tail);
        if (with_tail && hblocks > c->tail_blocks)
            return fail(c, ALPINE_ERR_STATE, "internal: the fused tail of %d cells writes %d blocks, its buffers hold %d", v.N, hblocks, c->tail_blocks);
        if (with_tail) { tail.gram_part = c->gramPartH; tail.stat_part = c->statPartH; }
        if (c->loss_type == ALPINE_LOSS_KL) {
            DISPATCH_KT(c->KT, hipLaunchKernelGGL((h_update_mfma_kernel<KT_, 0>), dim3(hblocks), dim3(256), h_bytes_mfma, c->stream, v.H, c->piecesB, v.gB,
                                                   c->WtW, v.Y, c->B[c->bcur], c->meta, v.N, v.Np, K, (float)c->eps, c->nB, k_lo, k_hi, only_cov, tail));
        } else {
            DISPATCH_KT(c->KT, hipLaunchKernelGGL((h_update_mfma_kernel<KT_, 1>), dim3(hblocks), dim3(256), h_bytes_mfma, c->stream, v.H, c->piecesB, v.gB,
                                                   c->WtW, v.Y, c->B[c->bcur], c->meta, v.N, v.Np, K, (float)c->eps, c->nB, k_lo, k_hi, only_cov, tail));
        }
    }
    HIPCHK(c, hipGetLastError());
    c->tail_valid = with_tail;
    return 0;
}

static int launch_sweep_wtx(alpine_ctx* c, const CellView& v)
{
    int rc;
    if ((rc = prof_begin(c, ALPINE_KERNEL_SWEEP_WTX))) return rc;
    if ((rc = launch_sweep(c, v.gB, v.Xgn, c->W, c->piecesB, 1))) return rc;
    return prof_end(c, ALPINE_KERNEL_SWEEP_WTX);
}

// Block-coordinate branch, one component group (main.py:525-588).  als_group_hht: H H^T of the view with the groups
// before `grp` already updated (a sum over cells: in a sharded run the caller all-reduces that K x K slot of the reduce
// block before als_group_update); als_group_update: W_grp with the block-local orthogonality term, W^TW, W^TX sweep, H_grp.
static int launch_gram_wide(alpine_ctx* c, float* A, int64_t R, float* out);
static int als_group_update_wide(alpine_ctx* c, const CellView& v, int grp, int k_lo, int k_hi);

static int als_group_hht(alpine_ctx* c, const CellView& v, int grp)
{
    if (grp == 0) return 0;                      // the reduce block of phase 1 already holds H H^T of the old H
    if (c->wide) return launch_gram_wide(c, v.H, v.Np, c->red + c->red_hht);
    return launch_gram(c, v.H, v.Np, v.gramBlocksH, c->red + c->red_hht);
}

static int als_group_update(alpine_ctx* c, const CellView& v, int grp)
{
    int rc;
    int k_lo = 0;
    for (int g = 0; g < grp; ++g) k_lo += c->cov_k[g];
    const int k_hi = grp < c->n_cov ? k_lo + c->cov_k[grp] : c->K;
    if (c->wide) return als_group_update_wide(c, v, grp, k_lo, k_hi);
    const float* HHt = c->red + c->red_hht;
    if ((rc = launch_w_update(c, HHt, true, k_lo, k_hi, true))) return rc;
    if ((rc = launch_gram(c, c->W, c->Gp, c->gramBlocksW, c->WtW))) return rc;
    if ((rc = launch_sweep_wtx(c, v))) return rc;
    return launch_h_update(c, v, k_lo, k_hi, grp);
}

// phase 2 on a view: [loss row of the factors that produced the reduce block], W update, B updates, W^TW, W^TX sweep,
// H update of the view's cells.  finalize: append a loss row (the reduce block must then describe the FULL shard).
static int phase2(alpine_ctx* c, const CellView& v, bool update, bool finalize)
{
    int rc;
    const int KP = c->KP, K = c->K;
    const float* HHt = c->red + c->red_hht;
    const bool mu = update && !c->use_als;
    const bool fused_w = mu && c->fused_w;
    // MU: this launch also updates W (and, fused, emits the partial blocks of W_new^T W_new); block-coordinate: dot partials
    // only (the group loop below updates W)
    if ((rc = launch_w_update(c, HHt, mu, 0, K, false, fused_w ? c->gramPart : nullptr))) return rc;
    if (finalize && c->loss_rows == c->loss_cap && (rc = grow_losses(c))) return rc;
    double* loss_row = c->loss_dev + c->loss_rows * (c->n_cov + 2);
    if (fused_w) {
        // ... then ONE launch: partial blocks -> the other W^T W buffer || pending loss row (current buffer) || B updates
        Phase2Tail a{};
        a.nb_slabs = (KP * KP + 15) / 16; a.n_slab = (c->G + 127) / 128;
        a.do_loss = finalize ? 1 : 0; a.do_b = c->n_cov > 0 ? 1 : 0;
        float* other = c->WtW == c->WtWbuf[0] ? c->WtWbuf[1] : c->WtWbuf[0];
        a.gram_part = c->gramPart; a.WtW_new = other;
        a.dotpart = c->dotpart; a.ndot = c->ndot; a.WtW_old = c->WtW; a.HHt = HHt; a.stats = c->red + c->red_stats; a.nstat = c->nstat;
        a.lam64 = c->lam_dev; a.row = loss_row;
        a.Bold = c->B[c->bcur]; a.Bnew = c->B[c->bcur ^ 1]; a.eps = (float)c->eps;
        hipLaunchKernelGGL(phase2_tail_kernel, dim3(a.nb_slabs + 2), dim3(256), 0, c->stream, a, c->meta, KP);
        HIPCHK(c, hipGetLastError());
        c->WtW = other;
        if (finalize) c->loss_rows++;
        if (c->n_cov > 0) c->bcur ^= 1;
        if ((rc = launch_sweep_wtx(c, v))) return rc;
        return launch_h_update(c, v, 0, K, -1, v.H == c->H && !c->no_tail);
    }
    if (update && !c->use_als && !c->unfused_mid) {
        // MU branch: gram(W_new) partials, the pending loss row and the B updates share one launch (phase2_mid_kernel)
        const int rpw = gram_rows_per_wave(c->Gp, c->n_cu);
        Phase2Mid a{};
        a.gram_blocks = (int)((c->Gp + 4 * rpw - 1) / (4 * rpw));
        a.do_loss = finalize ? 1 : 0; a.do_b = c->n_cov > 0 ? 1 : 0;
        a.dotpart = c->dotpart; a.ndot = c->ndot; a.WtW = c->WtW; a.HHt = HHt; a.stats = c->red + c->red_stats; a.nstat = c->nstat;
        a.lam64 = c->lam_dev; a.row = loss_row;
        a.Bold = c->B[c->bcur]; a.Bnew = c->B[c->bcur ^ 1]; a.eps = (float)c->eps;
        DISPATCH_KT(c->KT, hipLaunchKernelGGL(phase2_mid_kernel<KT_>, dim3(a.gram_blocks + 2), dim3(256), 0, c->stream, c->W, c->gramPart,
                                               (int)c->Gp, rpw, a, c->meta));
        HIPCHK(c, hipGetLastError());
        if (finalize) c->loss_rows++;
        if (c->n_cov > 0) c->bcur ^= 1;
        const int n = KP * KP;
        hipLaunchKernelGGL(reduce_many_kernel, dim3((n + 63) / 64), dim3(1024), 0, c->stream, c->gramPart, c->WtW, n, a.gram_blocks);
        HIPCHK(c, hipGetLastError());
        if ((rc = launch_sweep_wtx(c, v))) return rc;
        return launch_h_update(c, v, 0, K, -1, v.H == c->H && !c->no_tail);
    }
    if (finalize) {
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, c->stream, c->dotpart, c->ndot, c->WtW, HHt,
                           c->red + c->red_stats, c->meta, c->nstat, KP, c->lam_dev, loss_row);
        HIPCHK(c, hipGetLastError());
        c->loss_rows++;
    }
    if (!update) return 0;

    // every B_i depends only on its own old B_i and old H_i, so all of them can be updated up front in both branches
    if (c->n_cov > 0) {
        hipLaunchKernelGGL(b_update_kernel, dim3(1), dim3(256), 0, c->stream, c->B[c->bcur], c->B[c->bcur ^ 1], c->red + c->red_stats,
                           HHt, c->meta, KP, (float)c->eps);
        HIPCHK(c, hipGetLastError());
        c->bcur ^= 1;
    }
    if (!c->use_als) {
        if ((rc = launch_gram(c, c->W, c->Gp, c->gramBlocksW, c->WtW))) return rc;
        if ((rc = launch_sweep_wtx(c, v))) return rc;
        return launch_h_update(c, v, 0, K, -1);
    }
    // block-coordinate branch, main.py:525-588: groups in the order [cov_1 .. cov_C, unguided]
    for (int grp = 0; grp <= c->n_cov; ++grp) {
        if ((rc = als_group_hht(c, v, grp))) return rc;
        if ((rc = als_group_update(c, v, grp))) return rc;
    }
    return 0;
}


// ---------------------------------------------------------------------------------- wide models (128 < K <= 1024, kernels_wide.hpp)
static float* wide_half(float* base, int64_t rows_pad, int h) { return base + (int64_t)h * rows_pad * WIDE_KH; }
static float* wide_block(const alpine_ctx* c, float* base, int a, int b) { return base + (int64_t)(a * c->NH + b) * WIDE_KH * WIDE_KH; }
// 16-component tiles of half h that hold real components (0 = all of them): only the last half is partly filled
static int wide_active16(const alpine_ctx* c, int h) { return h < c->NH - 1 ? 0 : ((c->K - (c->NH - 1) * WIDE_KH + 15) / 16) % 8; }

// out (blocked [NH][NH][128][128]) = A^T A for a blocked A [NH][R][128]
static int launch_gram_wide(alpine_ctx* c, float* A, int64_t R, float* out)
{
    const int rpw = gram_rows_per_wave(R, c->n_cu);
    const int blocks = (int)((R + 4 * rpw - 1) / (4 * rpw));
    if (blocks > c->gramPart_cap) return fail(c, ALPINE_ERR_STATE, "internal: Gram partial buffer too small");
    const int n = WIDE_KH * WIDE_KH;
    const int NH = c->NH;
    int tiles[WIDE_MAX_NH];                                               // 32-component tiles with real components, per half
    for (int h = 0; h < NH; ++h) tiles[h] = h < NH - 1 ? WIDE_KT : (c->K - (NH - 1) * WIDE_KH + 31) / 32;
    for (int a = 0; a < NH; ++a)
        for (int b = a; b < NH; ++b) {
#define GRAM_X(TA_, TB_) hipLaunchKernelGGL((gram_cross_kernel<WIDE_KT, TA_, TB_>), dim3(blocks), dim3(256), 0, c->stream, wide_half(A, R, a), wide_half(A, R, b), c->gramPart, (int)R, rpw)
            // (a <= b and only the last half is partly filled: the tile counts are (4, 4), (4, t) or (t, t))
            if (tiles[a] == WIDE_KT) { switch (tiles[b]) { case 1: GRAM_X(4, 1); break; case 2: GRAM_X(4, 2); break; case 3: GRAM_X(4, 3); break; default: GRAM_X(4, 4); break; } }
            else { switch (tiles[b]) { case 1: GRAM_X(1, 1); break; case 2: GRAM_X(2, 2); break; default: GRAM_X(3, 3); break; } }
#undef GRAM_X
            hipLaunchKernelGGL(reduce_many_kernel, dim3((n + 63) / 64), dim3(1024), 0, c->stream, c->gramPart, wide_block(c, out, a, b), n, blocks);
            // block (b, a) = block (a, b) transposed (the same products summed in the same order)
            if (b > a) hipLaunchKernelGGL(transpose_kernel, dim3(WIDE_KH / 32, WIDE_KH / 32), dim3(256), 0, c->stream, wide_block(c, out, a, b), (int64_t)WIDE_KH,
                                          wide_block(c, out, b, a), (int64_t)WIDE_KH, WIDE_KH, WIDE_KH);
        }
    HIPCHK(c, hipGetLastError());
    return 0;
}

static int launch_wide_den(alpine_ctx* c, const float* A, int64_t rows_pad, const float* G, int mode, int k_lo, int k_hi, bool block_orth)
{
    WideDenArgs a{};
    a.rows_pad = (int)rows_pad; a.K = c->K; a.mode = mode;
    a.orth = (float)c->orth; a.l2 = (float)((1.0 - c->l1r) * c->alpha);
    a.k_lo = k_lo; a.k_hi = k_hi; a.block_orth = block_orth ? 1 : 0; a.nh = c->NH;
    if (rows_pad > c->wide_den_rows) return fail(c, ALPINE_ERR_STATE, "internal: the blocked update has %lld rows, its buffer holds %lld", (long long)rows_pad, (long long)c->wide_den_rows);
    const size_t lds = sizeof(float) * (WIDE_KH * WIDE_KH + 4 * 32 * (WIDE_KH + 4));
    if (lds > c->lds_dev) return fail(c, ALPINE_ERR_UNSUPPORTED, "internal: the blocked update needs %zu bytes of LDS, the device has %zu", lds, c->lds_dev);
    hipLaunchKernelGGL(wide_den_kernel, dim3((unsigned)(rows_pad / 128)), dim3(256), lds, c->stream, A, G, c->wide_den, a);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// The sweep of a wide model over all its components: pieces of component half h land in pieces + h * cap / NH (each [.][128] with the
// geometry g).  x3 at K <= 224: ONE launch of stream_gemm_x3w2_kernel (X read once); else one launch per half of the blocked panel P.
static int launch_sweeps_wide(alpine_ctx* c, const SweepGeom& g, const float* S, float* P, int64_t rows_pad, float* pieces, int64_t cap, int which)
{
    int rc;
    const int evt = which == 0 ? ALPINE_KERNEL_SWEEP_XHT : ALPINE_KERNEL_SWEEP_WTX;
    if (c->wide_one_pass) {
        const int64_t need = (int64_t)g.nwg * g.maxp * g.bf * WIDE_KH;
        if (need > cap / 2 || (which == 0 && c->transform_only))
            return fail(c, ALPINE_ERR_STATE, "internal: sweep %d needs %lld floats of pieces per half, the buffer holds %lld", which, (long long)need, (long long)(cap / 2));
        if (g.bf != (c->wide_two_wave ? 512 : 256) * g.gw)
            return fail(c, ALPINE_ERR_STATE, "internal: the one-pass wide sweep was divided for %d-column workgroup tiles, its kernel works on %d", g.bf / std::max(1, g.gw), c->wide_two_wave ? 512 : 256);
        if (c->NH != 2) return fail(c, ALPINE_ERR_STATE, "internal: the one-pass wide sweep covers two component halves, the model has %d", c->NH);
        int* xcc_out = c->probe_placement ? c->xcc_dev : nullptr;
        const int64_t ldS = g.F;
        const int m16 = (c->K + 15) / 16;                     // 9..16 tiles with real components; instantiated for the even counts
        // the panel as three exact bf16 planes, once per sweep (every workgroup tile is only 256 columns wide: splitting it there costs
        // as much vector work as the X split itself)
        if (rows_pad > c->wide_den_rows || rows_pad % 8) return fail(c, ALPINE_ERR_STATE, "internal: panel of %lld rows, the plane buffer holds %lld", (long long)rows_pad, (long long)c->wide_den_rows);
        const int m16i = m16 <= 10 ? 10 : (m16 <= 12 ? 12 : (m16 <= 14 ? 14 : 16));        // the instantiation that runs
        const int kpa = 16 * m16i;                             // components the sweep stages and multiplies
        const int64_t plane_stride = (rows_pad / 8) * kpa;
        hipLaunchKernelGGL(pack_panel3_wide_kernel, dim3((unsigned)std::min<int64_t>((plane_stride + 255) / 256, (int64_t)c->n_cu * 16)), dim3(256), 0, c->stream,
                           wide_half(P, rows_pad, 0), wide_half(P, rows_pad, 1), (int)rows_pad, kpa, c->wide_panel3, plane_stride);
        HIPCHK(c, hipGetLastError());
        if ((rc = prof_begin(c, evt))) return rc;
        const bool one_plane = c->x_one_plane && c->x3_variant < 0;
#define X3W2(M_) do { if (one_plane) hipLaunchKernelGGL((stream_gemm_x3w2_kernel<M_, true>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, c->wide_panel3, plane_stride, \
                                                        pieces, pieces + cap / 2, ldS, g, xcc_out); \
                      else hipLaunchKernelGGL((stream_gemm_x3w2_kernel<M_, false>), dim3(sweep_grid(g)), dim3(256), 0, c->stream, S, c->wide_panel3, plane_stride, \
                                              pieces, pieces + cap / 2, ldS, g, xcc_out); } while (0)
        if (c->wide_two_wave)
            hipLaunchKernelGGL((stream_gemm_x3w2_kernel<10, true, 8>), dim3(sweep_grid(g)), dim3(512), 0, c->stream, S, c->wide_panel3, plane_stride, pieces, pieces + cap / 2, ldS, g, xcc_out);
        else if (m16 <= 10) X3W2(10); else if (m16 <= 12) X3W2(12); else if (m16 <= 14) X3W2(14); else X3W2(16);
#undef X3W2
        HIPCHK(c, hipGetLastError());
        return prof_end(c, evt);
    }
    for (int h = 0; h < c->NH; ++h) {
        if ((rc = prof_begin(c, evt))) return rc;
        if ((rc = launch_sweep(c, g, S, wide_half(P, rows_pad, h), pieces + h * (cap / c->NH), which, wide_active16(c, h)))) return rc;
        if ((rc = prof_end(c, evt))) return rc;
    }
    return 0;
}

static int phase1_wide(alpine_ctx* c, const CellView& v)
{
    int rc;
    c->prof_now = c->prof && (c->prof_tick++ % c->prof_every) == 0;
    int max_k, max_ct;
    cov_maxima(c, &max_k, &max_ct);
    if (v.statBlocks > c->statPart_cap) return fail(c, ALPINE_ERR_STATE, "internal: %d statistics blocks, the buffer holds %lld", v.statBlocks, (long long)c->statPart_cap);
    if (c->n_cov > 0) {
        hipLaunchKernelGGL(hstats_kernel, dim3(v.statBlocks), dim3(HS_CELLS), hstats_group_bytes(max_k, max_ct), c->stream, v.H, v.Y, c->B[c->bcur],
                           c->meta, c->statPart, v.N, v.Np, WIDE_KH, (float)c->eps, c->nstat, max_k, max_ct);         // guided components: first half
        HIPCHK(c, hipGetLastError());
    }
    hipLaunchKernelGGL(reduce_stats_kernel, dim3(c->nstat + 1), dim3(256), 0, c->stream, c->statPart, c->kind, c->red + c->red_stats,
                       v.statBlocks, c->nstat, c->xnorm2);
    HIPCHK(c, hipGetLastError());
    if ((rc = launch_gram_wide(c, v.H, v.Np, c->red + c->red_hht))) return rc;
    if ((rc = launch_sweeps_wide(c, v.gA, v.Xng, v.H, v.Np, c->piecesA, c->piecesA_cap, 0))) return rc;
    for (int h = 0; h < c->NH; ++h)
        if ((rc = launch_reduce_pieces(c, c->piecesA + h * (c->piecesA_cap / c->NH), wide_half(c->red, c->Gp, h), (int)c->Gp, v.gA, WIDE_KH))) return rc;
    return 0;
}

// H[:, k_lo..k_hi) of the view from den (already in wide_den) and the W^TX pieces (or num_in); only_cov as in launch_h_update
static int wide_h_apply(alpine_ctx* c, const CellView& v, const float* num_in, int k_lo, int k_hi, int only_cov)
{
    const int blocks = (int)std::min<int64_t>((int64_t)c->n_cu * 16, ((int64_t)v.N + 3) / 4);
    const int64_t pstride = c->piecesB_cap / c->NH;
    if (c->loss_type == ALPINE_LOSS_KL)
        hipLaunchKernelGGL(wide_h_apply_kernel<0>, dim3(blocks), dim3(256), 0, c->stream, v.H, c->wide_den, c->piecesB, pstride, c->NH, v.gB, num_in, v.Y, c->B[c->bcur], c->meta,
                           v.N, v.Np, c->K, (float)c->eps, k_lo, k_hi, only_cov);
    else
        hipLaunchKernelGGL(wide_h_apply_kernel<1>, dim3(blocks), dim3(256), 0, c->stream, v.H, c->wide_den, c->piecesB, pstride, c->NH, v.gB, num_in, v.Y, c->B[c->bcur], c->meta,
                           v.N, v.Np, c->K, (float)c->eps, k_lo, k_hi, only_cov);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// W[:, k_lo..k_hi) from the reduce block (update) + the float64 partials of <XH^T, W_old> (always)
static int wide_w_step(alpine_ctx* c, bool update, int k_lo, int k_hi, bool block_orth)
{
    int rc;
    const float l1 = (float)(c->l1r * c->alpha);
    if (update && (rc = launch_wide_den(c, c->W, c->Gp, c->red + c->red_hht, 0, k_lo, k_hi, block_orth))) return rc;
    hipLaunchKernelGGL(wide_w_apply_kernel, dim3(c->ndot), dim3(256), 0, c->stream, c->W, c->red, c->wide_den, c->dotpart, c->G, c->Gp, c->K, l1,
                       (float)c->eps, update ? 1 : 0, k_lo, k_hi, c->NH);
    HIPCHK(c, hipGetLastError());
    return 0;
}

// W^TW of the current W, the two W^TX sweeps over the view, den = H . 2W^TW, then the H update of [k_lo, k_hi)
static int wide_h_step(alpine_ctx* c, const CellView& v, int k_lo, int k_hi, int only_cov)
{
    int rc;
    if ((rc = launch_gram_wide(c, c->W, c->Gp, c->WtW))) return rc;
    if ((rc = launch_sweeps_wide(c, v.gB, v.Xgn, c->W, c->Gp, c->piecesB, c->piecesB_cap, 1))) return rc;
    if ((rc = launch_wide_den(c, v.H, v.Np, c->WtW, 1, 0, c->K, false))) return rc;
    return wide_h_apply(c, v, nullptr, k_lo, k_hi, only_cov);
}

static int als_group_update_wide(alpine_ctx* c, const CellView& v, int grp, int k_lo, int k_hi)
{
    int rc;
    if ((rc = wide_w_step(c, true, k_lo, k_hi, true))) return rc;
    return wide_h_step(c, v, k_lo, k_hi, grp);
}

static int grow_losses(alpine_ctx* c);

// pending loss row (from the factors that produced the reduce block) and all B updates: what precedes the H side in both branches
static int wide_loss_and_b(alpine_ctx* c, bool update, bool finalize)
{
    int rc;
    float* HHt = c->red + c->red_hht;
    if (finalize) {
        if (c->loss_rows == c->loss_cap && (rc = grow_losses(c))) return rc;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, c->stream, c->dotpart, c->ndot, c->WtW, HHt, c->red + c->red_stats, c->meta,
                           c->nstat, c->KP, c->lam_dev, c->loss_dev + c->loss_rows * (c->n_cov + 2));
        HIPCHK(c, hipGetLastError());
        c->loss_rows++;
    }
    if (update && c->n_cov > 0) {
        hipLaunchKernelGGL(b_update_kernel, dim3(1), dim3(256), 0, c->stream, c->B[c->bcur], c->B[c->bcur ^ 1], c->red + c->red_stats,
                           wide_block(c, HHt, 0, 0), c->meta, WIDE_KH, (float)c->eps);                                    // guided components: block (0, 0)
        HIPCHK(c, hipGetLastError());
        c->bcur ^= 1;
    }
    return 0;
}

// [loss row of the factors that produced the reduce block], W update, B updates, W^TW, W^TX sweeps, H update: the unfused sequence of
// phase2 with every K-templated kernel replaced by its blocked form
static int phase2_wide(alpine_ctx* c, const CellView& v, bool update, bool finalize)
{
    int rc;
    c->tail_valid = false;
    if ((rc = wide_w_step(c, update && !c->use_als, 0, c->K, false))) return rc;
    if ((rc = wide_loss_and_b(c, update, finalize))) return rc;
    if (!update) return 0;
    if (!c->use_als) return wide_h_step(c, v, 0, c->K, -1);
    for (int grp = 0; grp <= c->n_cov; ++grp) {          // block-coordinate branch, main.py:525-588
        if ((rc = als_group_hht(c, v, grp))) return rc;
        if ((rc = als_group_update(c, v, grp))) return rc;
    }
    return 0;
}

static int transform_wide(alpine_ctx* c, int n_iter)
{
    int rc;
    if ((rc = launch_gram_wide(c, c->W, c->Gp, c->WtW))) return rc;
    if ((rc = launch_sweeps_wide(c, c->geomB, c->Xgn, c->W, c->Gp, c->piecesB, c->piecesB_cap, 1))) return rc;
    const int blocks = (int)std::min<int64_t>((int64_t)c->n_cu * 16, ((int64_t)c->N + 3) / 4);
    hipLaunchKernelGGL(wide_num_kernel, dim3(blocks), dim3(256), 0, c->stream, c->wide_num, c->piecesB, c->piecesB_cap / c->NH, c->NH, c->geomB, c->N, c->Np);
    HIPCHK(c, hipGetLastError());
    for (int it = 0; it < n_iter; ++it) {
        if ((rc = launch_wide_den(c, c->H, c->Np, c->WtW, 1, 0, c->K, false))) return rc;
        if ((rc = wide_h_apply(c, c->full, c->wide_num, 0, c->K, c->n_cov))) return rc;      // only_cov = n_cov: no guided terms
    }
    return 0;
}

extern "C" int alpine_iter_begin(alpine_ctx* c)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only) return fail(c, ALPINE_ERR_STATE, "ctx was created with ALPINE_FLAG_TRANSFORM_ONLY");
    return c->wide ? phase1_wide(c, c->full) : phase1(c, c->full);
}

extern "C" int alpine_iter_end(alpine_ctx* c, int update)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only) return fail(c, ALPINE_ERR_STATE, "ctx was created with ALPINE_FLAG_TRANSFORM_ONLY");
    rc = c->wide ? phase2_wide(c, c->full, update != 0, c->pending_loss && c->loss_enabled)
                 : phase2(c, c->full, update != 0, c->pending_loss && c->loss_enabled);
    if (rc) return rc;
    c->pending_loss = update != 0;
    return 0;
}

// use_als iteration split at ITS exchange points (the group loop needs H H^T of all cells after every group):
//   alpine_iter_begin -> [all-reduce the reduce block] -> alpine_als_begin          (pending loss row, all B updates)
//   for grp in 0..C:  alpine_als_group_begin(grp) -> [grp > 0: all-reduce the H H^T slot] -> alpine_als_group_end(grp)
// alpine_iter_end(ctx, 1) is exactly this sequence without the exchanges (single shard).
extern "C" int alpine_als_begin(alpine_ctx* c)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only || !c->use_als) return fail(c, ALPINE_ERR_STATE, "alpine_als_begin needs a ctx created with ALPINE_FLAG_USE_ALS");
    const float* HHt = c->red + c->red_hht;
    const bool finalize = c->pending_loss && c->loss_enabled;
    if (c->wide) {
        if ((rc = wide_w_step(c, false, 0, c->K, false))) return rc;
        if ((rc = wide_loss_and_b(c, true, finalize))) return rc;
        c->pending_loss = true;
        return 0;
    }
    if ((rc = launch_w_update(c, HHt, false, 0, c->K, false))) return rc;        // dot partials of <XH^T, W_old> only
    if (finalize) {
        if (c->loss_rows == c->loss_cap && (rc = grow_losses(c))) return rc;
        hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, c->stream, c->dotpart, c->ndot, c->WtW, HHt,
                           c->red + c->red_stats, c->meta, c->nstat, c->KP, c->lam_dev, c->loss_dev + c->loss_rows * (c->n_cov + 2));
        HIPCHK(c, hipGetLastError());
        c->loss_rows++;
    }
    if (c->n_cov > 0) {
        hipLaunchKernelGGL(b_update_kernel, dim3(1), dim3(256), 0, c->stream, c->B[c->bcur], c->B[c->bcur ^ 1], c->red + c->red_stats,
                           HHt, c->meta, c->KP, (float)c->eps);
        HIPCHK(c, hipGetLastError());
        c->bcur ^= 1;
    }
    c->pending_loss = true;
    return 0;
}

extern "C" int alpine_als_group_begin(alpine_ctx* c, int grp)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only || !c->use_als) return fail(c, ALPINE_ERR_STATE, "ctx was not created with ALPINE_FLAG_USE_ALS");
    if (grp < 0 || grp > c->n_cov) return fail(c, ALPINE_ERR_BAD_ARG, "group %d outside [0, %d]", grp, c->n_cov);
    return als_group_hht(c, c->full, grp);
}

extern "C" int alpine_als_group_end(alpine_ctx* c, int grp)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only || !c->use_als) return fail(c, ALPINE_ERR_STATE, "ctx was not created with ALPINE_FLAG_USE_ALS");
    if (grp < 0 || grp > c->n_cov) return fail(c, ALPINE_ERR_BAD_ARG, "group %d outside [0, %d]", grp, c->n_cov);
    return als_group_update(c, c->full, grp);
}

// ---------------------------------------------------------------------------------- multi-GPU (RCCL over xGMI)
// One process per GPU, one ctx per process; the cell axis is sharded over the ranks of the communicator.  The only data
// that crosses shards is the reduce block (one sum all-reduce per iteration), enqueued on the ctx stream
// right behind the kernels that fill it -- no host synchronisation, no second stream.
extern "C" int alpine_comm_get_unique_id(void* id_out)
{
    if (!id_out) return fail(nullptr, ALPINE_ERR_BAD_ARG, "id_out is NULL");
    static_assert(sizeof(ncclUniqueId) == ALPINE_COMM_ID_BYTES, "ALPINE_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId id;
    NCCLCHK(nullptr, ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int alpine_comm_version(int* version_out)
{
    if (!version_out) return fail(nullptr, ALPINE_ERR_BAD_ARG, "version_out is NULL");
    NCCLCHK(nullptr, ncclGetVersion(version_out));
    return 0;
}

extern "C" int alpine_comm_init_rank(alpine_ctx* c, const void* id, int nranks, int rank)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!id || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, ALPINE_ERR_BAD_ARG, "bad communicator arguments (nranks %d, rank %d)", nranks, rank);
    if (c->comm) return fail(c, ALPINE_ERR_STATE, "the ctx already has a communicator");
    HIPCHK(c, hipSetDevice(c->device));
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    NCCLCHK(c, ncclCommInitRank(&c->comm, nranks, uid, rank));
    c->comm_ranks = nranks; c->comm_rank = rank;
    return 0;
}

// What the ATTACHED communicator itself says (ncclCommCount / ncclCommUserRank), not what alpine_comm_init_rank was told: a run report
// that prints these shows that RCCL saw N ranks (bench.py fails the run when they disagree with the launch).
extern "C" int alpine_comm_count(alpine_ctx* c, int* nranks_out, int* rank_out)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!c->comm) return fail(c, ALPINE_ERR_STATE, "no communicator attached (alpine_comm_init_rank / alpine_comm_init_all)");
    int n = -1, r = -1;
    NCCLCHK(c, ncclCommCount(c->comm, &n));
    NCCLCHK(c, ncclCommUserRank(c->comm, &r));
    if (nranks_out) *nranks_out = n;
    if (rank_out) *rank_out = r;
    return 0;
}

// ONE process, n ctxs on n different GPUs (the single-process drop-in of SURVEY.md 8b: ALPINE(devices=[...]), examples/fit_c --devices):
// ncclCommInitAll over the ctxs' devices, communicator i attached to ctxs[i] as rank i.  No unique id, no launcher.  Each ctx is then
// driven by its own host thread (alpine_run & co. enqueue the all-reduce on the ctx's stream; the ranks meet inside RCCL).
extern "C" int alpine_comm_init_all(alpine_ctx* const* ctxs, int n)
{
    if (!ctxs || n < 1 || n > 64) return fail(nullptr, ALPINE_ERR_BAD_ARG, "alpine_comm_init_all: bad arguments (n = %d)", n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail(nullptr, ALPINE_ERR_BAD_ARG, "alpine_comm_init_all: ctxs[%d] is NULL", i);
        if (ctxs[i]->comm) return fail(ctxs[i], ALPINE_ERR_STATE, "the ctx already has a communicator");
    }
    // (two ctxs on one device: RCCL itself refuses that -- its error text comes back through ALPINE_ERR_RCCL)
    std::vector<int> devs(n);
    std::vector<ncclComm_t> comms(n, nullptr);
    for (int i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
    {
        const ncclResult_t r = ncclCommInitAll(comms.data(), n, devs.data());
        if (r != ncclSuccess) {
            for (int i = 0; i < n; ++i) fail(ctxs[i], ALPINE_ERR_RCCL, "ncclCommInitAll over %d device(s) failed: %s", n, ncclGetErrorString(r));
            return fail(nullptr, ALPINE_ERR_RCCL, "ncclCommInitAll over %d device(s) failed: %s", n, ncclGetErrorString(r));
        }
    }
    for (int i = 0; i < n; ++i) { ctxs[i]->comm = comms[i]; ctxs[i]->comm_ranks = n; ctxs[i]->comm_rank = i; }
    return 0;
}

extern "C" int alpine_comm_destroy(alpine_ctx* c)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!c->comm) return 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    NCCLCHK(c, ncclCommDestroy(c->comm));
    c->comm = nullptr; c->comm_ranks = 1; c->comm_rank = 0;
    return 0;
}

// in-place sum all-reduce of red[off, off + n) on the ctx stream; a no-op without a communicator
static int comm_all_reduce(alpine_ctx* c, int64_t off, int64_t n)
{
    if (!c->comm) return 0;
    int rc;
    if ((rc = prof_begin(c, ALPINE_KERNEL_ALLREDUCE))) return rc;
    NCCLCHK(c, ncclAllReduce(c->red + off, c->red + off, (size_t)n, ncclFloat, ncclSum, c->comm, c->stream));
    return prof_end(c, ALPINE_KERNEL_ALLREDUCE);
}

extern "C" int alpine_comm_all_reduce(alpine_ctx* c, int64_t offset_floats, int64_t n_floats)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (!c->comm) return fail(c, ALPINE_ERR_STATE, "no communicator attached (alpine_comm_init_rank)");
    if (offset_floats < 0 || n_floats < 0 || offset_floats + n_floats > c->red_floats)
        return fail(c, ALPINE_ERR_BAD_ARG, "range outside the reduce block (%lld floats)", (long long)c->red_floats);
    HIPCHK(c, hipSetDevice(c->device));
    return comm_all_reduce(c, offset_floats, n_floats);
}

// One whole iteration INCLUDING its exchanges when a communicator is attached (identical to alpine_iter_begin +
// alpine_iter_end without one): begin -> all-reduce -> end; block-coordinate branch: + the K x K slot after every group.
extern "C" int alpine_iter(alpine_ctx* c, int update)
{
    int rc = alpine_iter_begin(c);
    if (rc) return rc;
    if ((rc = comm_all_reduce(c, 0, c->red_floats))) return rc;
    if (!update || !c->use_als || !c->comm) return alpine_iter_end(c, update);
    if ((rc = alpine_als_begin(c))) return rc;
    for (int grp = 0; grp <= c->n_cov; ++grp) {
        if ((rc = alpine_als_group_begin(c, grp))) return rc;
        if (grp > 0 && (rc = comm_all_reduce(c, c->red_hht, (int64_t)c->KP * c->KP))) return rc;
        if ((rc = alpine_als_group_end(c, grp))) return rc;
    }
    return 0;
}

// where the K x K H H^T slot sits inside the reduce block (for the per-group exchange of the use_als branch)
extern "C" int alpine_reduce_block_hht(alpine_ctx* c, int64_t* offset_floats, int64_t* n_floats)
{
    if (!c || !offset_floats || !n_floats) return fail(c, ALPINE_ERR_BAD_ARG, "NULL argument");
    *offset_floats = c->red_hht;
    *n_floats = (int64_t)c->KP * c->KP;
    return 0;
}

// One mini-batch update (main.py:512-663 for one batch): the n cells idx[0..n) (local indices, duplicates allowed:
// "weighted" sampling draws with replacement) are gathered into a contiguous view, the same phase 1 / phase 2 kernels
// run on the view, and the updated rows of H are scattered back.  Split at the exchange point like an iteration:
//   alpine_batch_begin : gather + sums over the batch's LOCAL cells -> reduce block   (n == 0 is allowed: a shard
//                        that holds none of the batch's cells contributes zeros and still updates its replica of W, B)
//   [ multi-GPU: the caller all-reduces the reduce block ]
//   alpine_batch_end   : W, B updates, W^TX sweep and H update on the view, scatter
extern "C" int alpine_batch_begin(alpine_ctx* c, const int64_t* idx, int64_t n)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only || c->bf16) return fail(c, ALPINE_ERR_UNSUPPORTED, "mini-batches need the float32 two-copy layout");
    if (c->batch_open) return fail(c, ALPINE_ERR_STATE, "alpine_batch_begin: the previous batch was not ended");
    if (c->use_als && c->comm_ranks > 1)
        return fail(c, ALPINE_ERR_UNSUPPORTED, "mini-batches with ALPINE_FLAG_USE_ALS are single-shard (the group loop of a batch has no exchange step)");
    if (n < 0 || n > c->batch_cap || (n > 0 && !idx))
        return fail(c, ALPINE_ERR_BAD_ARG, "batch of %lld cells outside the ctx's batch_capacity %lld", (long long)n, (long long)c->batch_cap);
    c->batch_n = n;
    if (n == 0) {
        HIPCHK(c, hipMemsetAsync(c->red, 0, sizeof(float) * c->red_floats, c->stream));
        c->batch_open = true;
        return 0;
    }
    std::vector<int> h((size_t)n);
    for (int64_t j = 0; j < n; ++j) {
        if (idx[j] < 0 || idx[j] >= c->N) return fail(c, ALPINE_ERR_BAD_ARG, "batch index %lld out of range", (long long)idx[j]);
        h[(size_t)j] = (int)idx[j];
    }
    HOSTCOPY(c, c->idx_dev, h.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice);     // (the drain also ends the previous batch's scatter, which reads idx_dev)
    const int KP = c->KP;
    const int64_t Bp = round_up(n, 128);
    const int nb = c->n_cu * 8;
    // gather the view: rows of the cells x genes copy, rows of H, columns of Y; then the genes x cells copy by transpose
    hipLaunchKernelGGL(gather_rows_kernel, dim3(nb), dim3(256), 0, c->stream, c->Xng, c->Gp, c->idx_dev, (int)n, (int)Bp, c->Xb_ng, c->Gp, (int)c->Gp);
    if (c->wide) {
        for (int h = 0; h < c->NH; ++h)  // blocked factors: each half is a [rows][128] array of its own (the view's halves are Bp rows apart)
            hipLaunchKernelGGL(gather_rows_kernel, dim3(nb), dim3(256), 0, c->stream, wide_half(c->H, c->Np, h), (int64_t)WIDE_KH, c->idx_dev, (int)n, (int)Bp,
                               wide_half(c->Hb, Bp, h), (int64_t)WIDE_KH, WIDE_KH);
    } else {
        hipLaunchKernelGGL(gather_rows_kernel, dim3(nb), dim3(256), 0, c->stream, c->H, (int64_t)KP, c->idx_dev, (int)n, (int)Bp, c->Hb, (int64_t)KP, KP);
    }
    if (c->nYrows > 0)
        hipLaunchKernelGGL(gather_cols_kernel, dim3(nb), dim3(256), 0, c->stream, c->Y, c->Np, c->idx_dev, (int)n, (int)Bp, c->Yb, Bp, c->nYrows);
    {
        dim3 grid((unsigned)((c->Gp + 31) / 32), (unsigned)((Bp + 31) / 32));
        hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, c->stream, c->Xb_ng, c->Gp, c->Xb_gn, Bp, (int)Bp, (int)c->Gp);
    }
    HIPCHK(c, hipGetLastError());
    CellView& v = c->batch_view;
    v.Xgn = c->Xb_gn; v.Xng = c->Xb_ng; v.H = c->Hb; v.Y = c->Yb;
    v.N = (int)n; v.Np = Bp;
    v.gA = make_geom(c, c->Gp, Bp, 0, c->sweep_bf);
    v.gB = make_geom(c, Bp, c->Gp, 0, c->sweep_bf);
    if ((int64_t)v.gA.nwg * v.gA.maxp * v.gA.bf * c->KP > c->piecesA_cap || (int64_t)v.gB.nwg * v.gB.maxp * v.gB.bf * c->KP > c->piecesB_cap)
        return fail(c, ALPINE_ERR_STATE, "internal: batch view needs more sweep pieces than were allocated");
    v.statBlocks = (int)((n + HS_CELLS - 1) / HS_CELLS);
    v.gramBlocksH = (int)((Bp + 4 * GR_ROWS_PER_WAVE - 1) / (4 * GR_ROWS_PER_WAVE));
    {
        const int rpw = gram_rows_per_wave(Bp, c->n_cu);
        if (v.statBlocks > c->statPart_cap || (Bp + 4 * rpw - 1) / (4 * rpw) > c->gramPart_cap)
            return fail(c, ALPINE_ERR_STATE, "internal: batch view needs more partial blocks than were allocated");
    }
    if ((rc = c->wide ? phase1_wide(c, v) : phase1(c, v))) return rc;
    c->batch_open = true;
    return 0;
}

extern "C" int alpine_batch_end(alpine_ctx* c)
{
    int rc = ready(c);
    if (rc) return rc;
    if (!c->batch_open) return fail(c, ALPINE_ERR_STATE, "alpine_batch_end without alpine_batch_begin");
    c->batch_open = false;
    if (c->batch_n == 0) {
        // none of the batch's cells live here: only the replicated updates (W from the reduced sums, every B_i)
        if (c->use_als) return fail(c, ALPINE_ERR_UNSUPPORTED, "empty local batches are not supported with ALPINE_FLAG_USE_ALS");
        const float* HHt = c->red + c->red_hht;
        c->pending_loss = false;
        if (c->wide) {
            if ((rc = wide_w_step(c, true, 0, c->K, false))) return rc;
            return wide_loss_and_b(c, true, false);
        }
        if ((rc = launch_w_update(c, HHt, true, 0, c->K, false))) return rc;
        if (c->n_cov > 0) {
            hipLaunchKernelGGL(b_update_kernel, dim3(1), dim3(256), 0, c->stream, c->B[c->bcur], c->B[c->bcur ^ 1], c->red + c->red_stats,
                               HHt, c->meta, c->KP, (float)c->eps);
            HIPCHK(c, hipGetLastError());
            c->bcur ^= 1;
        }
        return 0;
    }
    if ((rc = c->wide ? phase2_wide(c, c->batch_view, true, false) : phase2(c, c->batch_view, true, false))) return rc;
    const int nb = c->n_cu * 8;
    if (c->wide) {
        for (int h = 0; h < c->NH; ++h)
            hipLaunchKernelGGL(scatter_rows_kernel, dim3(nb), dim3(256), 0, c->stream, wide_half(c->Hb, c->batch_view.Np, h), (int64_t)WIDE_KH, c->idx_dev,
                               (int)c->batch_n, wide_half(c->H, c->Np, h), (int64_t)WIDE_KH, WIDE_KH);
    } else {
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(nb), dim3(256), 0, c->stream, c->Hb, (int64_t)c->KP, c->idx_dev, (int)c->batch_n, c->H, (int64_t)c->KP, c->KP);
    }
    HIPCHK(c, hipGetLastError());
    c->pending_loss = false;
    return 0;
}

extern "C" int alpine_batch_step(alpine_ctx* c, const int64_t* idx, int64_t n)
{
    // single shard: an empty batch is a caller error; with a communicator a rank may hold none of the batch's cells
    if (c && (n < 0 || (n == 0 && !c->comm))) return fail(c, ALPINE_ERR_BAD_ARG, "batch of %lld cells outside the ctx's batch_capacity %lld", (long long)n, (long long)c->batch_cap);
    int rc = alpine_batch_begin(c, idx, n);
    if (rc) return rc;
    if ((rc = comm_all_reduce(c, 0, c->red_floats))) return rc;
    return alpine_batch_end(c);
}

// Loss row of the CURRENT factors over all cells (main.py:666 after the batches of an epoch): full-view phase 1,
// [multi-GPU: all-reduce], <XH^T, W> partials, trace-form finalise.  W^TW is refreshed from the current W.
extern "C" int alpine_epoch_loss_begin(alpine_ctx* c)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only) return fail(c, ALPINE_ERR_STATE, "ctx was created with ALPINE_FLAG_TRANSFORM_ONLY");
    if (c->batch_open) return fail(c, ALPINE_ERR_STATE, "alpine_epoch_loss_begin inside an open batch");
    if (c->wide) {
        if ((rc = launch_gram_wide(c, c->W, c->Gp, c->WtW))) return rc;
        return phase1_wide(c, c->full);
    }
    if ((rc = launch_gram(c, c->W, c->Gp, c->gramBlocksW, c->WtW))) return rc;
    return phase1(c, c->full);
}

extern "C" int alpine_epoch_loss_end(alpine_ctx* c)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->transform_only) return fail(c, ALPINE_ERR_STATE, "ctx was created with ALPINE_FLAG_TRANSFORM_ONLY");
    if ((rc = c->wide ? phase2_wide(c, c->full, false, true) : phase2(c, c->full, false, true))) return rc;
    c->pending_loss = false;
    return 0;
}

extern "C" int alpine_epoch_loss(alpine_ctx* c)
{
    int rc = alpine_epoch_loss_begin(c);
    if (rc) return rc;
    if ((rc = comm_all_reduce(c, 0, c->red_floats))) return rc;
    return alpine_epoch_loss_end(c);
}

extern "C" int alpine_transform(alpine_ctx* c, int n_iter)
{
    int rc = ready(c);
    if (rc) return rc;
    if (n_iter < 0) return fail(c, ALPINE_ERR_BAD_ARG, "n_iter must be >= 0");
    const int KP = c->KP;
    if (c->wide) {
        if ((rc = transform_wide(c, n_iter))) return rc;
        c->pending_loss = false;
        c->tail_valid = false;
        return 0;
    }
    if ((rc = launch_gram(c, c->W, c->Gp, c->gramBlocksW, c->WtW))) return rc;
    if ((rc = prof_begin(c, ALPINE_KERNEL_SWEEP_WTX))) return rc;
    if ((rc = launch_sweep(c, c->geomB, c->Xgn, c->W, c->piecesB, 1))) return rc;
    if ((rc = prof_end(c, ALPINE_KERNEL_SWEEP_WTX))) return rc;
    const int hblocks = (int)((c->N + 127) / 128);
    DISPATCH_KT(c->KT, hipLaunchKernelGGL(h_iterate_mfma_kernel<KT_>, dim3(hblocks), dim3(256), sizeof(float) * KP * KP, c->stream, c->H,
                                           c->piecesB, c->geomB, c->WtW, c->N, c->K, (float)c->eps, n_iter));
    HIPCHK(c, hipGetLastError());
    c->pending_loss = false;
    c->tail_valid = false;
    return 0;
}

extern "C" int alpine_reduce_block(alpine_ctx* c, void** dev_ptr, int64_t* n_floats)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (dev_ptr) *dev_ptr = c->red;
    if (n_floats) *n_floats = c->red_floats;
    return 0;
}

extern "C" int alpine_run(alpine_ctx* c, int n_iters, int with_loss)
{
    int rc = ready(c);
    if (rc) return rc;
    if (n_iters < 0) return fail(c, ALPINE_ERR_BAD_ARG, "n_iters must be >= 0");
    c->loss_enabled = with_loss != 0;
    for (int it = 0; it < n_iters; ++it)
        if ((rc = alpine_iter(c, 1))) return rc;
    if (with_loss && c->pending_loss && (rc = alpine_iter(c, 0))) return rc;
    c->loss_enabled = true;
    return 0;
}

extern "C" int alpine_get_losses(alpine_ctx* c, double* rows, int64_t max_rows, int64_t* n_rows)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int64_t n = std::min(max_rows, c->loss_rows);
    if (rows && n > 0) HOSTCOPY(c, rows, c->loss_dev, sizeof(double) * n * (c->n_cov + 2), hipMemcpyDeviceToHost);
    if (n_rows) *n_rows = c->loss_rows;
    return 0;
}

extern "C" int alpine_reset_losses(alpine_ctx* c)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    c->loss_rows = 0;
    return 0;
}

extern "C" int alpine_scale(alpine_ctx* c)
{
    int rc = ready(c);
    if (rc) return rc;
    const int rows_per_block = 256;
    const int nblk = (c->G + rows_per_block - 1) / rows_per_block;
    const int halves = c->wide ? c->NH : 1, KP = c->wide ? WIDE_KH : c->KP;      // (wide: one half of the blocked factors at a time)
    if ((int64_t)nblk * KP > c->f64part_n) return fail(c, ALPINE_ERR_UNSUPPORTED, "too many genes for the scaling scratch");
    for (int h = 0; h < halves; ++h) {
        const int K = std::min(KP, c->K - h * KP);
        float* Wh = c->W + (int64_t)h * c->Gp * KP;
        float* Hh = c->H + (int64_t)h * c->Np * KP;
        float* sc = c->scale + h * KP;
        hipLaunchKernelGGL(colsum_part_kernel, dim3(nblk), dim3(256), 0, c->stream, Wh, KP, c->G, rows_per_block, c->f64part);
        hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, c->stream, c->f64part, nblk, KP, sc);
        const int64_t nw = (int64_t)c->G * K, nh = (int64_t)c->N * K;
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, c->stream, Wh, KP, (int64_t)c->G, K, sc, 1);
        hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((nh + 255) / 256)), dim3(256), 0, c->stream, Hh, KP, (int64_t)c->N, K, sc, 0);
    }
    if (c->n_cov > 0) hipLaunchKernelGGL(scale_b_kernel, dim3(1), dim3(256), 0, c->stream, c->B[c->bcur], c->meta, c->scale);
    HIPCHK(c, hipGetLastError());
    c->pending_loss = false;      // W^TW / reduce terms no longer describe these factors
    c->tail_valid = false;
    return 0;
}

extern "C" int alpine_synchronize(alpine_ctx* c)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int alpine_eval_recon_direct(alpine_ctx* c, double* out)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->bf16) return fail(c, ALPINE_ERR_UNSUPPORTED, "direct evaluation needs a float32 copy of X");
    if (!out) return fail(c, ALPINE_ERR_BAD_ARG, "out is NULL");
    // sum_n sum_g (X[n][g] - sum_k W[g][k] H[n][k])^2 is symmetric in (genes, W) <-> (cells, H): a transform-only ctx
    // keeps only the genes x cells copy and evaluates the same kernel with the roles swapped
    const bool swap = c->transform_only;
    const int A = swap ? c->N : c->G, Bn = swap ? c->G : c->N;       // A: thread axis (contiguous in X), Bn: looped axis
    const int gx = (A + 255) / 256;
    int cells_per_block = 256;
    while ((int64_t)gx * ((Bn + cells_per_block - 1) / cells_per_block) > c->f64part_n) cells_per_block *= 2;
    const int gy = (Bn + cells_per_block - 1) / cells_per_block;
    if (c->wide) {
        if (swap) hipLaunchKernelGGL(eval_recon_wide_kernel, dim3(gx, gy), dim3(256), 0, c->stream, c->Xgn, c->Np, c->H, c->Np, c->W, c->Gp, A, Bn, cells_per_block, c->f64part, c->NH);
        else hipLaunchKernelGGL(eval_recon_wide_kernel, dim3(gx, gy), dim3(256), 0, c->stream, c->Xng, c->Gp, c->W, c->Gp, c->H, c->Np, A, Bn, cells_per_block, c->f64part, c->NH);
        HIPCHK(c, hipGetLastError());
        return sum_f64_partials(c, gx * gy, out);
    }
    if (swap) {
        DISPATCH_KT(c->KT, hipLaunchKernelGGL(eval_recon_kernel<KT_>, dim3(gx, gy), dim3(256), 0, c->stream, c->Xgn, c->Np, c->H, c->W,
                                               A, Bn, cells_per_block, c->f64part));
    } else {
        DISPATCH_KT(c->KT, hipLaunchKernelGGL(eval_recon_kernel<KT_>, dim3(gx, gy), dim3(256), 0, c->stream, c->Xng, c->Gp, c->W, c->H,
                                               A, Bn, cells_per_block, c->f64part));
    }
    HIPCHK(c, hipGetLastError());
    return sum_f64_partials(c, gx * gy, out);
}

// Diagnostics (tools/graph_vs_eager.py): the steady-state MU iteration as a hipGraph.  TWO iterations are captured (the W^TW and
// B double buffers flip once per iteration, so the launch parameters repeat with period 2), without loss rows (their row
// pointer advances every iteration) and without a communicator, and replayed n_pairs times.  Same kernels, same order, same
// results as alpine_run(2 * n_pairs, 0); only the submission path differs.  Measured: no difference -- the stream is
// never starved by the host (the kernel trace shows 0.3 us of gaps per iteration), so the production loop stays eager.
extern "C" int alpine_debug_run_graph(alpine_ctx* c, int n_pairs)
{
    int rc = ready(c);
    if (rc) return rc;
    if (c->comm || c->use_als || c->transform_only || n_pairs < 1) return fail(c, ALPINE_ERR_UNSUPPORTED, "single-shard MU loop only");
    // whatever happens below, the ctx gets its profiling / loss switches back and the graph objects are released
    struct Restore {
        alpine_ctx* c; bool prof;
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        ~Restore() {
            if (exec) (void)hipGraphExecDestroy(exec);
            if (graph) (void)hipGraphDestroy(graph);
            c->loss_enabled = true; c->prof = prof; c->pending_loss = true;
        }
    } keep{c, c->prof};
    c->prof = false;
    c->loss_enabled = false;
    for (int it = 0; it < 2; ++it) if ((rc = alpine_iter(c, 1))) return rc;          // reach the steady state (fused tails valid)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    for (int it = 0; it < 2 && !rc; ++it) rc = alpine_iter(c, 1);
    const hipError_t e = hipStreamEndCapture(c->stream, &keep.graph);               // always ends the capture, also after a failed launch
    if (rc) return rc;
    HIPCHK(c, e);
    HIPCHK(c, hipGraphInstantiate(&keep.exec, keep.graph, nullptr, nullptr, 0));
    for (int p = 0; p < n_pairs; ++p) HIPCHK(c, hipGraphLaunch(keep.exec, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// Diagnostics (tools/xcd_bias_sweep.py): re-divide the two sweeps with another even/odd bias ON THE SAME ctx, i.e. on the same
// physical placement of X -- sweep times of two ctxs differ by +-2 % from placement alone, which hides an effect of this size.
extern "C" int alpine_debug_set_xcd_bias(alpine_ctx* c, int per_mille)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int old = c->xcd_bias_pm;
    const bool old_auto = c->xcd_bias_auto;
    c->xcd_bias_pm = std::max(-200, std::min(200, per_mille));
    c->xcd_bias_auto = false;                          // an explicit request: applied as given (make_geom)
    apply_sweep_geometry(c, c->sweep_bf);
    auto need = [&](const SweepGeom& g) { return (int64_t)g.nwg * g.maxp * g.bf * c->KP; };
    if (need(c->geomA) > c->piecesA_cap || need(c->geomB) > c->piecesB_cap) {
        c->xcd_bias_pm = old;
        c->xcd_bias_auto = old_auto;
        apply_sweep_geometry(c, c->sweep_bf);
        return fail(c, ALPINE_ERR_UNSUPPORTED, "the pieces buffers are too small for that division");
    }
    c->tail_valid = false;
    return 0;
}

// Diagnostics / tests: the result-preserving knobs of the library (same factors up to summation order, different kernels or launch
// structure), one explicit call each -- the production library takes none of them from the environment.
//   "no_tail" 0|1        separate phase1_open_kernel every iteration instead of the H update's fused tail
//   "fused_w" 0|1        W update and W^T W partial blocks in one launch (default 1)
//   "unfused_mid" 0|1    every small kernel and reduction in its own launch (implies no_tail, fused_w 0)
//   "guided_scalar" 0|1  the per-(covariate, class) scalar form of the H update's guided terms instead of the MFMA products
//   "tail_stats_per_covariate" 0|1
//   "sg_variant" 0|1|2   pipeline shape of the float32-MFMA sweep
//   "x3_variant" -1|0|2  matrix instruction of the x3 sweeps: 0 = 32x32x16, 2 = 16x16x32 general form, -1 = from the data; BEFORE alpine_finalize_X
//   "wide_one_pass" 0|1  128 < K <= 256 on the x3 sweeps: one pass over X per sweep (default up to K = 224) or one per component half; BEFORE alpine_finalize_X
//   "x3_narrow" 0|1      512-column workgroup tiles at K <= 64 (default: shards of <= 131 072 cells on the 32x32x16 form); BEFORE alpine_finalize_X
//   "x3_two_wave" -1|0|1 64 < K <= 128 on the x3 sweeps: the two-waves-per-SIMD kernel (default) or the one-wave forms; BEFORE alpine_finalize_X
extern "C" int alpine_debug_set_option(alpine_ctx* c, const char* name, int value)
{
    if (!c || !name) return fail(c, ALPINE_ERR_BAD_ARG, "NULL argument");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const std::string n(name);
    const bool before_finalize = n == "x3_variant" || n == "x3_narrow" || n == "wide_one_pass" || n == "x3_two_wave";
    if (before_finalize && c->x_final) return fail(c, ALPINE_ERR_STATE, "option %s must be set before alpine_finalize_X", name);
    if (n == "no_tail") c->no_tail = value != 0 || c->unfused_mid;
    else if (n == "fused_w") c->fused_w = value != 0 && !c->unfused_mid;
    else if (n == "unfused_mid") { c->unfused_mid = value != 0; if (c->unfused_mid) { c->no_tail = true; c->fused_w = false; } }
    else if (n == "guided_scalar") c->no_guided_mfma = value != 0;
    else if (n == "tail_stats_per_covariate") c->tail_stats_per_cov = value != 0;
    else if (n == "sg_variant") c->sg_variant = value;
    else if (n == "x3_variant") { if (value != -1 && value != 0 && value != 2) return fail(c, ALPINE_ERR_BAD_ARG, "x3_variant must be -1, 0 or 2"); c->x3_variant = value; }
    else if (n == "x3_two_wave") { if (value < -1 || value > 1) return fail(c, ALPINE_ERR_BAD_ARG, "x3_two_wave must be -1, 0 or 1"); c->x3_two_wave_opt = value; }
    else if (n == "wide_one_pass") {
        if (!c->wide || !c->x3 || c->x3_ablate) return 0;     // only wide models on the x3 sweeps have the two forms
        c->wide_one_pass = value != 0 && std::max(c->Gp, c->Np) <= ((int64_t)1 << 25);
        c->wide_two_wave = false;                            // (alpine_finalize_X decides again)
        apply_sweep_geometry(c, c->wide_one_pass ? 256 : 512);
    }
    else if (n == "x3_narrow") {
        if (!c->x3 || c->KT > 2) return 0;                 // only the x3 sweeps at K <= 64 have two tile widths: nothing to switch
        c->x3_narrow_pref = value != 0; c->x3_narrow_forced = true;
        c->x3_narrow = c->x3_narrow_pref;
        apply_sweep_geometry(c, c->x3_narrow ? 512 : 1024);
    }
    else return fail(c, ALPINE_ERR_BAD_ARG, "unknown option %s", name);
    c->tail_valid = false;
    return 0;
}

// Diagnostics / tests: re-divide the two sweeps with teams of `width` workgroups (SweepGeom::gw; 0 = the library's own choice, 1 = no
// teams).  Same results up to summation order; a width the grid cannot be dealt in (or a non-x3 ctx) falls back to 1.
extern "C" int alpine_debug_set_team_width(alpine_ctx* c, int width)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    if (width < 0 || width > 32) return fail(c, ALPINE_ERR_BAD_ARG, "team width %d outside [0, 32]", width);
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    const int old = c->team_force;
    c->team_force = width;
    apply_sweep_geometry(c, c->sweep_bf);
    auto need = [&](const SweepGeom& g) { return (int64_t)g.nwg * g.maxp * g.bf * c->KP; };
    if ((!c->transform_only && need(c->geomA) > c->piecesA_cap) || need(c->geomB) > c->piecesB_cap) {
        c->team_force = old;
        apply_sweep_geometry(c, c->sweep_bf);
        return fail(c, ALPINE_ERR_UNSUPPORTED, "the pieces buffers are too small for that division");
    }
    c->tail_valid = false;
    return 0;
}

extern "C" int alpine_set_profiling(alpine_ctx* c, int enabled)
{
    if (!c) return ALPINE_ERR_BAD_ARG;
    // enabled = n > 0: events around the sweeps / all-reduce of every n-th iteration (1 = every launch).  An event record
    // between two kernels costs the stream ~2 us: sampling keeps the measurement from slowing sub-millisecond iterations
    c->prof = enabled != 0;
    c->prof_every = enabled > 1 ? enabled : 1;
    c->prof_tick = 0;
    c->prof_now = c->prof;
    for (int k = 0; k < ALPINE_KERNEL_COUNT; ++k) c->ev_used[k] = 0;
    return 0;
}

extern "C" int alpine_get_kernel_time(alpine_ctx* c, int which, double* total_ms, int64_t* launches)
{
    if (!c || which < 0 || which >= ALPINE_KERNEL_COUNT) return fail(c, ALPINE_ERR_BAD_ARG, "bad kernel id");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double t = 0;
    for (size_t i = 0; i < c->ev_used[which]; ++i) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev[which][i].first, c->ev[which][i].second));
        t += ms;
    }
    if (total_ms) *total_ms = t;
    if (launches) *launches = (int64_t)c->ev_used[which];
    return 0;
}

extern "C" int alpine_read_buffer(alpine_ctx* c, int which, int64_t offset, int64_t n, float* host)
{
    if (!c || !host || offset < 0 || n < 0) return fail(c, ALPINE_ERR_BAD_ARG, "bad arguments");
    const float* base = nullptr; int64_t size = 0;
    switch (which) {
        case ALPINE_BUF_REDUCE_BLOCK: base = c->red; size = c->red_floats; break;
        case ALPINE_BUF_WTW: base = c->WtW; size = (int64_t)c->KP * c->KP; break;
        case ALPINE_BUF_W: base = c->W; size = c->Gp * c->KP; break;
        case ALPINE_BUF_H: base = c->H; size = c->Np * c->KP; break;
        case ALPINE_BUF_X_GENES_BY_CELLS: base = c->Xgn; size = c->bf16 ? 0 : c->Gp * c->Np; break;
        case ALPINE_BUF_X_CELLS_BY_GENES: base = c->Xng; size = (c->bf16 || c->transform_only) ? 0 : c->Np * c->Gp; break;
        default: return fail(c, ALPINE_ERR_BAD_ARG, "unknown buffer %d", which);
    }
    if (offset + n > size) return fail(c, ALPINE_ERR_BAD_ARG, "range outside buffer (%lld floats)", (long long)size);
    HIPCHK(c, hipSetDevice(c->device));
    HOSTCOPY(c, host, base + offset, sizeof(float) * n, hipMemcpyDeviceToHost);
    return 0;
}

#ifdef ALPINE_STAMPS
extern "C" int alpine_debug_read_sweep_stamps(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(alpine::g_sweep_stamps), sizeof(unsigned long long) * n);
}
extern "C" int alpine_debug_read_sweep_hist(unsigned int* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(alpine::g_sweep_hist), sizeof(unsigned int) * n);
}
extern "C" int alpine_debug_read_hu_stamps(unsigned long long* host, int n)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(alpine::g_hu_stamps), sizeof(unsigned long long) * n);
}
#endif
