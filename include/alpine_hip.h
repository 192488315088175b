/*
 * alpine_hip.h -- C ABI of libalpine_hip.so: the MI355X (gfx950) implementation of ALPINE's
 * full-batch multiplicative-update NMF fit loop.
 *
 * The reference (ylaboratory/ALPINE v0.2.0) has no FFI/plugin seam of its own: the path sits
 * behind a Python class whose only backend selector is `device` (alpine/main.py:58,70).  This
 * header therefore defines the seam a maintainer binds with ctypes (INTEGRATION.md shows the
 * stub): each entry point names the reference code it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.
 *  - every call returns 0 on success or a negative alpine_status; the text of the failure is
 *    retrievable with alpine_last_error(ctx) (or alpine_last_error(NULL) when alpine_create
 *    itself failed).  Nothing throws or aborts across the boundary.
 *  - host buffers are borrowed for the duration of the call only; the library owns all device
 *    memory except an optional caller-provided reduce block (see alpine_config).
 *  - one ctx = one GPU = one shard of the cell axis; a ctx is driven by one host thread.
 *  - all matrices are float32.  Host-side shapes follow the reference: W is G x K, H is K x N,
 *    B_i is C_i x k_i, Y_i is C_i x N (alpine/main.py:28-34, :446-449), row-major, with the K
 *    columns ordered [cov_1 | ... | cov_C | unguided] (main.py:79).
 */
#ifndef ALPINE_HIP_H
#define ALPINE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALPINE_HIP_ABI_VERSION 6

typedef struct alpine_ctx alpine_ctx;

typedef enum {
    ALPINE_OK = 0,
    ALPINE_ERR_BAD_ARG = -1,
    ALPINE_ERR_HIP = -2,
    ALPINE_ERR_OOM = -3,
    ALPINE_ERR_STATE = -4,     /* call made in the wrong order (e.g. run before X / factors set) */
    ALPINE_ERR_UNSUPPORTED = -5,
    ALPINE_ERR_RCCL = -6       /* a collective or communicator call failed (text in alpine_last_error) */
} alpine_status;

enum { ALPINE_LOSS_KL = 0, ALPINE_LOSS_FROBENIUS = 1 };      /* main.py:57, :371 */
/* alpine_config.flags: a ctx that will only run alpine_transform keeps a single copy of X (genes x cells) and no
 * resources of the XH^T sweep; alpine_iter_begin / alpine_run then fail with ALPINE_ERR_STATE. */
enum {
    ALPINE_FLAG_TRANSFORM_ONLY = 1,
    /* bf16 storage path: both resident copies of X and the MFMA operand copies of W / H are bf16 (k-packed for
     * v_mfma_f32_32x32x16_bf16), accumulation and all factor updates stay float32.  X chunks must start at a cell
     * index that is a multiple of 8.  Halves the HBM traffic of the two sweeps; tolerance vs the float32 path is
     * reported by the tests. */
    ALPINE_FLAG_X_BF16 = 2,
    /* use_als=True of the reference (main.py:55, :523-588): block-coordinate updates, one component group
     * (covariate blocks first, unguided last) at a time: W_j (orthogonality within the block only), then H_j, with
     * HH^T / W^TW refreshed between groups.  One XH^T sweep + (C+1) W^TX sweeps per iteration.  With a sharded cell
     * axis the group loop needs one more (K x K) exchange per group: alpine_als_begin / alpine_als_group_begin/end. */
    ALPINE_FLAG_USE_ALS = 4,
    /* exact-split storage: X is kept as one or two bf16 planes whose sum is EXACTLY the float32 input (integer counts
     * < 256 need one plane, 16 significant bits two), the MFMA operand copies of W / H as three bf16 planes that sum
     * exactly to the float32 masters; the sweeps run on the bf16 matrix pipe (3 or 5 MFMAs per fragment pair), products
     * are exact in float32 and accumulate in float32, so results agree with the float32 path to rounding -- at HBM-bound
     * instead of fp32-MFMA-bound speed.  alpine_finalize_X fails with ALPINE_ERR_UNSUPPORTED if some element of X is not
     * exactly representable by two bf16 planes (then use the float32 layout).  X chunks start at multiples of 8 cells. */
    ALPINE_FLAG_X_SPLIT = 8,
    /* float32 X in HBM (layout, ingest and memory exactly as without any storage flag), but the two sweeps form every
     * product x*p from the exact bf16 planes of x and p on the bf16 matrix pipe: the six plane products that carry
     * everything above 2^-24 |x p| are accumulated in float32 (kernels_x3.hpp).  No precondition on X; results agree
     * with the float32-MFMA sweeps to float32 rounding.  All K <= 128: a wave owns 256 columns x K <= 64 or 128 columns x
     * K <= 128 (256 accumulator registers either way).  Ignored together with ALPINE_FLAG_X_BF16 / ALPINE_FLAG_X_SPLIT. */
    ALPINE_FLAG_X3_PRODUCTS = 16
};
enum { ALPINE_X_CELLS_BY_GENES = 0, ALPINE_X_GENES_BY_CELLS = 1 };
/* longest run of contraction rows (cells in the XH^T sweep, genes in the W^TX sweep) that one float32 accumulator covers */
#define ALPINE_MAX_ACCUMULATION_ROWS 16384

/* Constructor arguments of ALPINE (main.py:47-61) plus the shard geometry. */
typedef struct {
    int32_t struct_size;            /* sizeof(alpine_config), for ABI checking */
    int32_t device_id;              /* HIP device ordinal */
    int64_t n_genes;                /* G */
    int64_t n_cells;                /* N held by THIS ctx (the local shard of the cell axis) */
    int32_t n_components;           /* unguided K_u            (main.py:49).  K = K_u + sum k_i <= 1024; K <= 128 is the fast path, 128 < K <= 1024 a
                                       blocked path of ceil(K / 128) column blocks (slower: above K = 224 X is read once per block and sweep; float32
                                       storage only, no bf16-plane flags), guided components within the first 128 columns (sum k_i <= 128) */
    int32_t n_covariates;           /* C = len(covariate_keys) (main.py:50) */
    const int32_t* cov_components;  /* k_i, length C           (main.py:50); 0 <= k_i <= 64 (0: the covariate only contributes its loss column) */
    const int32_t* cov_levels;      /* C_i = rows of Y_i, length C (encoder.py:31) */
    const double* lam;              /* lambda_i, length C      (main.py:51) */
    double orth_W;                  /* main.py:52 */
    double alpha_W;                 /* main.py:53 */
    double l1_ratio_W;              /* main.py:54 */
    double eps;                     /* main.py:59 */
    int32_t loss_type;              /* ALPINE_LOSS_*           (main.py:57) */
    int32_t split_a;                /* XH^T sweep: 0 (default) = the library's own work division: one equal share of the
                                       (tile, row) space per resident workgroup.  n > 0 (a test / diagnostic knob) = about n
                                       workgroups per tile instead.  Either way a share is cut into spans of at most
                                       ALPINE_MAX_ACCUMULATION_ROWS contraction rows, each accumulated in float32 on its own
                                       and summed with the others in float64, so the knob changes the launch structure and
                                       the summation order but NOT the accuracy class (tests/test_gpu_float64_arbiter.py). */
    int32_t split_b;                /* W^TX sweep: same */
    int32_t flags;                  /* ALPINE_FLAG_* */
    void* stream;                   /* hipStream_t to enqueue on, NULL = the library creates one */
    void* reduce_block;             /* optional device buffer of alpine_reduce_block_floats() floats that the
                                       caller owns (e.g. a torch tensor it will all-reduce); NULL = library allocates */
    int64_t batch_capacity;         /* > 0: allocate the mini-batch view for up to this many cells per alpine_batch_step
                                       (batch_size of ALPINE.fit, main.py:86, :112); 0 = full batch only */
} alpine_config;

typedef struct {
    int32_t abi_version;
    int32_t k_total;                /* K = sum(k_i) + K_u */
    int32_t k_padded;               /* KP: K rounded up to a multiple of 32 (one MFMA tile) */
    int32_t split_a, split_b;       /* partial pieces a workgroup may write (stream-K spans can cross tiles) */
    int32_t grid_a, grid_b;         /* workgroups of the two sweeps */
    int64_t genes_padded, cells_padded;   /* leading dimensions of the two resident copies of X */
    int64_t reduce_block_floats;
    int64_t device_bytes;           /* total device memory held by the ctx */
    double x_sqnorm;                /* ||X_local||_F^2 (valid after alpine_finalize_X) */
    double x_multi_plane_fraction;  /* fraction of the elements of X that are not exactly one bf16 plane (float32 storage; after alpine_finalize_X) */
    int32_t x3_wide;                /* 1: the x3 sweeps run on v_mfma_f32_16x16x32_bf16 (full-significand data), 0: on 32x32x16 (count-like data) */
    int32_t sweep_waves_per_simd;   /* x3 sweeps: 2 = stream_gemm_x3v_kernel (64 < K <= 128: 8 waves per workgroup, 256 registers each), 1 = the
                                       one-wave-per-SIMD forms, 0 = not an x3 ctx (was `reserved`, always 0, before round 4) */
    int32_t span_rows_a, span_rows_b;                      /* contraction rows one float32 accumulator chain covers (<= ALPINE_MAX_ACCUMULATION_ROWS) */
    int32_t spans_per_workgroup_a, spans_per_workgroup_b;  /* accumulator restarts + 1 of a sweep workgroup (1 at BASELINE config 3) */
    int32_t xcd_bias_per_mille;     /* spans of even sweep workgroups are this much longer (negative: shorter) than the mean, odd ones the opposite */
    int32_t xcc_of_workgroup0;      /* placement probe: XCC id workgroup 0 of a one-per-CU grid landed on (-1: not probed) */
    int32_t team_width_a, team_width_b;   /* x3 sweeps: workgroups of one XCD that walk the same contraction rows side by side and share the
                                             panel through that XCD's L2 (1 = none); grid_a / grid_b count workgroups, not teams */
} alpine_info;

/* Number of floats in the per-iteration reduce block for a configuration (so a caller can allocate
 * it before alpine_create).  Returns <0 on bad arguments. */
int64_t alpine_reduce_block_floats(const alpine_config* cfg);

/* Replaces: device selection + tensor construction of ALPINE._initialize_matrices (main.py:445-470). */
int alpine_create(const alpine_config* cfg, alpine_ctx** out);
int alpine_destroy(alpine_ctx* ctx);
const char* alpine_last_error(const alpine_ctx* ctx);
int alpine_get_info(alpine_ctx* ctx, alpine_info* info);

/* Replaces `torch.tensor(X_array, device=...)` (main.py:445).  X arrives in cell chunks
 * [cell0, cell0+n_cells) of the LOCAL shard, in either layout, from host or device memory
 * (ld = elements between consecutive rows of the given chunk).  Both resident copies of X
 * (genes x cells for the W^TX sweep, cells x genes for the XH^T sweep) are written. */
int alpine_upload_X_host(alpine_ctx* ctx, const float* host, int layout, int64_t ld, int64_t cell0, int64_t n_cells);
int alpine_upload_X_device(alpine_ctx* ctx, const float* dev, int layout, int64_t ld, int64_t cell0, int64_t n_cells);
/* After the last chunk: computes ||X_local||^2 (used by the trace-form loss, replaces nothing). */
int alpine_finalize_X(alpine_ctx* ctx);

/* Replaces `torch.tensor(y.T, device=...)` (main.py:446-449).  host: C_i x n_cells_local, row stride ld. */
int alpine_upload_Y(alpine_ctx* ctx, int cov, const float* host, int64_t ld);

/* Replaces the torch.rand draws' destination (main.py:454-470): the caller draws on the host in
 * the reference's order and hands the factors over.  H is K x N_local with row stride ldH (so a
 * shard can point into a K x N_total matrix).  B = array of C pointers, B[i] is C_i x k_i. */
int alpine_set_factors(alpine_ctx* ctx, const float* W, const float* H, int64_t ldH, const float* const* B);
/* Replaces AlpineMatrices.to_numpy (main.py:36-43) for W, H, B (X and Y stay with the caller). */
int alpine_get_factors(alpine_ctx* ctx, float* W, float* H, int64_t ldH, float* const* B);

/* One MU iteration (main.py:589-663 + :666) split at the only point where shards exchange data:
 *   alpine_iter_begin : sums over LOCAL cells of everything that depends on the old H
 *                       (XH^T, HH^T, B-update sums, prediction-loss sums) -> reduce block
 *   [ multi-GPU: the caller all-reduces (sum) the reduce block on the ctx stream ]
 *   alpine_iter_end   : finalises the previous iteration's loss row, then W update (main.py:596-605),
 *                       B updates (:615-628), W^TW, W^TX sweep and H update (:631-663).
 *                       update=0 only finalises the pending loss row (used once after the last iteration).
 * Asynchronous: work is enqueued on the ctx stream. */
int alpine_iter_begin(alpine_ctx* ctx);
int alpine_iter_end(alpine_ctx* ctx, int update);
int alpine_reduce_block(alpine_ctx* ctx, void** dev_ptr, int64_t* n_floats);

/* use_als=True (ALPINE_FLAG_USE_ALS) with a sharded cell axis: the group loop (main.py:525-588) needs H H^T of ALL cells
 * after every group, so the iteration has one more exchange per group:
 *   alpine_iter_begin -> [all-reduce the reduce block] -> alpine_als_begin (pending loss row, all B updates)
 *   for grp = 0 .. n_covariates:  alpine_als_group_begin(grp)  (grp > 0: local H H^T -> its slot of the reduce block)
 *                                 [grp > 0: all-reduce that K_padded^2 slot, see alpine_reduce_block_hht]
 *                                 alpine_als_group_end(grp)    (W_grp, W^TW, W^TX sweep, H_grp)
 * A single shard just calls alpine_iter_end(ctx, 1), which is this sequence without the exchanges. */
int alpine_als_begin(alpine_ctx* ctx);
int alpine_als_group_begin(alpine_ctx* ctx, int grp);
int alpine_als_group_end(alpine_ctx* ctx, int grp);
int alpine_reduce_block_hht(alpine_ctx* ctx, int64_t* offset_floats, int64_t* n_floats);

/* Multi-GPU inside the library (the reference is single-device, main.py:70): one process per GPU, one
 * ctx per process, the cell axis sharded over the ranks of an RCCL communicator.  Rank 0 calls alpine_comm_get_unique_id
 * and hands the 128 bytes to every rank by any means (a file, MPI, torch.distributed ...); every rank then calls
 * alpine_comm_init_rank on its ctx (collective: returns when all ranks have joined).  With a communicator attached the
 * COMPOSITE entry points -- alpine_iter, alpine_run, alpine_batch_step, alpine_epoch_loss -- enqueue the sum all-reduce of
 * the reduce block (ncclAllReduce over xGMI) on the ctx stream themselves, between their begin and end halves (the
 * block-coordinate branch adds the K x K slot after every group); alpine_batch_step then accepts n == 0.  The split
 * entry points (alpine_iter_begin / _end, ...) never communicate: they remain for callers who bring their own collective,
 * or who call alpine_comm_all_reduce (in place, on a range of the reduce block) in between.  W, B are replicated, every
 * rank ends with identical copies; H, X, Y stay local.  alpine_destroy also destroys the communicator. */
#define ALPINE_COMM_ID_BYTES 128
int alpine_comm_get_unique_id(void* id_out /* ALPINE_COMM_ID_BYTES bytes */);
int alpine_comm_init_rank(alpine_ctx* ctx, const void* id, int nranks, int rank);
/* ncclGetVersion of the librccl this library is linked against, as one integer (e.g. 22203); for run reports. */
int alpine_comm_version(int* version_out);
/* Rank count and this ctx's rank as the ATTACHED communicator reports them (ncclCommCount / ncclCommUserRank), for run reports that
 * must show RCCL saw N ranks; ALPINE_ERR_STATE without a communicator.  Either pointer may be NULL. */
int alpine_comm_count(alpine_ctx* ctx, int* nranks_out, int* rank_out);
/* ONE process, n ctxs on n DIFFERENT GPUs (the reference's fit is one blocking call in one process, main.py:82-147): ncclCommInitAll
 * over the ctxs' devices, ctxs[i] becomes rank i.  No unique id, no launcher.  Afterwards each ctx is driven by its own host thread
 * (a ctx stays single-threaded): the composite entry points called concurrently on the n ctxs meet inside the all-reduce. */
int alpine_comm_init_all(alpine_ctx* const* ctxs, int n);
int alpine_comm_destroy(alpine_ctx* ctx);
int alpine_comm_all_reduce(alpine_ctx* ctx, int64_t offset_floats, int64_t n_floats);
/* alpine_iter_begin + [all-reduce when a communicator is attached] + alpine_iter_end(update). */
int alpine_iter(alpine_ctx* ctx, int update);

/* Mini-batch fitting (main.py:509-521, :512-663; alpine/utils/sampling.py:58-71): the caller draws the epoch's index
 * stream exactly as the reference does (torch.randperm, or the weighted sampler with replacement) and feeds it one batch
 * at a time.  alpine_batch_step gathers the n cells idx[0..n) of the local shard into a contiguous view, runs the W, B
 * and H updates on that view and scatters the updated columns of H back (replaces X[:, idx] / H[:, idx] = ...).
 * alpine_epoch_loss appends the loss row of the current factors over ALL cells (main.py:666).  float32 storage only.
 * Both are split at the point where shards exchange data, like an iteration:
 *   alpine_batch_begin(idx, n)  gather + sums over the batch's LOCAL cells -> reduce block; n == 0 is allowed here (a
 *                               shard that holds none of the batch's cells contributes zeros)
 *   [ caller all-reduces the reduce block ]
 *   alpine_batch_end()          W and B updates (replicated), W^TX sweep and H update on the view, scatter
 *   alpine_epoch_loss_begin() / [all-reduce] / alpine_epoch_loss_end()
 * alpine_batch_step = begin + end, alpine_epoch_loss = begin + end (single shard). */
int alpine_batch_step(alpine_ctx* ctx, const int64_t* idx, int64_t n);
int alpine_batch_begin(alpine_ctx* ctx, const int64_t* idx, int64_t n);
int alpine_batch_end(alpine_ctx* ctx);
int alpine_epoch_loss(alpine_ctx* ctx);
int alpine_epoch_loss_begin(alpine_ctx* ctx);
int alpine_epoch_loss_end(alpine_ctx* ctx);

/* Replaces the loop of ALPINE._fit for one device (main.py:500-667): n_iters iterations; with_loss!=0
 * also produces one loss row per iteration ([total, recon, pred_1..pred_C], main.py:726-753). */
int alpine_run(alpine_ctx* ctx, int n_iters, int with_loss);
/* Loss rows accumulated so far (float64, n_rows x (C+2)); synchronises the stream. */
int alpine_get_losses(alpine_ctx* ctx, double* rows, int64_t max_rows, int64_t* n_rows);
int alpine_reset_losses(alpine_ctx* ctx);

/* Replaces the loop of ALPINE._transform (alpine/main.py:705-709): n_iter times H *= 2W^TX / max(2W^T(WH), eps) with W
 * frozen and no covariate terms, from the W and H given to alpine_set_factors (B is ignored; pass NULL with
 * n_covariates = 0).  One W^TX sweep, then all iterations of a cell tile in registers.  Asynchronous. */
int alpine_transform(alpine_ctx* ctx, int n_iter);

/* Replaces ALPINE._scale_matrices (main.py:772-781). */
int alpine_scale(alpine_ctx* ctx);

int alpine_synchronize(alpine_ctx* ctx);

/* Direct-form ||X_local - W H_local||_F^2 of the CURRENT factors with float64 accumulation: the "common evaluator" of
 * SURVEY.md section 7, and the G x N part of ALPINE.compute_loss (main.py:216-219: np.linalg.norm(X - W @ H)**2).
 * Needs float32 storage of X (any ctx but the bf16 / split ones; a transform-only ctx works).  Synchronises. */
int alpine_eval_recon_direct(alpine_ctx* ctx, double* out);

/* Measurement: when enabled, hipEvents bracket every launch of the two streaming sweeps and every all-reduce the
 * library enqueues (the latter: transfer + waiting for the slowest rank). */
enum { ALPINE_KERNEL_SWEEP_XHT = 0, ALPINE_KERNEL_SWEEP_WTX = 1, ALPINE_KERNEL_ALLREDUCE = 2, ALPINE_KERNEL_COUNT = 3 };
/* enabled: 0 = off, 1 = every launch, n > 1 = the launches of every n-th iteration only (an event record between two kernels
 * costs the stream about 2 us, which matters for sub-millisecond iterations). */
int alpine_set_profiling(alpine_ctx* ctx, int enabled);
/* Diagnostics: span length of even sweep workgroups +per_mille, odd -per_mille (same results up to summation order). */
int alpine_debug_set_xcd_bias(alpine_ctx* ctx, int per_mille);
/* Diagnostics / tests: the result-preserving knobs of the library (other kernels or launch structure, same factors up to summation
 * order) as explicit calls; the production library reads NONE of them from the environment (it reads ALPINE_HIP_GUARD, ALPINE_HIP_LDS_LIMIT
 * and ALPINE_HIP_XCD_BIAS only).  Names: "no_tail", "fused_w", "unfused_mid", "guided_scalar", "tail_stats_per_covariate", "sg_variant",
 * and -- before alpine_finalize_X -- "x3_variant" (-1 | 0 | 2), "x3_narrow" (0 | 1), "wide_one_pass" (0 | 1) and "x3_two_wave" (-1 | 0 | 1:
 * 64 < K <= 128, the two-waves-per-SIMD sweep; -1 = for data with more than one bf16 plane, alpine_info.sweep_waves_per_simd reports it). */
int alpine_debug_set_option(alpine_ctx* ctx, const char* name, int value);
/* Diagnostics / tests: teams of `width` sweep workgroups (alpine_info.team_width_*): 0 = the library's own choice, 1 = none.  Same
 * results up to summation order. */
int alpine_debug_set_team_width(alpine_ctx* ctx, int width);
/* Diagnostics: 2 * n_pairs steady-state MU iterations (no loss rows, no communicator) replayed from a hipGraph of two. */
int alpine_debug_run_graph(alpine_ctx* ctx, int n_pairs);
int alpine_get_kernel_time(alpine_ctx* ctx, int which, double* total_ms, int64_t* launches);

/* Debug/test access to device-resident intermediates (synchronises): copies n floats starting at
 * element `offset` of the named buffer to host. */
enum {
    ALPINE_BUF_REDUCE_BLOCK = 0,   /* [XH^T: genes_padded x KP | HH^T: KP x KP | covariate sums | ||X||^2 hi,lo] */
    ALPINE_BUF_WTW = 1,            /* KP x KP */
    ALPINE_BUF_W = 2,              /* genes_padded x KP, gene-major */
    ALPINE_BUF_H = 3,              /* cells_padded x KP, cell-major */
    ALPINE_BUF_X_GENES_BY_CELLS = 4,
    ALPINE_BUF_X_CELLS_BY_GENES = 5
};
int alpine_read_buffer(alpine_ctx* ctx, int which, int64_t offset, int64_t n, float* host);

#ifdef __cplusplus
}
#endif
#endif /* ALPINE_HIP_H */
