/* A host program in plain C that drives libalpine_hip.so through include/alpine_hip.h only: what a binding in any
 * language does.  It replaces the device part of ALPINE._fit (alpine/main.py:436-472 upload, :500-667 loop, :772-781
 * scaling) for a problem read from a flat binary file and writes factors + loss rows to another.
 *
 *   gcc -O2 -Iinclude examples/fit_c.c -o examples/fit_c -Lalpine_amd -lalpine_hip -Wl,-rpath,$PWD/alpine_amd \
 *       -Wl,-rpath-link,/opt/rocm/lib
 *   examples/fit_c problem.bin result.bin
 *   examples/fit_c --ranks R problem.bin result.bin     cell axis sharded over R GPUs (devices 0..R-1), one process each
 *   examples/fit_c --devices D problem.bin result.bin   the same over D GPUs of THIS process: one ctx and one host thread per GPU
 *
 * --devices D: no fork, no id exchange -- D ctxs, alpine_comm_init_all (ncclCommInitAll) makes ctx r rank r, then D threads each run the
 * very call sequence of the one-GPU program on their ctx; the ranks meet inside alpine_run's all-reduce.  (The reference's fit is one
 * blocking call in one process, alpine/main.py:82-147: this is the form a drop-in binding uses.)
 *
 * --ranks R: the parent forks R workers BEFORE anything touches a GPU.  Worker r takes the cells [N r/R, N (r+1)/R), rank 0
 * draws the RCCL id (alpine_comm_get_unique_id) and publishes it through a file next to the result, every worker joins
 * with alpine_comm_init_rank, and alpine_run then enqueues the one all-reduce per iteration itself -- the host loop is the
 * same three calls as on one GPU.  Workers write result.bin.rank<r>; the parent splices the columns of H together.
 *
 * problem.bin (little endian): int32 {magic 0x414c5031, G, N, Ku, C, loss_type, flags, T, scale}, then per covariate
 * int32 {k_i, C_i}; float64 {orth_W, alpha_W, l1_ratio_W, eps}; float64 lam[C]; float32 X[N][G] (cells x genes);
 * per covariate float32 Y_i[C_i][N]; float32 W0[G][K]; float32 H0[K][N]; per covariate float32 B0_i[C_i][k_i].
 * result.bin: int32 n_loss_rows; float64 losses[n][C+2]; float32 W[G][K]; float32 H[K][N]; per covariate B_i.
 * tests/test_c_abi_example.py writes the problem from a golden case and checks the result against the reference. */
#ifndef _DEFAULT_SOURCE
#define _DEFAULT_SOURCE
#endif
#include <pthread.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include "alpine_hip.h"

#define MAXC 16

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

static void die(const char* what, alpine_ctx* ctx)
{
    fprintf(stderr, "fit_c: %s: %s\n", what, alpine_last_error(ctx));
    exit(1);
}
static void rd(void* p, size_t sz, size_t n, FILE* f) { if (fread(p, sz, n, f) != n) { fprintf(stderr, "fit_c: short read\n"); exit(2); } }
static float* rdf(size_t n, FILE* f) { float* p = (float*)malloc(sizeof(float) * (n ? n : 1)); if (!p) exit(3); rd(p, sizeof(float), n, f); return p; }

/* rank 0 writes the id to <base>.id.tmp and renames it (atomic); the others wait for <base>.id */
static void exchange_id(const char* base, int rank, unsigned char* id)
{
    char path[4096], tmp[4096];
    snprintf(path, sizeof path, "%s.id", base);
    if (rank == 0) {
        if (alpine_comm_get_unique_id(id)) die("alpine_comm_get_unique_id", NULL);
        snprintf(tmp, sizeof tmp, "%s.id.tmp", base);
        FILE* f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, ALPINE_COMM_ID_BYTES, f) != ALPINE_COMM_ID_BYTES) { perror(tmp); exit(2); }
        fclose(f);
        if (rename(tmp, path)) { perror(path); exit(2); }
        return;
    }
    for (int tries = 0; tries < 6000; ++tries) {            /* up to 60 s */
        FILE* f = fopen(path, "rb");
        if (f) {
            const size_t n = fread(id, 1, ALPINE_COMM_ID_BYTES, f);
            fclose(f);
            if (n == ALPINE_COMM_ID_BYTES) return;
        }
        usleep(10000);
    }
    fprintf(stderr, "fit_c: rank %d never saw the communicator id\n", rank);
    exit(2);
}

/* one shard: cells [c0, c1) of the problem on device `device`; nranks > 0 attaches a communicator */
static int run_shard(const char* problem, const char* result, int rank, int nranks, int device)
{
    FILE* f = fopen(problem, "rb");
    if (!f) { perror(problem); return 2; }
    int32_t hdr[9];
    rd(hdr, sizeof(int32_t), 9, f);
    if (hdr[0] != 0x414c5031) { fprintf(stderr, "fit_c: bad magic\n"); return 2; }
    const int64_t G = hdr[1], N = hdr[2];
    const int Ku = hdr[3], C = hdr[4], T = hdr[7], scale = hdr[8];
    if (C > MAXC) { fprintf(stderr, "fit_c: too many covariates\n"); return 2; }
    int32_t k[MAXC], lev[MAXC];
    int K = Ku;
    for (int i = 0; i < C; ++i) { int32_t kc[2]; rd(kc, sizeof(int32_t), 2, f); k[i] = kc[0]; lev[i] = kc[1]; K += kc[0]; }
    double reg[4], lam[MAXC];
    rd(reg, sizeof(double), 4, f);
    rd(lam, sizeof(double), (size_t)C, f);
    float* X = rdf((size_t)(N * G), f);
    float* Y[MAXC]; float* B[MAXC];
    for (int i = 0; i < C; ++i) Y[i] = rdf((size_t)(lev[i] * N), f);
    float* W = rdf((size_t)(G * K), f);
    float* H = rdf((size_t)(K * N), f);
    for (int i = 0; i < C; ++i) B[i] = rdf((size_t)(lev[i] * k[i]), f);
    fclose(f);
    const int R = nranks > 0 ? nranks : 1;
    const int64_t c0 = N * rank / R, c1 = N * (rank + 1) / R, n = c1 - c0;

    alpine_config cfg = {0};
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device_id = device;
    cfg.n_genes = G; cfg.n_cells = n;
    cfg.n_components = Ku; cfg.n_covariates = C;
    cfg.cov_components = k; cfg.cov_levels = lev; cfg.lam = lam;
    cfg.orth_W = reg[0]; cfg.alpha_W = reg[1]; cfg.l1_ratio_W = reg[2]; cfg.eps = reg[3];
    cfg.loss_type = hdr[5];
    cfg.flags = hdr[6];

    alpine_ctx* ctx = NULL;
    if (alpine_create(&cfg, &ctx)) die("alpine_create", ctx);
    if (nranks > 0) {
        unsigned char id[ALPINE_COMM_ID_BYTES];
        exchange_id(result, rank, id);
        if (alpine_comm_init_rank(ctx, id, nranks, rank)) die("alpine_comm_init_rank", ctx);
    }
    /* the shard's rows of X, columns of Y and H: pointers into the full arrays with the full leading dimensions */
    if (alpine_upload_X_host(ctx, X + c0 * G, ALPINE_X_CELLS_BY_GENES, G, 0, n)) die("alpine_upload_X_host", ctx);
    if (alpine_finalize_X(ctx)) die("alpine_finalize_X", ctx);
    for (int i = 0; i < C; ++i) if (alpine_upload_Y(ctx, i, Y[i] + c0, N)) die("alpine_upload_Y", ctx);
    if (alpine_set_factors(ctx, W, H + c0, N, (const float* const*)B)) die("alpine_set_factors", ctx);
    if (alpine_run(ctx, T, 1)) die("alpine_run", ctx);          /* with a communicator: + one ncclAllReduce per iteration */
    if (scale && alpine_scale(ctx)) die("alpine_scale", ctx);
    if (alpine_get_factors(ctx, W, H + c0, N, B)) die("alpine_get_factors", ctx);
    double* rows = (double*)malloc(sizeof(double) * (size_t)(T > 0 ? T : 1) * (size_t)(C + 2));
    int64_t n_rows = 0;
    if (alpine_get_losses(ctx, rows, T, &n_rows)) die("alpine_get_losses", ctx);
    alpine_info info;
    if (alpine_get_info(ctx, &info)) die("alpine_get_info", ctx);
    alpine_destroy(ctx);

    char path[4096];
    if (nranks > 0) snprintf(path, sizeof path, "%s.rank%d", result, rank); else snprintf(path, sizeof path, "%s", result);
    FILE* o = fopen(path, "wb");
    if (!o) { perror(path); return 2; }
    int32_t n32 = (int32_t)n_rows;
    fwrite(&n32, sizeof(int32_t), 1, o);
    fwrite(rows, sizeof(double), (size_t)n_rows * (size_t)(C + 2), o);
    fwrite(W, sizeof(float), (size_t)(G * K), o);
    fwrite(H, sizeof(float), (size_t)(K * N), o);              /* K x N with this shard's columns filled in */
    for (int i = 0; i < C; ++i) fwrite(B[i], sizeof(float), (size_t)(lev[i] * k[i]), o);
    fclose(o);
    printf("fit_c[%d/%d]: G=%lld cells [%lld, %lld) of %lld, K=%d (padded %d), %lld loss rows, last total loss %.9g, %.1f MiB on device %d\n",
           rank, R, (long long)G, (long long)c0, (long long)c1, (long long)N, K, info.k_padded, (long long)n_rows,
           n_rows ? rows[(n_rows - 1) * (C + 2)] : 0.0, (double)info.device_bytes / 1048576.0, device);
    return 0;
}

/* result.bin = rank 0's file with the other ranks' columns of H spliced in (W, B, losses are replicated) */
static int splice(const char* problem, const char* result, int R)
{
    FILE* f = fopen(problem, "rb");
    if (!f) { perror(problem); return 2; }
    int32_t hdr[9];
    rd(hdr, sizeof(int32_t), 9, f);
    const int64_t G = hdr[1], N = hdr[2];
    const int C = hdr[4];
    int K = hdr[3];
    size_t nb = 0;
    for (int i = 0; i < C; ++i) { int32_t kc[2]; rd(kc, sizeof(int32_t), 2, f); K += kc[0]; nb += (size_t)kc[0] * (size_t)kc[1]; }
    fclose(f);
    unsigned char* out = NULL; size_t out_sz = 0, h_off = 0;
    for (int r = 0; r < R; ++r) {
        char path[4096];
        snprintf(path, sizeof path, "%s.rank%d", result, r);
        FILE* g = fopen(path, "rb");
        if (!g) { perror(path); return 2; }
        int32_t n_rows;
        rd(&n_rows, sizeof n_rows, 1, g);
        const size_t sz = 4 + 8 * (size_t)n_rows * (size_t)(C + 2) + 4 * ((size_t)(G * K) + (size_t)K * (size_t)N + nb);
        unsigned char* buf = (unsigned char*)malloc(sz);
        if (!buf) return 3;
        memcpy(buf, &n_rows, 4);
        rd(buf + 4, 1, sz - 4, g);
        fclose(g);
        remove(path);
        if (r == 0) { out = buf; out_sz = sz; h_off = 4 + 8 * (size_t)n_rows * (size_t)(C + 2) + 4 * (size_t)(G * K); continue; }
        const int64_t c0 = N * r / R, c1 = N * (r + 1) / R;
        for (int kk = 0; kk < K; ++kk)
            memcpy(out + h_off + 4 * ((size_t)kk * (size_t)N + (size_t)c0), buf + h_off + 4 * ((size_t)kk * (size_t)N + (size_t)c0), 4 * (size_t)(c1 - c0));
        free(buf);
    }
    FILE* o = fopen(result, "wb");
    if (!o || fwrite(out, 1, out_sz, o) != out_sz) { perror(result); return 2; }
    fclose(o);
    char idp[4096];
    snprintf(idp, sizeof idp, "%s.id", result);
    remove(idp);
    return 0;
}

/* ---- --devices D: one process, D ctxs, D host threads ---- */
typedef struct {
    alpine_ctx* ctx;
    int rank, D, C, T, scale;
    int64_t G, N, c0, c1;
    const float* X; float* const* Y; float* W; float* H; float* const* B;
    double* rows; int64_t n_rows;
    int failed; char err[512];
} dev_job;

static void* dev_thread(void* arg)
{
    dev_job* j = (dev_job*)arg;
    alpine_ctx* ctx = j->ctx;
    const int64_t n = j->c1 - j->c0;
#define STEP(call) do { if (call) { j->failed = 1; snprintf(j->err, sizeof j->err, "%s: %s", #call, alpine_last_error(ctx)); return NULL; } } while (0)
    STEP(alpine_upload_X_host(ctx, j->X + j->c0 * j->G, ALPINE_X_CELLS_BY_GENES, j->G, 0, n));
    STEP(alpine_finalize_X(ctx));
    for (int i = 0; i < j->C; ++i) STEP(alpine_upload_Y(ctx, i, j->Y[i] + j->c0, j->N));
    STEP(alpine_set_factors(ctx, j->W, j->H + j->c0, j->N, (const float* const*)j->B));
    STEP(alpine_run(ctx, j->T, 1));                              /* every thread: its kernels + the one all-reduce per iteration */
    if (j->scale) STEP(alpine_scale(ctx));
    STEP(alpine_synchronize(ctx));
    /* W, B and the loss rows are replicated: rank 0 writes them; every rank writes its own columns of H (disjoint) */
    STEP(alpine_get_factors(ctx, j->rank == 0 ? j->W : NULL, j->H + j->c0, j->N, j->rank == 0 ? j->B : NULL));
    if (j->rank == 0) STEP(alpine_get_losses(ctx, j->rows, j->T, &j->n_rows));
#undef STEP
    return NULL;
}

static int run_devices(const char* problem, const char* result, int D)
{
    FILE* f = fopen(problem, "rb");
    if (!f) { perror(problem); return 2; }
    int32_t hdr[9];
    rd(hdr, sizeof(int32_t), 9, f);
    if (hdr[0] != 0x414c5031) { fprintf(stderr, "fit_c: bad magic\n"); return 2; }
    const int64_t G = hdr[1], N = hdr[2];
    const int Ku = hdr[3], C = hdr[4], T = hdr[7], scale = hdr[8];
    if (C > MAXC) { fprintf(stderr, "fit_c: too many covariates\n"); return 2; }
    int32_t k[MAXC], lev[MAXC];
    int K = Ku;
    for (int i = 0; i < C; ++i) { int32_t kc[2]; rd(kc, sizeof(int32_t), 2, f); k[i] = kc[0]; lev[i] = kc[1]; K += kc[0]; }
    double reg[4], lam[MAXC];
    rd(reg, sizeof(double), 4, f);
    rd(lam, sizeof(double), (size_t)C, f);
    float* X = rdf((size_t)(N * G), f);
    float* Y[MAXC]; float* B[MAXC];
    for (int i = 0; i < C; ++i) Y[i] = rdf((size_t)(lev[i] * N), f);
    float* W = rdf((size_t)(G * K), f);
    float* H = rdf((size_t)(K * N), f);
    for (int i = 0; i < C; ++i) B[i] = rdf((size_t)(lev[i] * k[i]), f);
    fclose(f);
    /* ctx r on device r % FIT_C_DEVICE_COUNT (RCCL wants one device per rank; fewer is for the tests' stand-in communicator) */
    const char* dc = getenv("FIT_C_DEVICE_COUNT");
    const int ND = dc && atoi(dc) >= 1 ? atoi(dc) : D;
    alpine_ctx* ctxs[64] = {0};
    dev_job jobs[64];
    memset(jobs, 0, sizeof jobs);
    for (int r = 0; r < D; ++r) {
        alpine_config cfg = {0};
        cfg.struct_size = (int32_t)sizeof(cfg);
        cfg.device_id = r % ND;
        cfg.n_genes = G; cfg.n_cells = N * (r + 1) / D - N * r / D;
        cfg.n_components = Ku; cfg.n_covariates = C;
        cfg.cov_components = k; cfg.cov_levels = lev; cfg.lam = lam;
        cfg.orth_W = reg[0]; cfg.alpha_W = reg[1]; cfg.l1_ratio_W = reg[2]; cfg.eps = reg[3];
        cfg.loss_type = hdr[5];
        cfg.flags = hdr[6];
        if (alpine_create(&cfg, &ctxs[r])) die("alpine_create", ctxs[r]);
    }
    if (alpine_comm_init_all(ctxs, D)) die("alpine_comm_init_all", ctxs[0]);
    for (int r = 0; r < D; ++r) {
        int nr = -1, me = -1;
        if (alpine_comm_count(ctxs[r], &nr, &me) || nr != D || me != r) { fprintf(stderr, "fit_c: communicator reports %d ranks / rank %d for ctx %d of %d\n", nr, me, r, D); return 1; }
    }
    double* rows = (double*)malloc(sizeof(double) * (size_t)(T > 0 ? T : 1) * (size_t)(C + 2));
    pthread_t th[64];
    for (int r = 0; r < D; ++r) {
        dev_job* j = &jobs[r];
        j->ctx = ctxs[r]; j->rank = r; j->D = D; j->C = C; j->T = T; j->scale = scale;
        j->G = G; j->N = N; j->c0 = N * r / D; j->c1 = N * (r + 1) / D;
        j->X = X; j->Y = Y; j->W = W; j->H = H; j->B = B; j->rows = rows;
        if (pthread_create(&th[r], NULL, dev_thread, j)) { perror("pthread_create"); return 2; }
    }
    int bad = 0;
    for (int r = 0; r < D; ++r) {
        pthread_join(th[r], NULL);
        if (jobs[r].failed) { fprintf(stderr, "fit_c: device thread %d failed: %s\n", r, jobs[r].err); bad = 1; }
    }
    alpine_info info;
    if (!bad && alpine_get_info(ctxs[0], &info)) die("alpine_get_info", ctxs[0]);
    for (int r = 0; r < D; ++r) alpine_destroy(ctxs[r]);
    if (bad) return 1;
    FILE* o = fopen(result, "wb");
    if (!o) { perror(result); return 2; }
    int32_t n32 = (int32_t)jobs[0].n_rows;
    fwrite(&n32, sizeof(int32_t), 1, o);
    fwrite(rows, sizeof(double), (size_t)jobs[0].n_rows * (size_t)(C + 2), o);
    fwrite(W, sizeof(float), (size_t)(G * K), o);
    fwrite(H, sizeof(float), (size_t)(K * N), o);
    for (int i = 0; i < C; ++i) fwrite(B[i], sizeof(float), (size_t)(lev[i] * k[i]), o);
    fclose(o);
    printf("fit_c[--devices %d]: G=%lld N=%lld K=%d (padded %d), %lld loss rows, last total loss %.9g\n", D, (long long)G, (long long)N, K, info.k_padded,
           (long long)jobs[0].n_rows, jobs[0].n_rows ? rows[(jobs[0].n_rows - 1) * (C + 2)] : 0.0);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc == 3) return run_shard(argv[1], argv[2], 0, 0, 0);
    if (argc == 5 && strcmp(argv[1], "--devices") == 0 && atoi(argv[2]) >= 1 && atoi(argv[2]) <= 64)
        return run_devices(argv[3], argv[4], atoi(argv[2]));
    if (argc != 5 || strcmp(argv[1], "--ranks") != 0 || atoi(argv[2]) < 1 || atoi(argv[2]) > 64) {
        fprintf(stderr, "usage: fit_c [--ranks R | --devices D] problem.bin result.bin\n");
        return 2;
    }
    const int R = atoi(argv[2]);
    setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);          /* RCCL's cross-process handles need dmabuf IPC; read when the HIP runtime starts */
    char idp[4096];
    snprintf(idp, sizeof idp, "%s.id", argv[4]);
    remove(idp);                                           /* a stale id of an earlier run must not be picked up */
    /* rank r runs on device r % D; D = R unless FIT_C_DEVICE_COUNT says otherwise (RCCL itself wants one device per rank:
     * fewer devices than ranks is for tests that preload a stand-in communicator on a one-GPU machine) */
    const char* dc = getenv("FIT_C_DEVICE_COUNT");
    const int D = dc && atoi(dc) >= 1 ? atoi(dc) : R;
    pid_t pids[64];
    for (int r = 0; r < R; ++r) {                          /* fork BEFORE any GPU call: the parent never touches the device */
        pids[r] = fork();
        if (pids[r] < 0) { perror("fork"); return 2; }
        if (pids[r] == 0) _exit(run_shard(argv[3], argv[4], r, R, r % D));
    }
    /* Reap the workers in the order they END.  A rank that dies before ncclCommInitRank would leave its peers blocked in the
     * communicator set-up for ever: as soon as one worker fails (or the deadline FIT_C_TIMEOUT_S, default 1800 s, passes) the
     * remaining ones -- only the pids forked above -- are killed and the run fails. */
    const char* to = getenv("FIT_C_TIMEOUT_S");
    const double deadline = now_s() + (to && atof(to) > 0 ? atof(to) : 1800.0);
    int bad = 0, left = R;
    while (left > 0) {
        int st = 0;
        const pid_t p = waitpid(-1, &st, WNOHANG);
        if (p == 0) {
            if (now_s() > deadline) { fprintf(stderr, "fit_c: workers still running at the deadline\n"); bad = 1; break; }
            struct timespec ts = {0, 20 * 1000 * 1000};
            nanosleep(&ts, NULL);
            continue;
        }
        if (p < 0) { perror("waitpid"); bad = 1; break; }
        for (int r = 0; r < R; ++r) if (pids[r] == p) {
            pids[r] = -1; --left;
            if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) { fprintf(stderr, "fit_c: rank %d failed\n", r); bad = 1; }
        }
        if (bad) break;
    }
    if (bad) {
        for (int r = 0; r < R; ++r) if (pids[r] > 0) kill(pids[r], SIGKILL);
        for (int r = 0; r < R; ++r) if (pids[r] > 0) waitpid(pids[r], NULL, 0);
        return 1;
    }
    return splice(argv[3], argv[4], R);
}
