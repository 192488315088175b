/* A host program in plain C that drives libalpine_hip.so through include/alpine_hip.h only: what a binding in any
 * language does.  It replaces the device part of ALPINE._fit (alpine/main.py:436-472 upload, :500-667 loop, :772-781
 * scaling) for a problem read from a flat binary file and writes factors + loss rows to another.
 *
 *   gcc -O2 -Iinclude examples/fit_c.c -o examples/fit_c -Lalpine_amd -lalpine_hip -Wl,-rpath,$PWD/alpine_amd \
 *       -Wl,-rpath-link,/opt/rocm/lib
 *   examples/fit_c problem.bin result.bin
 *
 * problem.bin (little endian): int32 {magic 0x414c5031, G, N, Ku, C, loss_type, flags, T, scale}, then per covariate
 * int32 {k_i, C_i}; float64 {orth_W, alpha_W, l1_ratio_W, eps}; float64 lam[C]; float32 X[N][G] (cells x genes);
 * per covariate float32 Y_i[C_i][N]; float32 W0[G][K]; float32 H0[K][N]; per covariate float32 B0_i[C_i][k_i].
 * result.bin: int32 n_loss_rows; float64 losses[n][C+2]; float32 W[G][K]; float32 H[K][N]; per covariate B_i.
 * tests/test_c_abi_example.py writes the problem from a golden case and checks the result against the reference. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "alpine_hip.h"

#define MAXC 16

static void die(const char* what, alpine_ctx* ctx)
{
    fprintf(stderr, "fit_c: %s: %s\n", what, ctx ? alpine_last_error(ctx) : "(no ctx)");
    exit(1);
}
static void rd(void* p, size_t sz, size_t n, FILE* f) { if (fread(p, sz, n, f) != n) { fprintf(stderr, "fit_c: short read\n"); exit(2); } }
static float* rdf(size_t n, FILE* f) { float* p = (float*)malloc(sizeof(float) * (n ? n : 1)); if (!p) exit(3); rd(p, sizeof(float), n, f); return p; }

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: fit_c problem.bin result.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
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

    alpine_config cfg = {0};
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.device_id = 0;
    cfg.n_genes = G; cfg.n_cells = N;
    cfg.n_components = Ku; cfg.n_covariates = C;
    cfg.cov_components = k; cfg.cov_levels = lev; cfg.lam = lam;
    cfg.orth_W = reg[0]; cfg.alpha_W = reg[1]; cfg.l1_ratio_W = reg[2]; cfg.eps = reg[3];
    cfg.loss_type = hdr[5];
    cfg.flags = hdr[6];

    alpine_ctx* ctx = NULL;
    if (alpine_create(&cfg, &ctx)) die("alpine_create", ctx);
    if (alpine_upload_X_host(ctx, X, ALPINE_X_CELLS_BY_GENES, G, 0, N)) die("alpine_upload_X_host", ctx);
    if (alpine_finalize_X(ctx)) die("alpine_finalize_X", ctx);
    for (int i = 0; i < C; ++i) if (alpine_upload_Y(ctx, i, Y[i], N)) die("alpine_upload_Y", ctx);
    if (alpine_set_factors(ctx, W, H, N, (const float* const*)B)) die("alpine_set_factors", ctx);
    if (alpine_run(ctx, T, 1)) die("alpine_run", ctx);
    if (scale && alpine_scale(ctx)) die("alpine_scale", ctx);
    if (alpine_get_factors(ctx, W, H, N, B)) die("alpine_get_factors", ctx);
    double* rows = (double*)malloc(sizeof(double) * (size_t)(T > 0 ? T : 1) * (size_t)(C + 2));
    int64_t n_rows = 0;
    if (alpine_get_losses(ctx, rows, T, &n_rows)) die("alpine_get_losses", ctx);
    alpine_info info;
    if (alpine_get_info(ctx, &info)) die("alpine_get_info", ctx);
    alpine_destroy(ctx);

    FILE* o = fopen(argv[2], "wb");
    if (!o) { perror(argv[2]); return 2; }
    int32_t n32 = (int32_t)n_rows;
    fwrite(&n32, sizeof(int32_t), 1, o);
    fwrite(rows, sizeof(double), (size_t)n_rows * (size_t)(C + 2), o);
    fwrite(W, sizeof(float), (size_t)(G * K), o);
    fwrite(H, sizeof(float), (size_t)(K * N), o);
    for (int i = 0; i < C; ++i) fwrite(B[i], sizeof(float), (size_t)(lev[i] * k[i]), o);
    fclose(o);
    printf("fit_c: G=%lld N=%lld K=%d (padded %d), %lld loss rows, last total loss %.9g, %.1f MiB on the device\n",
           (long long)G, (long long)N, K, info.k_padded, (long long)n_rows, n_rows ? rows[(n_rows - 1) * (C + 2)] : 0.0,
           (double)info.device_bytes / 1048576.0);
    return 0;
}
