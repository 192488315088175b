// Standalone harness around the library's x3 sweep kernels (diagnostics; not part of the library): the forms of stream_gemm_x3_kernel /
// stream_gemm_x3w_kernel (general, one-plane, no zero-plane test) with and without teams (SweepGeom::gw), on count-like and on
// full-significand X, both sweep orientations, at BASELINE config 4's per-GPU share (K = 105) and config 3 (K = 60).  Every variant's
// pieces are reduced with the library's own reduce_pieces_kernel and compared with the first variant's; variants are timed interleaved.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o tools/x3w_bench tools/x3w_bench.hip && tools/x3w_bench
#include "../alpine_amd/csrc/kernels_x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace alpine;

__global__ void fill_kernel(float* __restrict__ x, size_t n, int mode, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761ull) ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        float v;
        if (mode == 0) { const unsigned r = h & 255u; v = r < 150u ? 0.f : (float)((h >> 8) & 7u); }        // counts: ~60 % zeros, small integers
        else if (mode == 1) v = (float)(h >> 8) * (1.0f / 16777216.0f) * 3.7f;                                  // full 24-bit significands
        else v = (float)(h >> 8) * (1.0f / 16777216.0f);                                                        // panel: U[0, 1)
        x[i] = v;
    }
}

struct Variant { const char* name; int kind; int gw; };     // kind 0 = x3w, 1 = x3 (32x32x16), 2 = x3w one-plane, 3 = x3w no test, 4 = x3v (two waves per SIMD), 5 = x3v one-plane

template <int KT, int NH, int M16A>
static void launch_once(const Variant& v, const float* S, const float* P, float* pieces, int64_t ldS, const SweepGeom& g)
{
#define ARGS dim3(sg_grid(g)), dim3(256), 0, 0, S, P, pieces, ldS, g, (int*)nullptr
    switch (v.kind) {
        case 0: hipLaunchKernelGGL((stream_gemm_x3w_kernel<KT, NH, M16A, false>), ARGS); break;
        case 1: hipLaunchKernelGGL((stream_gemm_x3_kernel<KT, NH>), ARGS); break;
        case 3: hipLaunchKernelGGL((stream_gemm_x3w_kernel<KT, NH, M16A, false, true>), ARGS); break;
        case 2: hipLaunchKernelGGL((stream_gemm_x3w_kernel<KT, NH, M16A, true>), ARGS); break;
        default:
            if constexpr (M16A * NH <= 8) {           // the two-waves-per-SIMD form
#define ARGS8 dim3(sg_grid(g)), dim3(512), 0, 0, S, P, pieces, ldS, g, (int*)nullptr
                if (v.kind == 4) hipLaunchKernelGGL((stream_gemm_x3v_kernel<KT, NH, M16A, false>), ARGS8);
                else hipLaunchKernelGGL((stream_gemm_x3v_kernel<KT, NH, M16A, true>), ARGS8);
#undef ARGS8
            }
            break;
    }
#undef ARGS
}

// Variants are timed INTERLEAVED (round r launches every variant once, in order; medians over the rounds): timed one after the other, 7
// launches each, the SECOND variant of a list came out 13 % faster than the rest whichever it was -- the chip's clock state drifts over
// the first seconds of a process.
template <int KT, int NH, int M16A>
static void shape(int64_t G, int64_t N, float* X, float* P, float* pieces, size_t piece_floats, float* out, float* ref, int reps, const std::vector<Variant>& all)
{
    constexpr int KP = 32 * KT, BF = 512 * NH;
    printf("#### K padded %d, %d of %d 16-component tiles, workgroup tiles of %d columns; X %lld x %lld\n", KP, M16A, 2 * KT, BF, (long long)G, (long long)N);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, P, (size_t)N * KP, 2, 77u);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int data = 0; data < 2; ++data) {
        hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, X, (size_t)G * N, data, 12345u);
        for (int orient = 0; orient < 2; ++orient) {
            const int64_t F = orient == 0 ? N : G, R = orient == 0 ? G : N;        // W^TX: rows = genes; XH^T: rows = cells (same bytes: a reinterpretation of X)
            printf("== %s X, F = %lld columns x R = %lld contraction rows (%s sweep)\n", data == 0 ? "count-like" : "full-significand", (long long)F, (long long)R,
                   orient == 0 ? "W^TX" : "XH^T");
            std::vector<Variant> variants; std::vector<SweepGeom> geoms; std::vector<double> rel;
            for (const Variant& v : all) {
                if ((v.kind == 2 || v.kind == 5) && data != 0) continue;
                if (v.kind == 3 && data != 1) continue;
                if (256 % (8 * v.gw) != 0 && v.gw != 1) continue;
                const SweepGeom g = sg_make_geom(F, R, 256 / v.gw, 0, BF * v.gw, 0, v.gw);
                if ((size_t)g.nwg * g.maxp * g.bf * KP > piece_floats) { printf("%-28s pieces buffer too small\n", v.name); continue; }
                variants.push_back(v); geoms.push_back(g);
            }
            // correctness pass: every variant's reduced result against the first one's
            for (size_t i = 0; i < variants.size(); ++i) {
                (void)hipMemset(pieces, 0, sizeof(float) * piece_floats);
                launch_once<KT, NH, M16A>(variants[i], X, P, pieces, F, geoms[i]);
                const int64_t n4 = F * KP / 4;
                hipLaunchKernelGGL(reduce_pieces_kernel, dim3((unsigned)std::min<int64_t>(2048, (n4 + 255) / 256)), dim3(256), 0, 0, pieces, i ? out : ref, (int)F, KP, geoms[i]);
                (void)hipDeviceSynchronize();
                double r = 0;
                if (i) {
                    std::vector<float> a((size_t)F * KP), b((size_t)F * KP);
                    (void)hipMemcpy(a.data(), out, sizeof(float) * a.size(), hipMemcpyDeviceToHost);
                    (void)hipMemcpy(b.data(), ref, sizeof(float) * b.size(), hipMemcpyDeviceToHost);
                    double num = 0, den = 0;
                    for (size_t k = 0; k < a.size(); ++k) { const double d = (double)a[k] - b[k]; num += d * d; den += (double)b[k] * b[k]; }
                    r = std::sqrt(num / std::max(den, 1e-300));
                }
                rel.push_back(r);
            }
            std::vector<std::vector<float>> ms(variants.size());
            for (int r = -3; r < reps; ++r)
                for (size_t i = 0; i < variants.size(); ++i) {
                    (void)hipEventRecord(e0);
                    launch_once<KT, NH, M16A>(variants[i], X, P, pieces, F, geoms[i]);
                    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                    float t; (void)hipEventElapsedTime(&t, e0, e1);
                    if (r >= 0) ms[i].push_back(t);
                }
            const hipError_t e = hipGetLastError();
            if (e != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(e)); exit(1); }
            for (size_t i = 0; i < variants.size(); ++i) {
                std::sort(ms[i].begin(), ms[i].end());
                const double med = ms[i][ms[i].size() / 2];
                const int64_t tiles = (F + BF - 1) / BF, padded = (tiles + variants[i].gw - 1) / variants[i].gw * variants[i].gw;
                printf("%-28s grid %4d (spans %4d x %d pieces, %2lld of %3lld members idle)  min %.3f med %.3f ms  %.2f TB/s  %.3f of 8 TB/s   rel-Frobenius vs first: %.2e\n",
                       variants[i].name, sg_grid(geoms[i]), geoms[i].nwg, geoms[i].maxp, (long long)(padded - tiles), (long long)padded, ms[i][0], med,
                       (double)F * R * 4.0 / (med * 1e-3) / 1e12, (double)F * R * 4.0 / (med * 1e-3) / 8e12, rel[i]);
            }
        }
    }
}

int main(int argc, char** argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 15;
    const int which = argc > 2 ? atoi(argv[2]) : 3;
    const int64_t G = 20096, Nmax = 200064;
    float *X, *P, *pieces, *out, *ref;
    (void)hipMalloc(&X, sizeof(float) * G * Nmax);
    (void)hipMalloc(&P, sizeof(float) * Nmax * 128);
    const size_t piece_floats = (size_t)600 * 3 * 512 * 128;
    (void)hipMalloc(&pieces, sizeof(float) * piece_floats);
    (void)hipMalloc(&out, sizeof(float) * Nmax * 128);
    (void)hipMalloc(&ref, sizeof(float) * Nmax * 128);
    if (which & 1)      // BASELINE config 4's per-GPU share: K = 105
        shape<4, 1, 7>(G, 125056, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w (round 3)", 0, 1}, {"x3w teams of 2", 0, 2}, {"x3w teams of 4", 0, 4}, {"x3w teams of 8", 0, 8}, {"x3w teams of 16", 0, 16}, {"x3w teams of 32", 0, 32},
                        {"x3w one-plane", 2, 1}, {"x3w one-plane teams of 8", 2, 8}, {"x3w no test", 3, 1}, {"x3w no test teams of 8", 3, 8}, {"x3w (again)", 0, 1}});
    if (which & 4)      // K = 105 again: the two-waves-per-SIMD form against x3w
        shape<4, 1, 7>(G, 125056, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w", 0, 1}, {"x3w teams of 8", 0, 8}, {"x3v", 4, 1}, {"x3v teams of 2", 4, 2}, {"x3v teams of 4", 4, 4}, {"x3v teams of 8", 4, 8},
                        {"x3w one-plane teams of 8", 2, 8}, {"x3v one-plane", 5, 1}, {"x3v one-plane teams of 8", 5, 8}, {"x3w (again)", 0, 1}});
    if (which & 8)      // K = 60 again: the two-waves-per-SIMD form against x3 / x3w
        shape<2, 2, 4>(G, 200064, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w", 0, 1}, {"x3 (32x32x16)", 1, 1}, {"x3 teams of 8", 1, 8}, {"x3v", 4, 1}, {"x3v teams of 2", 4, 2}, {"x3v teams of 4", 4, 4}, {"x3v teams of 8", 4, 8},
                        {"x3v one-plane", 5, 1}, {"x3v one-plane teams of 4", 5, 4}, {"x3v one-plane teams of 8", 5, 8}, {"x3w (again)", 0, 1}});
    if (which & 16)     // K = 128 (no padding tile) and K = 96 / 80 (three 32-component tiles): x3 (32x32x16) / x3w / x3v
        shape<4, 1, 8>(G, 125056, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w", 0, 1}, {"x3 (32x32x16)", 1, 1}, {"x3 teams of 8", 1, 8}, {"x3w teams of 8", 0, 8}, {"x3v", 4, 1}, {"x3v teams of 2", 4, 2}, {"x3v teams of 8", 4, 8},
                        {"x3w one-plane teams of 8", 2, 8}, {"x3v one-plane teams of 8", 5, 8}, {"x3w (again)", 0, 1}});
    if (which & 32) {
        shape<3, 1, 6>(G, 125056, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w", 0, 1}, {"x3 (32x32x16)", 1, 1}, {"x3 teams of 8", 1, 8}, {"x3w teams of 8", 0, 8}, {"x3v", 4, 1}, {"x3v teams of 2", 4, 2}, {"x3v teams of 8", 4, 8},
                        {"x3w one-plane teams of 8", 2, 8}, {"x3v one-plane teams of 8", 5, 8}, {"x3w (again)", 0, 1}});
        shape<3, 1, 5>(G, 125056, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w", 0, 1}, {"x3w teams of 8", 0, 8}, {"x3v", 4, 1}, {"x3v teams of 2", 4, 2}, {"x3v teams of 8", 4, 8},
                        {"x3w one-plane teams of 8", 2, 8}, {"x3v one-plane teams of 8", 5, 8}, {"x3w (again)", 0, 1}});
    }
    if (which & 64)     // K = 60 on a 25 000-cell shard, 512-column tiles: x3 (the library's choice there) / x3w / x3v
        shape<2, 1, 4>(G, 25088, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w", 0, 1}, {"x3 (32x32x16)", 1, 1}, {"x3 teams of 2", 1, 2}, {"x3 teams of 4", 1, 4}, {"x3 teams of 8", 1, 8}, {"x3v", 4, 1}, {"x3v teams of 2", 4, 2}, {"x3v teams of 4", 4, 4},
                        {"x3v teams of 8", 4, 8}, {"x3v one-plane teams of 4", 5, 4}, {"x3 (again)", 1, 1}});
    if (which & 128)    // K = 60 at 200 000 cells on 512-column tiles (the library takes 1024-column tiles there): x3 / x3v with teams
        shape<2, 1, 4>(G, 200064, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3w", 0, 1}, {"x3 (32x32x16)", 1, 1}, {"x3 teams of 4", 1, 4}, {"x3 teams of 8", 1, 8}, {"x3 teams of 16", 1, 16}, {"x3v", 4, 1}, {"x3v teams of 4", 4, 4},
                        {"x3v teams of 8", 4, 8}, {"x3v teams of 16", 4, 16}, {"x3v one-plane teams of 8", 5, 8}, {"x3 (again)", 1, 1}});
    if (which & 2)      // BASELINE config 3: K = 60
        shape<2, 2, 4>(G, 200064, X, P, pieces, piece_floats, out, ref, reps,
                       {{"x3 (round 3)", 1, 1}, {"x3 teams of 2", 1, 2}, {"x3 teams of 4", 1, 4}, {"x3 teams of 8", 1, 8}, {"x3 teams of 16", 1, 16},
                        {"x3w", 0, 1}, {"x3w teams of 4", 0, 4}, {"x3w teams of 8", 0, 8}, {"x3w one-plane", 2, 1}, {"x3w one-plane teams of 4", 2, 4}, {"x3w one-plane teams of 8", 2, 8},
                        {"x3w no test", 3, 1}, {"x3w no test teams of 4", 3, 4}, {"x3 (again)", 1, 1}});
    return 0;
}
