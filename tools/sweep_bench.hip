// Standalone harness around the library's streaming sweep kernel (diagnostics; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/sweep_bench tools/sweep_bench.hip
//   ./tools/sweep_bench [R=20096] [F=200064] [split=9] [reps=8] [stride0=0]
// Reports per variant: launch time, algorithmic-equivalent TFLOP/s at KP, in-kernel clock, MFMA cycles.
#include "../alpine_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace alpine;

static SweepGeom make_geom(int64_t F, int64_t R, int slots, int forced) { return sg_make_geom(F, R, slots, forced, SG_BLOCK_F, 0); }   // the library's own division

template <int KT, int RING, int PASSES>
static void run(const char* name, const float* S, const float* P, float* slab, int64_t ldS, int F, int R, int split, int reps,
                unsigned long long* clk)
{
    const SweepGeom g = make_geom(F, R, 512, split);
    const int grid = (g.nwg + g.sub - 1) / g.sub;
    const int rps = g.L;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms(reps);
    std::vector<unsigned long long> h(4 * grid);
    for (int i = 0; i < reps; ++i) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stream_gemm_kernel<KT, RING, PASSES>), dim3(grid), dim3(SG_THREADS), 0, 0, S, P, slab, ldS, g, clk);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[i], e0, e1);
    }
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 4 * grid, hipMemcpyDeviceToHost);
    double ghz = 0, wgcyc = 0; for (int i = 0; i < grid; ++i) { ghz += (double)h[4 * i] / (double)h[4 * i + 1] * 0.1; wgcyc += (double)h[4 * i]; }
    ghz /= grid; wgcyc /= grid;
    std::sort(ms.begin(), ms.end());
    const double med = ms[reps / 2];
    const double mfma = 2.0 * R * (double)F * (32 * KT) / 4096.0;                 // MFMA instructions in the launch
    const double mfma_per_wave = (double)(rps / 2) * 4 * KT;                       // per wave of a full work item
    printf("%-14s grid=%5d  min %.3f med %.3f ms  %.1f TF(KP)  clock %.3f GHz  cyc/MFMA/SIMD(chip)=%.1f  WG: %.0f cyc, %.1f cyc per own MFMA\n",
           name, grid, ms[0], med, 2.0 * R * (double)F * (32 * KT) / med / 1e9, ghz, med * 1e-3 * ghz * 1e9 * 1024.0 / mfma, wgcyc,
           wgcyc / mfma_per_wave);
    if (getenv("SWEEP_TIMELINE")) {
        // occupancy timeline: resident workgroups sampled every 50 us; start-time histogram per dispatch wave
        unsigned long long t0 = ~0ull, t1 = 0;
        for (int i = 0; i < grid; ++i) { t0 = std::min(t0, h[4 * i + 2]); t1 = std::max(t1, h[4 * i + 2] + h[4 * i + 1]); }
        printf("  kernel span by WG stamps: %.3f ms\n", (t1 - t0) * 1e-5);
        for (unsigned long long t = t0; t < t1; t += 20000) {
            int n = 0; for (int i = 0; i < grid; ++i) if (h[4 * i + 2] <= t && t < h[4 * i + 2] + h[4 * i + 1]) ++n;
            printf("  t=%.2f ms resident=%d\n", (t - t0) * 1e-5, n);
        }
        double dmin = 1e30, dmax = 0; for (int i = 0; i < grid; ++i) { double d = h[4 * i + 1] * 1e-5; dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
        printf("  WG duration min %.3f max %.3f ms\n", dmin, dmax);
        int xcc[8] = {0}; for (int i = 0; i < grid; ++i) xcc[h[4 * i + 3] & 7]++;
        printf("  WGs per XCC:"); for (int i = 0; i < 8; ++i) printf(" %d", xcc[i]); printf("\n");
    }
}

int main(int argc, char** argv)
{
    const int R = argc > 1 ? atoi(argv[1]) : 20096;
    const int F = argc > 2 ? atoi(argv[2]) : 200064;
    const int split = argc > 3 ? atoi(argv[3]) : 0;   // 0 = stream-K over 512 workgroups; >0 = pieces per tile
    const int reps = argc > 4 ? atoi(argv[4]) : 8;
    const int stride0 = argc > 5 ? atoi(argv[5]) : 0;
    constexpr int KT = 2; constexpr int KP = 64;
    float *S, *P, *slab; unsigned long long* clk;
    hipMalloc(&S, sizeof(float) * (size_t)R * F);
    hipMalloc(&P, sizeof(float) * (size_t)R * KP);
    hipMalloc(&slab, sizeof(float) * (size_t)8192 * 3 * SG_BLOCK_F * KP);
    hipMalloc(&clk, 32 * 65536);
    {   // random-ish fill (values matter for DVFS: never bench on zeros)
        std::vector<float> h((size_t)1 << 24);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) * (1.0f / 65536.0f);
        for (size_t off = 0; off < (size_t)R * F; off += h.size())
            hipMemcpy(S + off, h.data(), sizeof(float) * std::min(h.size(), (size_t)R * F - off), hipMemcpyHostToDevice);
        hipMemcpy(P, h.data(), sizeof(float) * (size_t)R * KP, hipMemcpyHostToDevice);
    }
    const int64_t ldS = stride0 ? 0 : F;
    printf("R=%d F=%d split=%d stride0=%d\n", R, F, split, stride0);
    for (int rep = 0; rep < 2; ++rep) {
        run<KT, 16, 1>("ring16 x1", S, P, slab, ldS, F, R, split, reps, clk);
        run<KT, 8, 4>("ring8 x4", S, P, slab, ldS, F, R, split, reps, clk);
        run<KT, 8, 2>("ring8 x2", S, P, slab, ldS, F, R, split, reps, clk);
    }
    return 0;
}
