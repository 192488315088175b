// Diagnostic: streaming-read bandwidth with the ACCESS PATTERN of the sweeps instead of one contiguous span per
// workgroup: a workgroup owns a column panel of `panel_bytes` per row of a row-major matrix with row pitch `ld_bytes`
// and walks down the rows (stream-K span of `rows` rows).  mode 0: 256 threads read one 4 KiB row segment per step
// (16 B per thread); mode 1: the x3 kernel's pattern (a wave owns 1 KiB of the segment as 2 x 512 B: lanes 0-31 row r,
// lanes 32-63 row r+8; 8 rows per group of 16).  Compare with tools/read_bw (contiguous: 7.2 TB/s).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int U, int MODE>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ x, size_t ld_bytes, int n_panels, int rows_total, int rows_per_wg, float* out)
{
    const long span0 = (long)blockIdx.x * rows_per_wg;           // position in (panel, row) space, panel-major
    const int panel = (int)(span0 / rows_total);
    const int r0 = (int)(span0 % rows_total);
    if (panel >= n_panels) return;
    const int r1 = min(rows_total, r0 + rows_per_wg);
    const char* base = x + (size_t)panel * 4096;
    f32x4 acc = {0, 0, 0, 0};
    const int t = threadIdx.x;
    if (MODE == 0) {
        for (int r = r0; r + U <= r1; r += U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)(r + u) * ld_bytes + 16 * t));
#pragma unroll
            for (int u = 0; u < U; ++u) acc += v[u];
        }
    } else {
        const int wave = t >> 6, lane = t & 63, c = lane & 31, h = lane >> 5;
        for (int r = r0; r + 16 <= r1; r += 16) {              // 16 rows x (2 halves of 512 B) per wave
            f32x4 v[16];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    v[hf * 8 + e] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(base + (size_t)(r + 8 * h + e) * ld_bytes + wave * 1024 + hf * 512 + 16 * c));
#pragma unroll
            for (int u = 0; u < 16; ++u) acc += v[u];
        }
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[blockIdx.x] = acc[0];
}

template <int U, int MODE>
void run(const char* x, size_t ld_bytes, int rows_total, int grid, float* out, const char* name)
{
    const int n_panels = (int)(ld_bytes / 4096);
    const long total = (long)n_panels * rows_total;
    int rows_per_wg = (int)((total + grid - 1) / grid);
    rows_per_wg = (rows_per_wg + 63) / 64 * 64;
    const int g = (int)((total + rows_per_wg - 1) / rows_per_wg);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((rd<U, MODE>), dim3(g), dim3(256), 0, 0, x, ld_bytes, n_panels, rows_total, rows_per_wg, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double bytes = (double)n_panels * 4096.0 * rows_total;
    printf("%-22s ld=%8zu B rows=%7d grid=%4d: %.3f ms  %.2f TB/s\n", name, ld_bytes, rows_total, g, best, bytes / (best * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)17 << 30;
    char* x; float* out;
    hipMalloc(&x, bytes); hipMalloc(&out, 1 << 20);
    hipMemset(x, 0, bytes);
    // cfg3: X_ng = 200704 rows x 20096 cols (pitch 80384 B -> 19 full 4 KiB panels), X_gn = 20096 rows x 200704 cols (pitch 802816 B)
    for (int grid : {256, 512}) {
        run<8, 0>(x, 80384, 200704, grid, out, "X_ng rows=cells WG4K");
        run<16, 0>(x, 80384, 200704, grid, out, "X_ng rows=cells WG4K");
        run<16, 1>(x, 80384, 200704, grid, out, "X_ng x3 pattern");
        run<8, 0>(x, 802816, 20096, grid, out, "X_gn rows=genes WG4K");
        run<16, 1>(x, 802816, 20096, grid, out, "X_gn x3 pattern");
        run<8, 0>(x, 4096, 4000000, grid, out, "panel-major (contig)");
        run<16, 1>(x, 4096, 4000000, grid, out, "panel-major x3 pat");
    }
    return 0;
}
