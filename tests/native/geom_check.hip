// Host-only check of the stream-K index arithmetic that the four sweep kernels and every consumer of their pieces share
// (alpine_amd/csrc/kernels.hpp: sg_make_geom, SgWalk, sg_tile_pieces, sg_piece_offset): for random shapes, grids, forced
// divisions and even/odd biases,
//   * the segments of all workgroups cover every (tile, row) exactly once,
//   * no span is longer than SG_MAX_CHAIN rows, every segment is a whole number of SG_ROW_ALIGN-row stages,
//   * every segment's piece slot is unique and inside the nwg * maxp slots the buffers are sized for,
//   * sg_tile_pieces(ft) names exactly the spans that wrote a piece for tile ft, in ascending order, and sg_piece_offset /
//     sg_piece_offset_inner return the slot the writer used.
// Built and run by tests/test_sweep_geometry.py (hipcc, no GPU needed: no device call is made).
#include "../../alpine_amd/csrc/kernels.hpp"
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <vector>
using namespace alpine;

static int check(int64_t F, int64_t R, int slots, int forced, int bf, int bias)
{
    const SweepGeom g = sg_make_geom(F, R, slots, forced, bf, bias);
    const int KP = 32;
    const int64_t total = (int64_t)g.nft * g.R;
    const int grid = (g.nwg + g.sub - 1) / g.sub;
#define FAIL(...) do { fprintf(stderr, "F=%lld R=%lld slots=%d forced=%d bf=%d bias=%d (L=%d dL=%d sub=%d nwg=%d maxp=%d): ", (long long)F, (long long)R, slots, forced, bf, bias, g.L, g.dL, g.sub, g.nwg, g.maxp); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)
    if (g.L % SG_ROW_ALIGN || g.dL % SG_ROW_ALIGN || g.L + std::abs(g.dL) > SG_MAX_CHAIN || g.L - std::abs(g.dL) <= 0) FAIL("bad span lengths");
    std::vector<int64_t> covered;                       // (start, end) in (tile,row) space per segment
    std::map<int, std::vector<std::pair<int, int64_t>>> by_tile;     // tile -> (span, slot)
    std::set<int64_t> slots_used;
    int64_t expect = 0;
    for (int wg = 0; wg < grid; ++wg) {
        SgWalk w;
        sg_walk_init(w, g, wg);
        int ft, r0, r1; int64_t slot;
        int64_t span_rows = 0; int cur_span = -1;
        while (sg_walk_next(w, g, ft, r0, r1, slot)) {
            const int64_t a = (int64_t)ft * g.R + r0, b = (int64_t)ft * g.R + r1;
            if (a != expect) FAIL("segment starts at %lld, expected %lld (workgroup %d)", (long long)a, (long long)expect, wg);
            if (r1 <= r0 || (r1 - r0) % SG_ROW_ALIGN) FAIL("segment of %d rows", r1 - r0);
            expect = b;
            if (w.span != cur_span) { cur_span = w.span; span_rows = 0; }
            span_rows += r1 - r0;
            if (span_rows > SG_MAX_CHAIN) FAIL("span %d accumulates %lld rows", w.span, (long long)span_rows);
            if (w.span / g.sub != wg) FAIL("span %d walked by workgroup %d", w.span, wg);
            if (slot < 0 || slot >= (int64_t)g.nwg * g.maxp) FAIL("slot %lld outside %lld", (long long)slot, (long long)g.nwg * g.maxp);
            if (!slots_used.insert(slot).second) FAIL("slot %lld written twice", (long long)slot);
            by_tile[ft].push_back({w.span, slot});
        }
    }
    if (expect != total) FAIL("covered %lld of %lld rows", (long long)expect, (long long)total);
    for (int ft = 0; ft < g.nft; ++ft) {
        int lo, hi;
        sg_tile_pieces(g, ft, lo, hi);
        const auto& v = by_tile[ft];
        if ((int)v.size() != hi - lo + 1) FAIL("tile %d: %zu writers, consumers expect spans %d..%d", ft, v.size(), lo, hi);
        for (int i = 0; i < (int)v.size(); ++i) {
            if (v[i].first != lo + i) FAIL("tile %d: writer %d is span %d, expected %d", ft, i, v[i].first, lo + i);
            const int64_t off = sg_piece_offset(g, lo + i, ft, KP);
            if (off != v[i].second * g.bf * KP) FAIL("tile %d span %d: offset %lld vs slot %lld", ft, lo + i, (long long)off, (long long)v[i].second);
            if (i > 0 && sg_piece_offset_inner(g, lo + i, KP) != off) FAIL("tile %d span %d: inner offset differs", ft, lo + i);
        }
    }
    return 0;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 3000;
    std::mt19937_64 rng(12345);
    int bad = 0, checked = 0;
    // the shapes the library runs at: cfg3 both sweeps, its 8-GPU shard, cfg4 whole and its share, the fuzz shape of round 2
    const int64_t fixed[][2] = {{20096, 200064}, {200064, 20096}, {20096, 25088}, {25088, 20096}, {20096, 1000064}, {1000064, 20096},
                                {20096, 125056}, {125056, 20096}, {1152, 77696}, {77696, 1152}, {128, 128}, {128, 64}};
    for (auto& fr : fixed)
        for (int bf : {512, 1024})
            for (int slots : {256, 512})
                for (int forced : {0, 1, 3})
                    for (int bias : {0, 25, -30, 100}) { bad += check(fr[0], fr[1], slots, forced, bf, bias); ++checked; }
    for (int i = 0; i < n && !bad; ++i) {
        const int64_t F = 128 * (1 + rng() % 200), R = 64 * (1 + rng() % (i % 7 == 0 ? 20000 : 600));
        const int bf = (rng() & 1) ? 512 : 1024, slots = (rng() & 1) ? 256 : 512, forced = (int)(rng() % 5);
        const int bias = (rng() % 3 == 0) ? 0 : (int)(rng() % 121) - 60;
        bad += check(F, R, slots, forced, bf, bias);
        ++checked;
    }
    printf("%d geometries checked, %d bad\n", checked, bad);
    return bad ? 1 : 0;
}
