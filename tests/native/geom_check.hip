// Host-only check of the stream-K index arithmetic that the four sweep kernels and every consumer of their pieces share
// (alpine_amd/csrc/kernels.hpp: sg_make_geom, SgWalk, sg_tile_pieces, sg_piece_offset): for random shapes, grids, forced
// divisions and even/odd biases,
//   * the segments of all workgroups cover every (tile, row) exactly once,
//   * no span is longer than SG_MAX_CHAIN rows, every segment is a whole number of SG_ROW_ALIGN-row stages,
//   * every segment's piece slot is unique and inside the nwg * maxp slots the buffers are sized for,
//   * sg_tile_pieces(ft) names exactly the spans that wrote a piece for tile ft, in ascending order, and sg_piece_offset /
//     sg_piece_offset_inner return the slot the writer used,
//   * teams (SweepGeom::gw > 1, the x3 sweeps): sg_team_of_block is a bijection of the launch grid (sg_grid) onto (team, member),
//     the members of a team have equal blockIdx % 8, teams past the last span walk nothing, and the members' column ranges
//     [member, member + 1) * bf / gw of every tile the team walks cover [0, F) exactly once per contraction row.
// Built and run by tests/test_sweep_geometry.py (hipcc, no GPU needed: no device call is made).
#include "../../alpine_amd/csrc/kernels.hpp"
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <vector>
using namespace alpine;

// bf = columns of ONE workgroup's tile; gw = team width (the geometry's tile is gw * bf columns wide and `slots / gw` teams share the work)
static int check(int64_t F, int64_t R, int slots, int forced, int bf, int bias, int gw = 1)
{
    const SweepGeom g = sg_make_geom(F, R, std::max(1, slots / gw), forced, bf * gw, bias, gw);
    const int KP = 32;
    const int64_t total = (int64_t)g.nft * g.R;
    const int grid = (g.nwg + g.sub - 1) / g.sub;          // teams that own spans
#define FAIL(...) do { fprintf(stderr, "F=%lld R=%lld slots=%d forced=%d bf=%d bias=%d gw=%d (L=%d dL=%d sub=%d nwg=%d maxp=%d): ", (long long)F, (long long)R, slots, forced, bf, bias, gw, g.L, g.dL, g.sub, g.nwg, g.maxp); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return 1; } while (0)
    {
        // launch grid -> (team, member): a bijection onto [0, teams padded to a multiple of 8) x [0, gw); members share blockIdx % 8
        const int launch = sg_grid(g);
        if (g.gw != gw || g.bf != bf * gw) FAIL("geometry does not carry the team width");
        if (gw == 1 ? launch != grid : (launch % (8 * gw) != 0 || launch < grid * gw || launch >= (grid + 8) * gw)) FAIL("launch grid %d for %d teams of %d", launch, grid, gw);
        std::set<std::pair<int, int>> seen;
        std::map<int, int> xcd_of_team;
        for (int b = 0; b < launch; ++b) {
            int team, member;
            sg_team_of_block(g, b, team, member);
            if (team < 0 || team >= launch / gw || member < 0 || member >= gw) FAIL("block %d -> team %d member %d", b, team, member);
            if (!seen.insert({team, member}).second) FAIL("block %d: (team %d, member %d) assigned twice", b, team, member);
            if (gw > 1) {
                auto it = xcd_of_team.find(team);
                if (it == xcd_of_team.end()) xcd_of_team[team] = b % 8;
                else if (it->second != b % 8) FAIL("team %d has members on blocks with different blockIdx %% 8", team);
                if ((team & 1) != (b & 1)) FAIL("team %d parity differs from its blocks' parity (the even/odd span bias keys on it)", team);
            }
            if (team >= grid) {                          // padding teams: nothing to walk
                SgWalk w; sg_walk_init(w, g, team);
                int ft, r0, r1; int64_t slot;
                if (sg_walk_next(w, g, ft, r0, r1, slot)) FAIL("padding team %d walks a segment", team);
            }
        }
        // the members' column ranges of a team tile: member j owns [j * bf, (j + 1) * bf) of the tile; ranges starting at or past F are skipped
        for (int ft = 0; ft < g.nft; ++ft) {
            int64_t next = (int64_t)ft * g.bf;
            for (int j = 0; j < gw; ++j) {
                const int64_t c0 = ((int64_t)ft * gw + j) * bf;
                if (c0 >= F) continue;
                if (c0 != next) FAIL("tile %d member %d starts at column %lld, expected %lld", ft, j, (long long)c0, (long long)next);
                next = std::min<int64_t>(F, c0 + bf);
            }
            if (next != std::min<int64_t>(F, ((int64_t)ft + 1) * g.bf)) FAIL("tile %d: members cover columns up to %lld", ft, (long long)next);
        }
    }
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
    for (auto& fr : fixed)
        for (int bf : {512, 1024})
            for (int gw : {2, 4, 8, 16, 32})
                for (int bias : {0, 40, -40}) { bad += check(fr[0], fr[1], 256, 0, bf, bias, gw); ++checked; }
    for (int i = 0; i < n && !bad; ++i) {
        const int64_t F = 128 * (1 + rng() % 200), R = 64 * (1 + rng() % (i % 7 == 0 ? 20000 : 600));
        const int bf = (rng() & 1) ? 512 : 1024, slots = (rng() & 1) ? 256 : 512, forced = (int)(rng() % 5);
        const int bias = (rng() % 3 == 0) ? 0 : (int)(rng() % 121) - 60;
        bad += check(F, R, slots, forced, bf, bias);
        ++checked;
        if (i % 2 == 0) { const int gw = 1 << (1 + rng() % 4); bad += check(F, R, 256, (int)(rng() % 3), bf, bias, gw); ++checked; }
    }
    printf("%d geometries checked, %d bad\n", checked, bad);
    return bad ? 1 : 0;
}
