// partition.hpp -- slab partition of the particle system over the GPUs of one node.
//
// The reference distributes by SEGMENT: a chunk's working set is its interior segment plus
// the 26 face / edge / corner segments around it (ps.cpp:380-487, set_pkg_segments
// app_common.cu:150-232), every segment being one contiguous slot range with its own
// free-slot queue.  The same unit is used here, cut along one axis only: i3, the slowest
// index of cell = i3*G*G + i1*G + i2 (app.cu:157), so that a rank's cells are one contiguous
// run of the cell-major sorted order.
//
// Along i3 the segments fall into 2F+1 "groups" (F = CHUNK_FACTOR, D = CHUNK_DIM):
//   plane p (group 2p)   cell layers {pD-1, pD} inside the grid: the segments that straddle the
//                        boundary between chunk layers p-1 and p (faces normal to i3, the
//                        edges lying in that plane, the corners);
//   inner k (group 2k+1) cell layers {kD+1 .. kD+D-2}: the chunk interiors of chunk layer k and
//                        the faces / edges that stay inside it.
// The reference numbers the segments of every type plane by plane (get_cell_info,
// app_common.cu:50-148), so a contiguous run of groups owns ONE contiguous slot range and one
// contiguous run of QUEUE_INFO records per segment type.
//
// Two partitions are derived, both contiguous in i3:
//   compute layers [cut_lo, cut_hi)   whose particles this rank evaluates collisions and
//                                     forces for; cut anywhere, balanced by stencil cost;
//   state layers   [state_lo, state_hi)  the groups whose particles, slots and queues live
//                                     on this rank: a group belongs to the rank that computes
//                                     its lowest layer, so every queue has exactly one owner
//                                     and is replayed there in the reference's serial order.
// Where a cut falls inside a group, the group's upper layers are computed by the rank above:
// their snapshot goes up with the halo layer ("lent" layers) and their (ax, ay, az, flag)
// records come back before the owner integrates them.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace psamd {

struct SlabPlan {
    int world = 1, rank = 0;
    int G = 0, D = 0, F = 0;
    int cut_lo = 0, cut_hi = 0;          // compute layers
    int state_lo = 0, state_hi = 0;      // layers whose particles live here
    int group_lo = 0, group_hi = 0;      // owned groups [group_lo, group_hi)
    int below_lo = 0, below_hi = 0;      // layers held from rank-1: halo layer + lent-in layers
    int above_lo = 0, above_hi = 0;      // layer held from rank+1: halo (only when the upper cut is group-aligned)
    int lentin_lo = 0, lentin_hi = 0;    // subset of below: layers this rank computes for rank-1
    int lentout_lo = 0, lentout_hi = 0;  // own state layers computed by rank+1
    int send_up_lo = 0, send_up_hi = 0;      // own layers whose snapshot goes to rank+1
    int send_down_lo = 0, send_down_hi = 0;  // own layers whose snapshot goes to rank-1
    int slot_lo[4] = {0, 0, 0, 0}, slot_hi[4] = {0, 0, 0, 0};   // owned slot range per segment type
    int rec_lo[4] = {0, 0, 0, 0}, rec_hi[4] = {0, 0, 0, 0};     // owned QUEUE_INFO records per segment type
    int up_rank = -1, down_rank = -1;    // ring neighbours for relocation traffic (periodic box); -1 when world == 1
    bool valid = false;
};

// First / one-past-last cell layer of group g.
inline void group_layers(int g, int D, int G, int &lo, int &hi)
{
    if (g & 1) { const int k = g >> 1; lo = k * D + 1; hi = k * D + D - 1; }
    else { const int p = g >> 1; lo = std::max(0, p * D - 1); hi = std::min(G, p * D + 1); }
}

inline int group_of_layer(int i3, int D)
{
    const int r = i3 % D, k = i3 / D;
    if (r == 0) return 2 * k;
    if (r == D - 1) return 2 * (k + 1);
    return 2 * k + 1;
}

// Segment ids [lo, hi) of type index t (0..3 = types 1, 2, 4, 8) that belong to group g.
inline void group_segments(int g, int t, int F, int &lo, int &hi)
{
    const int FF = F * F, S2 = 2 * F * (F + 1), E = (F + 1) * (F + 1);
    const int k = g >> 1;
    if (g & 1) {            // inner k
        switch (t) {
        case 0: lo = k * FF; hi = lo + FF; break;
        case 1: lo = k * (FF + S2) + FF; hi = lo + S2; break;
        case 2: lo = k * (S2 + E) + S2; hi = lo + E; break;
        default: lo = hi = (k + 1) * E; break;       // no corners inside a chunk layer
        }
    } else {                // plane k
        switch (t) {
        case 0: lo = hi = k * FF; break;             // no chunk interiors on a plane
        case 1: lo = k * (FF + S2); hi = lo + FF; break;
        case 2: lo = k * (S2 + E); hi = lo + S2; break;
        default: lo = k * E; hi = lo + E; break;
        }
    }
}

// Relative cost of computing one cell layer: the stencil is not periodic (app.cu:352-368),
// the two outermost layers see 18 of 27 cells.
inline double layer_cost(int i3, int G) { return (i3 == 0 || i3 == G - 1) ? 2.0 / 3.0 : 1.0; }

// Compute cuts for `world` ranks: contiguous, at least two layers each (a particle moves at
// most one cell per step, MAX_DX = CELL_SIZE; two layers keep every exchange between ring
// neighbours), minimising the most expensive share.  Returns world+1 boundaries or empty.
inline std::vector<int> slab_cuts(int G, int world)
{
    if (world < 1 || G < 2 * world) return {};
    std::vector<double> pre((size_t)G + 1, 0.0);
    for (int i = 0; i < G; i++) pre[(size_t)i + 1] = pre[(size_t)i] + layer_cost(i, G);
    const double INF = 1e300;
    // best[r][e]: smallest possible maximum over the first r ranks covering layers [0, e)
    std::vector<std::vector<double>> best((size_t)world + 1, std::vector<double>((size_t)G + 1, INF));
    std::vector<std::vector<int>> from((size_t)world + 1, std::vector<int>((size_t)G + 1, -1));
    best[0][0] = 0.0;
    for (int r = 1; r <= world; r++)
        for (int e = 2 * r; e <= G - 2 * (world - r); e++)
            for (int b = 2 * (r - 1); b <= e - 2; b++) {
                if (best[(size_t)r - 1][(size_t)b] >= INF) continue;
                const double v = std::max(best[(size_t)r - 1][(size_t)b], pre[(size_t)e] - pre[(size_t)b]);
                // ties: prefer the later cut (keeps the first ranks, which hold the cheap outer layer, full)
                if (v < best[(size_t)r][(size_t)e] - 1e-12 || (v < best[(size_t)r][(size_t)e] + 1e-12 && b > from[(size_t)r][(size_t)e])) {
                    best[(size_t)r][(size_t)e] = v; from[(size_t)r][(size_t)e] = b;
                }
            }
    if (best[(size_t)world][(size_t)G] >= INF) return {};
    std::vector<int> cuts((size_t)world + 1);
    int e = G;
    for (int r = world; r >= 1; r--) { cuts[(size_t)r] = e; e = from[(size_t)r][(size_t)e]; }
    cuts[0] = 0;
    return cuts;
}

// The plan of one rank.  seg_base / seg_size_t / info_base as in Geometry.  `cuts_in` (world+1
// boundaries) overrides the balanced cuts, e.g. for tests.
inline SlabPlan make_slab_plan(int F, int D, const int seg_base[5], const int seg_size_t[4], const int info_base[5],
                               int rank, int world, const int *cuts_in = nullptr)
{
    SlabPlan p;
    p.world = world; p.rank = rank; p.G = F * D; p.D = D; p.F = F;
    const int G = p.G, NG = 2 * F + 1;
    if (world < 1 || rank < 0 || rank >= world) return p;
    std::vector<int> cuts;
    if (cuts_in) cuts.assign(cuts_in, cuts_in + world + 1);
    else cuts = slab_cuts(G, world);
    if ((int)cuts.size() != world + 1 || cuts[0] != 0 || cuts[(size_t)world] != G) return p;
    for (int r = 0; r < world; r++) if (cuts[(size_t)r + 1] - cuts[(size_t)r] < (world > 1 ? 2 : 1)) return p;
    // group -> owner: the rank computing the group's lowest layer
    auto owner_of_layer = [&](int i3) { int r = 0; while (i3 >= cuts[(size_t)r + 1]) r++; return r; };
    std::vector<int> gown((size_t)NG);
    for (int g = 0; g < NG; g++) { int lo, hi; group_layers(g, D, G, lo, hi); gown[(size_t)g] = owner_of_layer(lo); }
    auto groups_of = [&](int r, int &g0, int &g1) {
        g0 = NG; g1 = 0;
        for (int g = 0; g < NG; g++) if (gown[(size_t)g] == r) { g0 = std::min(g0, g); g1 = std::max(g1, g + 1); }
        return g0 < g1;
    };
    auto state_of = [&](int r, int &lo, int &hi) {
        int g0, g1;
        if (!groups_of(r, g0, g1)) return false;
        int a, b;
        group_layers(g0, D, G, lo, a);
        group_layers(g1 - 1, D, G, b, hi);
        return true;
    };
    for (int r = 0; r < world; r++) { int a, b; if (!state_of(r, a, b)) return p; }   // every rank must own state
    p.cut_lo = cuts[(size_t)rank]; p.cut_hi = cuts[(size_t)rank + 1];
    groups_of(rank, p.group_lo, p.group_hi);
    state_of(rank, p.state_lo, p.state_hi);
    // snapshot layers needed: [cut_lo - 1, cut_hi + 1) inside the grid
    const int need_lo = std::max(0, p.cut_lo - 1), need_hi = std::min(G, p.cut_hi + 1);
    p.below_lo = std::min(need_lo, p.state_lo); p.below_hi = p.state_lo;
    p.above_lo = p.state_hi; p.above_hi = std::max(need_hi, p.state_hi);
    p.lentin_lo = std::max(p.below_lo, p.cut_lo); p.lentin_hi = p.below_hi;
    if (p.lentin_lo > p.lentin_hi) p.lentin_lo = p.lentin_hi;
    p.lentout_lo = std::min(p.cut_hi, p.state_hi); p.lentout_hi = p.state_hi;
    // what the neighbours hold of mine
    p.send_up_lo = p.send_up_hi = p.send_down_lo = p.send_down_hi = 0;
    if (rank + 1 < world) {
        int s_lo, s_hi; state_of(rank + 1, s_lo, s_hi);
        const int n_lo = std::max(0, cuts[(size_t)rank + 1] - 1);
        p.send_up_lo = std::min(n_lo, s_lo); p.send_up_hi = s_lo;           // == rank+1's below region
        if (p.send_up_lo < p.state_lo) return p;                           // would need layers of rank-1: unsupported cut
    }
    if (rank > 0) {
        int s_lo, s_hi; state_of(rank - 1, s_lo, s_hi);
        const int n_hi = std::min(G, cuts[(size_t)rank] + 1);
        p.send_down_lo = s_hi; p.send_down_hi = std::max(n_hi, s_hi);      // == rank-1's above region
        if (p.send_down_hi > p.state_hi) return p;
    }
    if (p.below_lo < p.below_hi) {          // must all live on rank-1
        if (rank == 0) return p;
        int s_lo, s_hi; state_of(rank - 1, s_lo, s_hi);
        if (p.below_lo < s_lo || p.below_hi != s_hi) return p;
    }
    if (p.above_lo < p.above_hi) {
        if (rank + 1 >= world) return p;
        int s_lo, s_hi; state_of(rank + 1, s_lo, s_hi);
        if (p.above_lo != s_lo || p.above_hi > s_hi) return p;
    }
    if (p.lentout_lo < p.lentout_hi && rank + 1 >= world) return p;
    for (int t = 0; t < 4; t++) {
        int a, b, c, d;
        group_segments(p.group_lo, t, F, a, b);
        group_segments(p.group_hi - 1, t, F, c, d);
        (void)b; (void)c;
        p.rec_lo[t] = info_base[t] + a; p.rec_hi[t] = info_base[t] + d;
        p.slot_lo[t] = seg_base[t] + a * seg_size_t[t]; p.slot_hi[t] = seg_base[t] + d * seg_size_t[t];
    }
    if (world > 1) { p.up_rank = (rank + 1) % world; p.down_rank = (rank + world - 1) % world; }
    p.valid = true;
    return p;
}

}  // namespace psamd
