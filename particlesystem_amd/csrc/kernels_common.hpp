// kernels_common.hpp -- device helpers shared by the stage files (grid.hip, pairs.hip, apply.hip, lifecycle.hip, slab.hip).
//
// The step's kernels, hand-written for gfx950 (MI355X, CDNA4), by stage:
//   grid.hip       AoS <-> SoA, init_iframe (the per-frame counts are zeroed by the last kernel of the step before;
//                  k_frame_reset only for a frame that was not left clean), build_grid: k_hist_lds / k_scan /
//                  k_scatter_lds counting sort of the live slots by cell (streaming, HBM), k_sort_cells: the reference's
//                  cell-list order (ps.cpp:1510-1516), the snapshot in that order, the cell-overflow rule, the collision
//                  halo lists
//   pairs.hip      calc_forces' two neighbour loops (ps.cpp:1182-1263): k_collide_cell (collision flags from LDS bins),
//                  k_plan_force, k_pairs_balanced / k_pairs: 27-cell softened gravity, one wave per 64
//                  particles of one cell, neighbour bodies as scalar operands of packed fp32 instructions (or LDS tiles on
//                  a small share), serial fp32 accumulation in the reference's order (fp32 VALU bound; no MFMA: no
//                  contraction here, every pair needs its own rsqrt)
//   apply.hip      k_apply: death / survive / integrate / wrap / re-hash in slot order (ps.cpp:1210-1333; streaming, HBM)
//   lifecycle.hip  free-slot queues + relocation replayed in the reference's serial order (ps.cpp:1335-1374,
//                  app_common.cu:305-376): k_ops_hist / k_ops_scatter / k_replay_commit (lists of any length), k_moves_stage
//   slab.hip       the messages of a multi-GPU step: halo snapshots, force records, status records, all-pairs snapshot
//
// Reference arithmetic is reproduced operation for operation: every file is built with
// -ffp-contract=off; where the reference evaluates in double (EPS2 add, 0.5*a*t*t) so
// does this, except where an fp32 form is proven bit-identical (see pairs.hip).
// Citations: ps.cpp = source/code/src/particleSystem.cpp of the reference.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cstdlib>

#include "device_types.h"
#include "geometry.hpp"
#include "kernels.h"

namespace psamd {

#define PS_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)
#define PS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

static inline int blocks_for(size_t n, int threads, int cap = 4096)
{
    size_t b = (n + threads - 1) / threads;
    if (b > (size_t)cap) b = cap;
    return b < 1 ? 1 : (int)b;
}

// (d_i2, d_i1, d_i3): the cell itself, then the reference's 26 candidates in the
// order fill_cells probes them (app.cu:375-408).
static __constant__ signed char c_stencil[27][3] = {
    {0, 0, 0},
    {-1, 0, 0}, {+1, 0, 0},
    {-1, -1, 0}, {0, -1, 0}, {+1, -1, 0},
    {-1, +1, 0}, {0, +1, 0}, {+1, +1, 0},
    {-1, -1, -1}, {0, -1, -1}, {+1, -1, -1},
    {-1, 0, -1}, {0, 0, -1}, {+1, 0, -1},
    {-1, +1, -1}, {0, +1, -1}, {+1, +1, -1},
    {-1, -1, +1}, {0, -1, +1}, {+1, -1, +1},
    {-1, 0, +1}, {0, 0, +1}, {+1, 0, +1},
    {-1, +1, +1}, {0, +1, +1}, {+1, +1, +1},
};

// Blocks are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give each
// XCD one contiguous run of tasks so neighbouring cells' tiles hit the same L2.
__device__ __forceinline__ int xcd_contiguous(int b, int nb)
{
    const int xcd = b & 7, idx = b >> 3, q = nb >> 3, r = nb & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// inclusive prefix sum across the 64 lanes of a wave
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = (int)__lane_id();
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// Slots (or queue operations) one 1024-thread workgroup of the counting sorts owns, and the cells (or queue records) its
// LDS histogram covers (grid.hip: k_hist_lds / k_scatter_lds; lifecycle.hip: k_ops_hist / k_ops_scatter)
constexpr int LDS_CELLS = 8192;
constexpr int SLOTS_PER_WG = 4096;

// ps.cpp:1491: a slot takes part when 0 <= cell < NUM_CELLS.  A rank walks the slots it owns
// (storage index order = slot order); their particles sit in the cell layers of region 0 by
// construction (a particle that leaves them is handed to the rank that owns its new segment).
__device__ __forceinline__ int own_local_cell(const DevParams &P, int gc, FrameScalars *fs)
{
    if (gc < 0 || gc >= P.num_cells_global) return -1;
    const int lc = gc - P.reg_first[0] * P.G * P.G;
    if (lc < 0 || lc >= P.n_own_cells) { atomicOr(&fs->error, ERR_FOREIGN_CELL); return -1; }
    return lc;
}

// The chunk lists' capacity (ps.cpp:1502-1508): build_grid walks the slots in order and appends
// a live particle to the list of the chunk of its cell only while that list holds fewer than
// MAX_PARTICLES_PER_CHUNK ids (the ones the cell-overflow rule kills a moment later included);
// calc_forces walks the stored lists, so a particle ranked at or past the capacity among its
// chunk's particles in slot order is not aged, collided or integrated that step (others still
// meet it as a neighbour: it is in its cell's list and in T_DATA).  One workgroup per chunk, at
// work only if the chunk's count passed the capacity: it walks the chunk's 27 segments (the
// only slots that can hold its particles) in slot order and writes chunk_skip for every
// particle of the chunk.  Runs before k_sort_cells resets the slots of overflowing cells: as extra
// workgroups of the scatter launch (k_scatter_lds only reads cell[]), so the usual frame, in which
// no chunk is over, pays no launch for it.
__device__ __forceinline__ void chunk_cap_block(const DevParams &P, int ch, const int *__restrict__ chunk_count,
                                                const int *__restrict__ cell_arr, const CellInfo *__restrict__ celltab,
                                                const int2 *__restrict__ chunk_segs, uint8_t *__restrict__ chunk_skip,
                                                const int *__restrict__ before4)
{
    // before4 (slab): per segment type, the chunk's particles in the slots BEFORE this rank's segments of
    // that type -- every lower type wherever it lives, and the same type on the ranks below (a rank's
    // slot range per type follows the lower ranks'); the walk then covers the own segments only.  By then
    // the cell-overflow rule has reset its victims: k_sort_cells left their cell as -2 - cell for this
    // walk (they were in the chunk's list), k_apply puts -1 there.
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (!before4 && chunk_count[ch] <= P.max_per_chunk) return;
    int run = 0, cur_t = -1;                              // particles of the chunk in the slots before the current batch
    for (int sgi = 0; sgi < 27; sgi++) {
        const int2 sg = chunk_segs[ch * 27 + sgi];
        if (before4) {
            if (slot_index(P, sg.x) < 0) continue;        // another rank's segment
            const int t = sgi < 1 ? 0 : sgi < 7 ? 1 : sgi < 19 ? 2 : 3;      // slot order = type order: 1 interior, 6 face, 12 edge, 8 corner segments
            if (t != cur_t) { cur_t = t; run = before4[t]; }
        }
        for (int b = 0; b < sg.y; b += 1024) {
            const int slot = sg.x + b + tid;
            int c = -1;
            if (b + tid < sg.y) c = cell_arr[slot_index(P, slot)];
            if (c <= -2) c = -2 - c;
            const bool in = c >= 0 && c < P.num_cells_global && celltab[c].chunk == ch;
            const unsigned long long m = __ballot(in);
            if (lane == 0) wave_tot[wv] = __popcll(m);
            __syncthreads();
            int before = 0, total = 0;
            for (int k = 0; k < 16; k++) { if (k < wv) before += wave_tot[k]; total += wave_tot[k]; }
            if (in) chunk_skip[slot_index(P, slot)] = (run + before + __popcll(m & ((1ull << lane) - 1ull))) >= P.max_per_chunk ? 1 : 0;
            run += total;
            __syncthreads();
        }
    }
}

// Halo bookkeeping of the two-pass pair stage.  Cell axes: i2 ~ +x, i1 ~ -y, i3 ~ -z
// (set_pos_t, app.cu:117-158).  halo_dirs packs, two bits per axis (i2, i1, i3), whether a
// body lies within P.halo_reach of the low (1) or high (2) face of cell (i1, i2, i3) on that
// axis; an axis whose neighbour would be outside the grid reports 0 (the stencil is not periodic).
// A body whose position is not a number (a child born with the direction (0, 0, 0): 0/0, ps.cpp:1306-1333; the
// reference files it under one fixed cell from then on, see the conversion in k_apply) is a candidate of EVERY neighbour: the reference's test `dist > COLLISION_RADIUS`
// does not fail for it, so once it is no kid every adult that scans it collides with it (app_common.cu:269-301).
constexpr int HALO_ALL = 0x40;
__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u && (__float_as_uint(y) & 0x7f800000u) != 0x7f800000u &&
           (__float_as_uint(z) & 0x7f800000u) != 0x7f800000u;
}

__device__ __forceinline__ int halo_dirs_of_numbers(const DevParams &P, int i1, int i2, int i3, float x, float y, float z);
__device__ __forceinline__ int halo_dirs(const DevParams &P, int i1, int i2, int i3, float x, float y, float z)
{
    if (!finite3(x, y, z)) return HALO_ALL;
    return halo_dirs_of_numbers(P, i1, i2, i3, x, y, z);
}

// (x, y, z numbers; anything else reports 0)
__device__ __forceinline__ int halo_dirs_of_numbers(const DevParams &P, int i1, int i2, int i3, float x, float y, float z)
{
    const int G = P.G;
    const float cs = (float)P.cell_size, half = (float)(G / 2), reach = P.halo_reach;
    const float u2 = (x / cs + half - (float)i2) * cs, u1 = (-y / cs + half - (float)i1) * cs,
                u3 = (-z / cs + half - (float)i3) * cs;                         // offsets inside the cell, [0, cs)
    int n2 = u2 < reach ? 1 : (cs - u2 < reach ? 2 : 0), n1 = u1 < reach ? 1 : (cs - u1 < reach ? 2 : 0),
        n3 = u3 < reach ? 1 : (cs - u3 < reach ? 2 : 0);
    if ((n2 == 1 && i2 == 0) || (n2 == 2 && i2 == G - 1)) n2 = 0;
    if ((n1 == 1 && i1 == 0) || (n1 == 2 && i1 == G - 1)) n1 = 0;
    if ((n3 == 1 && i3 == 0) || (n3 == 2 && i3 == G - 1)) n3 = 0;
    return n2 | (n1 << 2) | (n3 << 4);
}

// Direction index 0..26 ((d3+1)*9 + (d1+1)*3 + (d2+1)) of the neighbour reached by moving
// along the axes in subset m (bit 0: i2, bit 1: i1, bit 2: i3), or -1 if the body is not
// near a face on one of those axes.
__device__ __forceinline__ int halo_dir_of_subset(int dirs, int m)
{
    const int n2 = dirs & 3, n1 = (dirs >> 2) & 3, n3 = (dirs >> 4) & 3;
    if (((m & 1) && !n2) || ((m & 2) && !n1) || ((m & 4) && !n3)) return -1;
    const int d2 = (m & 1) ? (n2 == 1 ? -1 : 1) : 0, d1 = (m & 2) ? (n1 == 1 ? -1 : 1) : 0,
              d3 = (m & 4) ? (n3 == 1 ? -1 : 1) : 0;
    return (d3 + 1) * 9 + (d1 + 1) * 3 + (d2 + 1);
}

// Calls fn(dir) for every neighbour a body with face bits `dirs` is listed with: the (up to seven) neighbours
// beyond the faces it is near -- the loop every body takes, unrolled -- or all 26 for HALO_ALL.
template <class F>
__device__ __forceinline__ void for_each_halo_dir(int dirs, F fn)
{
    if (dirs != HALO_ALL) {
#pragma unroll
        for (int m = 1; m < 8; m++) {
            const int dir = halo_dir_of_subset(dirs, m);
            if (dir >= 0) fn(dir);
        }
    } else {
        for (int dir = 0; dir < 27; dir++)
            if (dir != 13) fn(dir);
    }
}

// local index of the neighbour in direction `dir`, -1 if this rank does not hold it
__device__ __forceinline__ int halo_neighbour(const DevParams &P, int i1, int i2, int i3, int dir)
{
    const int d3 = dir / 9 - 1, d1 = (dir / 3) % 3 - 1, d2 = dir % 3 - 1;
    return local_cell(P, i3 + d3, i1 + d1, i2 + d2);
}

// The snapshot in sorted order is kept once, as four arrays (x, y, z, w_eff planes of snap_soa: what the
// scalar-load walk streams); where a lane wants one body's four values, this reads them from the planes.
// (Round 2 also kept them as an array of float4: 16 bytes per particle written and never needed.)
struct SnapSoa {
    const float *p;
    size_t cap;
    __device__ __forceinline__ float4 operator[](int i) const { return make_float4(p[i], p[cap + i], p[2 * cap + i], p[3 * cap + i]); }
};

// Collision candidates of the neighbour cells (k_collide_cell): a body within HALO_REACH of a
// face, edge or corner of its cell is listed in the halo of the cell(s) beyond it.  Two
// bodies in different cells can only collide (distance <= COLLISION_RADIUS < HALO_REACH) if
// each is in the other's halo.  Called by all threads of a workgroup for one cell whose `kept`
// bodies start at sorted index `start` (snap_soa / snap_cid / sorted_id already written): count the
// cell's contributions per direction, reserve the room with one global atomic per direction,
// then write the bodies.  s_halo / s_halo_base: 27 ints of LDS each.
__device__ __forceinline__ void list_in_neighbour_halos(const DevParams &P, int lc, int start, int kept,
                                                        const SnapSoa snap4, const int *__restrict__ snap_cid,
                                                        int *__restrict__ halo_count, float *__restrict__ halo_f,
                                                        int *__restrict__ halo_id, int *s_halo, int *s_halo_base,
                                                        bool counted)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    int i1, i2, i3;
    cell_coords(P, lc, i1, i2, i3);
    if (!counted) {                      // else the caller counted while it had the bodies in hand
        if (tid < 27) s_halo[tid] = 0;
        __syncthreads();
        for (int e = tid; e < kept; e += nthr) {
            if (snap_cid[start + e] < 0) continue;
            const float4 q = snap4[start + e];
            const int m3 = halo_dirs(P, i1, i2, i3, q.x, q.y, q.z);
            if (!m3) continue;
            for_each_halo_dir(m3, [&](int dir) { atomicAdd(&s_halo[dir], 1); });
        }
    }
    __syncthreads();
    if (tid < 27) {
        int base = -1;
        const int cnt = s_halo[tid];
        if (cnt > 0) {
            const int nc = halo_neighbour(P, i1, i2, i3, tid);
            if (nc >= 0) base = atomicAdd(&halo_count[nc], cnt);
        }
        s_halo_base[tid] = base;
        s_halo[tid] = 0;
    }
    __syncthreads();
    for (int e = tid; e < kept; e += nthr) {
        const int id = snap_cid[start + e];
        if (id < 0) continue;
        const float4 q = snap4[start + e];
        const int m3 = halo_dirs(P, i1, i2, i3, q.x, q.y, q.z);
        if (!m3) continue;
        for_each_halo_dir(m3, [&](int dir) {
            if (s_halo_base[dir] < 0) return;
            const int k = s_halo_base[dir] + atomicAdd(&s_halo[dir], 1);
            if (k < HALO_CAP) {
                const size_t at = (size_t)halo_neighbour(P, i1, i2, i3, dir) * HALO_CAP + k, plane = (size_t)P.n_local_cells * HALO_CAP;
                halo_f[at] = q.x; halo_f[plane + at] = q.y; halo_f[2 * plane + at] = q.z;
                halo_id[at] = id;
            }
        });
    }
}

// Where a particle's force record (ax, ay, az, collision flag) goes.  For the OWN cells the acceleration goes straight
// into the particle's own record -- acc4[slot].xyz, the fertility age in .w stays -- and the flag into a byte by slot:
// k_apply streams its particles by slot and finds the new acceleration where the particle keeps it, so the record is
// neither copied nor read twice (until round 5 the records were an array of their own in sorted order, gathered by
// k_apply through rank_of_slot and copied into acc4: 32 bytes of traffic per moved particle and a dependent round trip
// in a kernel that is bound by its bytes).  Nothing reads a particle's old acceleration during a step -- with ONE
// exception: a particle ranked past its chunk list's capacity is not touched by calc_forces at all (ps.cpp:1502-1508;
// chunk_cap_block), its record keeps last step's acceleration: such a particle's record is not written (the rule's
// verdict, chunk_skip, is in before the pair stage: k_scan on one GPU, the chunk census of slab_pairs on a slab).
// The records of cells computed for the rank below (the lent region) stay in sorted order, where k_pack_force takes them
// from; and the sorted-order array is also the mailbox through which a task that is cut hands its partial sums from
// wave to wave (`buf + gi`).
struct ForceBuf {
    float4 *sorted;
    float4 *acc4;
    uint8_t *flag_slot;
    const int *sorted_id;
    const FrameScalars *fs;              // chunk_over: some chunk list is past its capacity this frame (rare)
    const uint8_t *chunk_skip;
    const int *chunk_count;
    const CellInfo *celltab;
    const int *cell_arr;
    __device__ __forceinline__ float4 *operator+(int gi) const { return sorted + gi; }
    __device__ __forceinline__ bool untouched(const DevParams &P, int si) const      // calc_forces does not process this particle this step
    {
        if (!fs->chunk_over) return false;
        int c = cell_arr[si];
        if (c <= -2) c = -2 - c;
        return c >= 0 && c < P.num_cells_global && chunk_count[celltab[c].chunk] > P.max_per_chunk && chunk_skip[si];
    }
    __device__ __forceinline__ void put_slot(const DevParams &P, int si, const float4 v) const
    {
        if (si < 0 || untouched(P, si)) return;
        *reinterpret_cast<float3 *>(acc4 + si) = make_float3(v.x, v.y, v.z);
        flag_slot[si] = (uint8_t)__float_as_int(v.w);
    }
    // lc: the (local) cell the particle at sorted index gi belongs to
    __device__ __forceinline__ void put(const DevParams &P, int lc, int gi, const float4 v) const
    {
        if (lc < P.n_own_cells) put_slot(P, slot_index(P, sorted_id[gi]), v); else sorted[gi] = v;
    }
    __device__ __forceinline__ void put_id(const DevParams &P, int lc, int gi, int id, const float4 v) const     // (the caller has the slot id at hand)
    {
        if (lc < P.n_own_cells) put_slot(P, slot_index(P, id), v); else sorted[gi] = v;
    }
    __device__ __forceinline__ float4 get(const DevParams &P, int lc, int gi) const
    {
        if (lc >= P.n_own_cells) return sorted[gi];
        const int si = slot_index(P, sorted_id[gi]);
        const float4 a = acc4[si];
        return make_float4(a.x, a.y, a.z, __int_as_float((int)flag_slot[si]));
    }
};
inline ForceBuf force_buf(const DeviceState &d)
{
    return ForceBuf{d.force4, d.acc4, d.flag_slot, d.sorted_id, d.fs, d.chunk_skip, d.chunk_count, d.celltab, d.cell};
}

// the four outboxes of a slab: records for the rank below [0] / above [1] (xfer_cap each), two ranks below [2] / above [3] (xfer2_cap)
constexpr int FAR_MAGIC = 0x21524146;       // "FAR!": header word 3 of a far outbox that was closed this step
struct Outboxes { XferRec *o[5]; };         // below, above, two below, two above, far (all-gathered)
struct OutboxMsgs { int *m[5]; };        // the messages the outboxes live in (their headers), null where there is none

__device__ __forceinline__ float clamp_mag(float v, float lim)   // ps.cpp:1279-1281, 1294-1296
{
    if (fabsf(v) > lim) v = lim * (v / fabsf(v));
    return v;
}

__device__ __forceinline__ int segment_record_of_slot(const SegLayout &S, int slot)
{
    int k = 0;
    while (k < 3 && slot >= S.seg_base[k + 1]) k++;
    return S.info_base[k] + (slot - S.seg_base[k]) / S.seg_size_t[k];
}

__device__ __forceinline__ int segment_record(const SegLayout &S, int seg_type, int seg_tid)
{
    const int k = seg_type == 1 ? 0 : seg_type == 2 ? 1 : seg_type == 4 ? 2 : 3;
    return S.info_base[k] + seg_tid;
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace psamd
