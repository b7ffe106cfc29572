// kernels.hip -- hand-written gfx950 (MI355X, CDNA4) kernels for the particle step.
//
// Step pipeline (all on the context's stream):
//   k_hist_lds / k_scatter_lds   counting sort of the live slots by cell, histogram
//               private to a workgroup in LDS (a window of 8192 cells from the first
//               cell the workgroup meets)                               (streaming, HBM)
//   k_scan      prefix over cells, hostGridMax, the collide work list
//   k_sort_cells   rank ids inside each cell (ascending = the reference's cell-list
//               order, ps.cpp:1510-1516), gather the T_DATA snapshot in that order
//   k_pairs     27-cell softened gravity + collision flags: one WAVE per 64 particles of
//               one cell; neighbour bodies arrive by scalar loads (SGPR operands of packed
//               fp32 instructions), serial fp32 accumulation in the reference's order (fp32
//               VALU bound; no MFMA: no contraction here, every pair needs its own rsqrt)
//   k_apply     death / survive / integrate / wrap / re-hash, in slot order (streaming, HBM)
//   k_ops_hist / k_ops_scan / k_ops_scatter / k_replay_bucket / k_moves_*
//               free-slot queues + relocation, replayed in the reference's serial order
//               (k_replay + a radix sort of the keys when one queue gets a very long list)
//
// Reference arithmetic is reproduced operation for operation: this file is built with
// -ffp-contract=off; where the reference evaluates in double (EPS2 add, 0.5*a*t*t) so
// does this, except where an fp32 form is proven bit-identical (see k_pairs).
// Citations: ps.cpp = source/code/src/particleSystem.cpp of the reference.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cstdlib>

#include "device_types.h"
#include "geometry.hpp"
#include "kernels.h"

namespace psamd {

// (d_i2, d_i1, d_i3): the cell itself, then the reference's 26 candidates in the
// order fill_cells probes them (app.cu:375-408).
__constant__ signed char c_stencil[27][3] = {
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

// ------------------------------------------------------------------ AoS <-> SoA
// P_DATA_TYPE is 18 dwords (common.h:94-120): id cell chunk seg_type seg_tid
// {seg_fault,is_parent,pad,pad} w age fert x y z vx vy vz ax ay az.
__global__ void k_unpack_aos(DevParams P, const uint32_t *__restrict__ aos, int first, int count, float half_box,
                             float4 *pos4, float4 *vel4, float4 *acc4, int *cell, uint8_t *pflags,
                             FrameScalars *fs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t *r = aos + (size_t)18 * i;
    const int slot = first + i;
    if ((int)r[0] != slot) atomicOr(&fs->error, ERR_BAD_ID);
    const int si = slot_index(P, slot);
    if (si < 0) return;                 // a slot another rank owns: its record is that rank's business
    // a live particle sits inside the box (set_pos_t wraps every position, app.cu:117-158);
    // the pair arithmetic is validated for in-box distances only
    if ((int)r[1] >= 0) {
        const float x = __uint_as_float(r[9]), y = __uint_as_float(r[10]), z = __uint_as_float(r[11]);
        bool bad = (int)r[1] >= P.num_cells_global;
        if (!bad) {
            // ... and inside the cell it claims (set_pos_t derives the cell from the position, app.cu:126-157;
            // the two-pass collision stage relies on it).  A wrapped position is rounded to float after the
            // cell was fixed, so allow it a sliver beyond the faces.  A coordinate that is no number (a child born
            // with the direction (0, 0, 0), a step later) belongs to the index the reference's conversion gives it:
            // INT_MIN, walked into the grid by the wrap loop (see k_apply).
            const int c = (int)r[1], G = P.G, i3 = c / (G * G), i1 = (c - i3 * G * G) / G, i2 = c - i3 * G * G - i1 * G;
            const float cs = (float)P.cell_size, tol = 1e-4f * cs, h = (float)(G / 2);
            int lost = (int)0x80000000;
            for (int guard = 0; guard < 4 && !(lost >= 0 && lost < G); guard++) lost = (lost + G) % G;
            auto axis_ok = [&](float v, float sign, int idx) {
                if ((__float_as_uint(v) & 0x7f800000u) == 0x7f800000u) return idx == lost;
                if (!(fabsf(v) <= half_box)) return false;
                const float u = (sign * v / cs + h - (float)idx) * cs;                      // in [0, cs) inside
                return u >= -tol && u <= cs + tol;
            };
            bad = !(axis_ok(x, 1.0f, i2) && axis_ok(y, -1.0f, i1) && axis_ok(z, -1.0f, i3));
        }
        if (bad) atomicOr(&fs->error, ERR_BAD_POS);
    }
    cell[si] = (int)r[1];
    if ((int)r[1] < 0) {
        // a free slot holds a reset record (reset_particle, app.cu:239-264), whatever the caller sent:
        // snapshot_restore relies on free slots being all-zero
        pflags[si] = 0;
        pos4[si] = vel4[si] = acc4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    pflags[si] = ((r[5] >> 8) & 0xffu) ? 1 : 0;
    pos4[si] = make_float4(__uint_as_float(r[9]), __uint_as_float(r[10]), __uint_as_float(r[11]), __uint_as_float(r[6]));
    vel4[si] = make_float4(__uint_as_float(r[12]), __uint_as_float(r[13]), __uint_as_float(r[14]), __uint_as_float(r[7]));
    acc4[si] = make_float4(__uint_as_float(r[15]), __uint_as_float(r[16]), __uint_as_float(r[17]), __uint_as_float(r[8]));
}

// slots this rank does not own come out as free records (reset_particle, app.cu:239-256)
__global__ void k_pack_aos(DevParams P, uint32_t *__restrict__ aos, int first, int count,
                           const float4 *pos4, const float4 *vel4, const float4 *acc4, const int *cell,
                           const uint8_t *pflags, const CellInfo *celltab)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint32_t *r = aos + (size_t)18 * i;
    const int slot = first + i;
    const int si = slot_index(P, slot);
    const int c = si >= 0 ? cell[si] : -1;
    CellInfo ci = {-1, -1, -1, 0};
    if (c >= 0 && c < P.num_cells_global) ci = celltab[c];
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 p = si >= 0 ? pos4[si] : zero, v = si >= 0 ? vel4[si] : zero, a = si >= 0 ? acc4[si] : zero;
    r[0] = (uint32_t)slot; r[1] = (uint32_t)c; r[2] = (uint32_t)ci.chunk;
    r[3] = (uint32_t)ci.seg_type; r[4] = (uint32_t)ci.seg_tid;
    r[5] = (si >= 0 && pflags[si]) ? 0x100u : 0u;  // seg_fault is never set between stages
    r[6] = __float_as_uint(p.w); r[7] = __float_as_uint(v.w); r[8] = __float_as_uint(a.w);
    r[9] = __float_as_uint(p.x); r[10] = __float_as_uint(p.y); r[11] = __float_as_uint(p.z);
    r[12] = __float_as_uint(v.x); r[13] = __float_as_uint(v.y); r[14] = __float_as_uint(v.z);
    r[15] = __float_as_uint(a.x); r[16] = __float_as_uint(a.y); r[17] = __float_as_uint(a.z);
}

// fill stage: drop freshly created particles into the slots the host dequeued
__global__ void k_place(DevParams P, int n, const int *__restrict__ ids, const float4 *__restrict__ p,
                        const float4 *__restrict__ v, const float4 *__restrict__ a,
                        const int *__restrict__ cells, float4 *pos4, float4 *vel4, float4 *acc4,
                        int *cell, uint8_t *pflags)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int si = slot_index(P, ids[i]);
    if (si < 0) return;
    pos4[si] = p[i]; vel4[si] = v[i]; acc4[si] = a[i];
    cell[si] = cells[i]; pflags[si] = 0;
}

// snapshot_restore: a slot that is free both now and in the snapshot holds the same
// (all-zero) record in both, so only slots occupied on either side are copied; the queue array
// (one word per slot) and the QUEUE_INFO records ride along in the same launch
__global__ void k_restore(int n, const float4 *__restrict__ s_pos, const float4 *__restrict__ s_vel,
                          const float4 *__restrict__ s_acc, const int *__restrict__ s_cell,
                          const uint8_t *__restrict__ s_flags, const int *__restrict__ s_queue,
                          const int *__restrict__ s_qinfo, int qinfo_words, int step, StepState *st,
                          float4 *pos4, float4 *vel4, float4 *acc4, int *cell, uint8_t *pflags, int *queue, int *qinfo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { st->step = step; st->pending = 0; }          // the step the snapshot was taken at
    if (i < qinfo_words) qinfo[i] = s_qinfo[i];
    if (i >= n) return;
    queue[i] = s_queue[i];
    const int cs = s_cell[i];
    if (cs < 0 && cell[i] < 0) return;
    pos4[i] = s_pos[i]; vel4[i] = s_vel[i]; acc4[i] = s_acc[i];
    cell[i] = cs; pflags[i] = s_flags[i];
}

hipError_t launch_restore(hipStream_t st, int n, const void *s_pos, const void *s_vel, const void *s_acc,
                          const void *s_cell, const void *s_flags, const void *s_queue, const void *s_qinfo, int qinfo_words,
                          int step, const DeviceState &d)
{
    const int threads = std::max(std::max(n, qinfo_words), 1);
    k_restore<<<(threads + 255) / 256, 256, 0, st>>>(n, (const float4 *)s_pos, (const float4 *)s_vel, (const float4 *)s_acc,
                                                    (const int *)s_cell, (const uint8_t *)s_flags, (const int *)s_queue,
                                                    (const int *)s_qinfo, qinfo_words, step, d.st, d.pos4, d.vel4, d.acc4,
                                                    d.cell, d.pflags, d.queue, (int *)d.qinfo);
    return hipGetLastError();
}

__global__ void k_fill_int(int *p, int v, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

__global__ void k_init_tdata(DevParams P, uint32_t *tdata)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.slots_total) return;
    uint32_t *r = tdata + (size_t)6 * i;  // T_DATA_TYPE: id x y z w age (ps.cpp:743-748)
    r[0] = (uint32_t)slot_of_index(P, i); r[1] = r[2] = r[3] = r[4] = r[5] = 0u;
}

// ------------------------------------------------------------------ grid build
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

// Same two kernels with a workgroup-private histogram in LDS: a workgroup owns SLOTS_PER_WG
// consecutive slots, which by the container's construction belong to one or two segments,
// i.e. a handful of cells a few grid planes apart, so almost all atomics stay in LDS and only
// the touched bins go to memory.  The LDS histogram is a WINDOW of LDS_CELLS cells starting at
// the smallest cell the workgroup meets (grids of up to LDS_CELLS cells: the whole grid); the
// rare slot outside it (a workgroup straddling two distant segments) goes to memory directly.
constexpr int LDS_CELLS = 8192;
constexpr int SLOTS_PER_WG = 4096;

// first cell of the workgroup's window; s_min: one int of LDS
__device__ __forceinline__ int hist_window(const DevParams &P, const int (&mine)[SLOTS_PER_WG / 1024], int *s_min)
{
    if (P.n_own_cells <= LDS_CELLS) return 0;
    if (threadIdx.x == 0) *s_min = 0x7fffffff;
    __syncthreads();
    int m = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WG / 1024; i++) if (mine[i] >= 0) m = min(m, mine[i]);
    for (int d = 32; d > 0; d >>= 1) m = min(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0 && m != 0x7fffffff) atomicMin(s_min, m);
    __syncthreads();
    return *s_min;
}

__global__ __launch_bounds__(1024) void k_hist_lds(DevParams P, const int *__restrict__ cell, int *__restrict__ cell_count,
                                                    FrameScalars *fs)
{
    __shared__ int h[LDS_CELLS];
    __shared__ int s_min;
    const int tid = threadIdx.x, base = blockIdx.x * SLOTS_PER_WG, ncell = P.n_own_cells;
    int mine[SLOTS_PER_WG / 1024];
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WG / 1024; i++) {
        const int si = base + i * 1024 + tid;
        mine[i] = si < P.slots_total ? own_local_cell(P, cell[si], fs) : -1;
    }
    const int w0 = hist_window(P, mine, &s_min);
    if (w0 == 0x7fffffff) return;                              // nothing alive in these slots
    const int span = min(ncell - w0, LDS_CELLS);
    for (int c = tid; c < span; c += 1024) h[c] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WG / 1024; i++) {
        const int c = mine[i];
        if (c < 0) continue;
        if (c - w0 < LDS_CELLS) atomicAdd(&h[c - w0], 1); else atomicAdd(&cell_count[c], 1);
    }
    __syncthreads();
    for (int c = tid; c < span; c += 1024) {
        const int v = h[c];
        if (v) atomicAdd(&cell_count[w0 + c], v);
    }
}

__device__ __forceinline__ void chunk_cap_block(const DevParams &P, int ch, const int *__restrict__ chunk_count,
                                                const int *__restrict__ cell_arr, const CellInfo *__restrict__ celltab,
                                                const int2 *__restrict__ chunk_segs, uint8_t *__restrict__ chunk_skip,
                                                const int *__restrict__ before4);

// Workgroups [0, nwg): the scatter.  Workgroups [nwg, nwg + num_chunks), one GPU only: the chunk
// lists' capacity rule for chunk blockIdx.x - nwg (chunk_cap_block; idle unless the chunk is over).
// The scatter also writes the T_DATA snapshot rows (id, x, y, z, w, age; ps.cpp:1495-1500): here a
// workgroup's live slots are consecutive, so are their 24-byte rows, and k_sort_cells gathers a
// particle's snapshot from its row -- one scattered read instead of two (pos4, vel4) and no scattered
// 24-byte row writes (which cost that kernel half again what it stored).
__global__ __launch_bounds__(1024) void k_scatter_lds(DevParams P, int nwg, const int *__restrict__ cell, int *__restrict__ cursor,
                                                       int *__restrict__ sorted_id, const int *__restrict__ chunk_count,
                                                       const CellInfo *__restrict__ celltab, const int2 *__restrict__ chunk_segs,
                                                       uint8_t *__restrict__ chunk_skip,
                                                       const float4 *__restrict__ pos4, const float4 *__restrict__ vel4,
                                                       uint32_t *__restrict__ tdata)
{
    if ((int)blockIdx.x >= nwg) { chunk_cap_block(P, (int)blockIdx.x - nwg, chunk_count, cell, celltab, chunk_segs, chunk_skip, nullptr); return; }
    __shared__ int h[LDS_CELLS];
    __shared__ int s_min;
    const int tid = threadIdx.x, base = blockIdx.x * SLOTS_PER_WG, ncell = P.n_own_cells;
    int mine[SLOTS_PER_WG / 1024];
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WG / 1024; i++) {
        const int si = base + i * 1024 + tid;
        int c = -1;
        if (si < P.slots_total) {
            c = cell[si];
            c = (c < 0 || c >= P.num_cells_global) ? -1 : c - P.reg_first[0] * P.G * P.G;
            if (c < 0 || c >= ncell) c = -1;          // foreign cells were flagged by the histogram pass
        }
        mine[i] = c;
    }
    const int w0 = hist_window(P, mine, &s_min);
    if (w0 == 0x7fffffff) return;
    const int span = min(ncell - w0, LDS_CELLS);
    for (int c = tid; c < span; c += 1024) h[c] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WG / 1024; i++)
        if (mine[i] >= 0 && mine[i] - w0 < LDS_CELLS) atomicAdd(&h[mine[i] - w0], 1);
    __syncthreads();
    for (int c = tid; c < span; c += 1024) {      // reserve this workgroup's run in each touched cell
        const int v = h[c];
        if (v) h[c] = atomicAdd(&cursor[w0 + c], v);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SLOTS_PER_WG / 1024; i++)
        if (mine[i] >= 0) {
            const int c = mine[i];
            const int pos = c - w0 < LDS_CELLS ? atomicAdd(&h[c - w0], 1) : atomicAdd(&cursor[c], 1);
            const int si = base + i * 1024 + tid, id = slot_of_index(P, si);
            sorted_id[pos] = id;
            const float4 p = pos4[si];
            const float age = vel4[si].w;
            uint2 *t = reinterpret_cast<uint2 *>(tdata + (size_t)6 * si);
            t[0] = make_uint2((uint32_t)id, __float_as_uint(p.x));
            t[1] = make_uint2(__float_as_uint(p.y), __float_as_uint(p.z));
            t[2] = make_uint2(__float_as_uint(p.w), __float_as_uint(age));
        }
}

// One workgroup: exclusive prefix of the own cells' counts, the scatter cursors, the chunk
// totals and hostGridMax (ps.cpp:1504-1516: maxima are of stored entries, so capped).
__global__ __launch_bounds__(1024) void k_scan(DevParams P, const int *__restrict__ cell_count,
                                                int *__restrict__ cell_start, int *__restrict__ cursor,
                                                int *__restrict__ task_start, int *__restrict__ task_list,
                                                int *__restrict__ chunk_count,
                                                const CellInfo *__restrict__ celltab, int *__restrict__ status_out, FrameScalars *fs)
{
    // Two prefix sums at once, packed in 64 bits: particles per cell (low word) and
    // 64-particle pair-kernel tasks per cell (high word; only the cells this rank computes).
    // Each thread owns a contiguous run of cells, so the whole scan needs one pass and two barriers.
    constexpr int LDS_CHUNKS = 4096;
    __shared__ long long wave_tot[16];
    __shared__ int maxcell_s, maxraw_s;
    __shared__ int chunk_s[LDS_CHUNKS];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool chunks_in_lds = P.num_chunks <= LDS_CHUNKS;
    const int ncell = P.n_own_cells, cell_off = P.reg_first[0] * P.G * P.G;
    if (tid == 0) { maxcell_s = 0; maxraw_s = 0; }
    if (chunks_in_lds) for (int ch = tid; ch < P.num_chunks; ch += 1024) chunk_s[ch] = 0;
    __syncthreads();
    const int per = (ncell + 1023) / 1024;
    const int c0 = min(ncell, tid * per), c1 = min(ncell, c0 + per);
    auto word = [&](int c, int v) {
        const bool computed = c >= P.own_comp0 && c < P.own_comp1;
        return ((long long)(computed ? (min(v, P.max_per_cell) + 63) >> 6 : 0) << 32) | (long long)v;
    };
    long long mine = 0;
    int mymax = 0, myraw = 0;
    // (a thread's first KEEP counts stay in registers for the second pass, their loads -- and then the cell table's --
    // go out as one batch each: the kernel is one workgroup's chain of round trips, nothing else)
    constexpr int KEEP = 8;
    int kept[KEEP];
    CellInfo kci[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; k++) kept[k] = (c0 + k < c1) ? cell_count[c0 + k] : 0;
#pragma unroll
    for (int k = 0; k < KEEP; k++) if (kept[k] > 0) kci[k] = celltab[c0 + k + cell_off];
    auto census = [&](int c, int v, const CellInfo &ci) {
        if (chunks_in_lds) atomicAdd(&chunk_s[ci.chunk], v);
        else atomicAdd(&chunk_count[ci.chunk], v);
        // slab: a particle lives in the segment of its cell, so this is also the census of the chunk's
        // particles per segment type held here, for the other ranks (status record, zeroed with the frame)
        if (P.world > 1) atomicAdd(&status_out[STATUS_CHUNK_OFF + 4 * ci.chunk + (ci.seg_type == 1 ? 0 : ci.seg_type == 2 ? 1 : ci.seg_type == 4 ? 2 : 3)], v);
    };
#pragma unroll
    for (int k = 0; k < KEEP; k++) {
        const int c = c0 + k, v = kept[k];
        if (c >= c1) continue;
        mymax = max(mymax, min(v, P.max_per_cell)); myraw = max(myraw, v);
        mine += word(c, v);
        if (v > 0) census(c, v, kci[k]);
    }
    for (int c = c0 + KEEP; c < c1; c++) {
        const int v = cell_count[c];
        mymax = max(mymax, min(v, P.max_per_cell)); myraw = max(myraw, v);
        mine += word(c, v);
        if (v > 0) census(c, v, celltab[c + cell_off]);
    }
    long long incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wave_tot[wv] = incl;
    if (mymax) atomicMax(&maxcell_s, mymax);
    if (myraw > P.max_per_cell || myraw > 1024) atomicMax(&maxraw_s, myraw);       // (only a crowded cell bothers)
    __syncthreads();
    long long run = incl - mine, total = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) run += wave_tot[k]; total += wave_tot[k]; }
    auto place = [&](int c, int v) {
        const int excl = (int)(run & 0xffffffffll);
        cell_start[c] = excl;
        cursor[c] = excl;
        const int t0 = (int)(run >> 32);
        task_start[c] = t0;
        const long long w = word(c, v);
        // the work list of the one-pass pair stage: one entry per non-empty (cell, 64-particle slice) of the own
        // computed cells (the lent ones are appended when their snapshot has arrived, k_halo_prefix_in); the
        // two-pass stage makes its own list of the particles that need a force (k_plan_force)
        if (!P.two_pass) for (int sl = 0; sl < (int)(w >> 32); sl++) task_list[t0 + sl] = c * P.slices + sl;
        run += w;
    };
#pragma unroll
    for (int k = 0; k < KEEP; k++) if (c0 + k < c1) place(c0 + k, kept[k]);
    for (int c = c0 + KEEP; c < c1; c++) place(c, cell_count[c]);
    if (tid == 0) {
        cell_start[ncell] = (int)(total & 0xffffffffll);     // the gap cell after region 0: end of the own bodies
        task_start[ncell] = (int)(total >> 32);
        fs->live = (int)(total & 0xffffffffll);
        fs->n_tasks = (int)(total >> 32);
        fs->gridmax[1] = maxcell_s;
        fs->max_cell_raw = max(maxraw_s, maxcell_s);
    }
    // chunk totals: complete after the barrier above (every thread added its cells before it)
    int cm = 0;
    bool over = false;
    if (chunks_in_lds) {
        for (int ch = tid; ch < P.num_chunks; ch += 1024) {
            const int v = chunk_s[ch];
            chunk_count[ch] = v;
            over |= v > P.max_per_chunk;
            cm = max(cm, min(v, P.max_per_chunk));
        }
    } else {
        __threadfence();
        __syncthreads();
        for (int ch = tid; ch < P.num_chunks; ch += 1024) { const int v = chunk_count[ch]; over |= v > P.max_per_chunk; cm = max(cm, min(v, P.max_per_chunk)); }
    }
    if (cm > 0) atomicMax(&fs->gridmax[0], cm);
    // The reference stores only MAX_PARTICLES_PER_CHUNK ids per chunk and calc_forces walks the
    // stored list (ps.cpp:1502-1508): past that (only possible while cells overflow, the count
    // includes the killed) the tail of the chunk's slot-ordered list is not processed that step.
    // One GPU: chunk_cap_block (riding on the scatter launch) marks that tail and k_apply leaves it
    // alone.  A slab holds only part of a chunk's segments: whether a chunk is over, and where this
    // rank's particles stand in its list, is settled when the status records of all ranks are in
    // (k_status_merge, before k_apply); the count above is this rank's part only.
    if (over && P.world == 1) fs->chunk_over = 1;
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

// One workgroup per own cell.  The scatter left the cell's ids in arrival order; the
// reference's list is in slot order (build_grid walks slots 0..CONTAINER_SIZE-1), so
// rank each id among the cell's ids.  Then gather the snapshot the pair kernel reads
// (T_DATA_TYPE role, ps.cpp:1495-1500) in that order: x,y,z and the mass, the mass
// zeroed for "kids" because bodyBodyInteraction ignores them (app_common.cu:240-243;
// adding r*0 = +-0 leaves an fp32 sum that started at +0 bit-identical).
// Ids ranked at or past the list capacity are the ones the reference kills
// (ps.cpp:1517-1526): their sorted_id entry becomes -1 and the slot is reset.
// CAP: the ids an instance ranks in LDS.  Two instances are launched back to back: CAP = 1024 (8 KB of
// LDS: the workgroups of sixteen cells per CU in flight -- the kernel is a chain of global round trips
// per cell, not a stream) serves the cells with up to 1024 ids, CAP = SORT_MAX (33 KB: four cells per
// CU) the fuller ones; a workgroup leaves at once where the cell is the other instance's.
template <int CAP>
__device__ __forceinline__ void sort_cell(const DevParams &P, const int c, const int *__restrict__ cell_start,
                                                     int *__restrict__ sorted_id,
                                                     float4 *pos4, float4 *vel4, float4 *acc4,
                                                     int *cell_arr, uint8_t *pflags,
                                                     float *__restrict__ snap_soa,
                                                     float *__restrict__ snap_age,
                                                     const uint32_t *__restrict__ tdata, int *__restrict__ rank_of_slot,
                                                     uint64_t *op_keys, int *op_args, int ops_cap,
                                                     int *__restrict__ halo_count, float *__restrict__ halo_f,
                                                     int *__restrict__ halo_id, int *__restrict__ snap_cid,
                                                     int *__restrict__ status_out, FrameScalars *fs, DevCounters *ctr)
{
    constexpr int SMALL = 1024;
    static_assert(CAP == SMALL || CAP == SORT_MAX, "two instances: ordinary cells, crowded cells");
    __shared__ __attribute__((aligned(16))) int ids[CAP + 4];
    __shared__ int ordered[CAP];
    __shared__ int s_halo[27], s_halo_base[27];    // bodies this cell lists in each neighbour's halo
    constexpr int BITMAP_WORDS = 512;              // the ids' span the bitmap ranking covers: 16384 slots
    __shared__ unsigned bitmap[BITMAP_WORDS];
    __shared__ int s_lo, s_hi, s_wt[4];
    const int tid = threadIdx.x;
    if (tid < 27) s_halo[tid] = 0;
    if (tid == 0) { s_lo = 0x7fffffff; s_hi = -1; }
    const int start = cell_start[c];
    int n = cell_start[c + 1] - start;
    if (n == 0 || (CAP == SMALL ? n > SMALL : n <= SMALL)) return;       // (empty, or the other instance's)
    int ci1, ci2, ci3;
    cell_coords(P, c, ci1, ci2, ci3);
    // A cell may hold more ids than fit the LDS ranking (its segment's capacity is the bound: a
    // dense clump in one cell of an 8-cell segment reaches 8 x 514 = 4112).  All but the
    // MAX_PARTICLES_PER_CELL lowest are killed anyway, so such a cell first finds that many lowest
    // ids -- bisection on the id value, counting in global memory -- ranks those in LDS as usual
    // and treats the rest as the overflow it is (rare: slow is fine, wrong is not).
    if (n > SORT_MAX && P.max_per_cell > SORT_MAX) {
        // (a list capacity above what the LDS ranking holds -- N = 2^24 in 16^3 cells -- AND a cell that
        // full: the kept ids alone do not fit; refused as before)
        if (tid == 0) atomicOr(&fs->error, ERR_CELL_TOO_BIG);
        n = SORT_MAX;
    }
    const int n_all = n;
    int big_limit = 0x7fffffff;                      // ids >= big_limit are past the list capacity (big cells only)
    if (n > SORT_MAX) {
        __shared__ int s_count;
        int lo = 0, hi = 0x7fffffff;                 // smallest t with #(id < t) >= max_per_cell
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if (tid == 0) s_count = 0;
            __syncthreads();
            int mine = 0;
            for (int e = tid; e < n_all; e += 256) mine += sorted_id[start + e] < mid ? 1 : 0;
            atomicAdd(&s_count, mine);
            __syncthreads();
            const int cnt = s_count;
            __syncthreads();
            if (cnt >= P.max_per_cell) hi = mid; else lo = mid + 1;
        }
        big_limit = lo;
        // gather the kept ids (exactly max_per_cell of them: ids are distinct) to the front of the LDS list
        if (tid == 0) s_count = 0;
        __syncthreads();
        for (int e = tid; e < n_all; e += 256) {
            const int id = sorted_id[start + e];
            if (id < big_limit) ids[atomicAdd(&s_count, 1)] = id;
        }
        __syncthreads();
        n = s_count;                                 // == max_per_cell
        __syncthreads();
        // the overflow: same treatment as the ranked tail below, in any order (the frees are keyed by id)
        for (int e = tid; e < n_all; e += 256) {
            const int id = sorted_id[start + e];
            if (id < big_limit) continue;
            const int si = slot_index(P, id);                      // (its snapshot row was written by the scatter pass)
            cell_arr[si] = P.world > 1 ? -2 - cell_arr[si] : -1; pflags[si] = 0;       // (slab: the chunk-capacity walk still needs the cell, see chunk_cap_block)
            pos4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            vel4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            acc4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            atomicAdd(&(ctr + (blockIdx.x % COUNTER_COPIES))->cell_overflow_kills, 1ull);
            if (owns_record(P, 0)) {
                const int k = atomicAdd(&fs->n_ops, 1);
                if (k < ops_cap) { op_keys[k] = ((uint64_t)(uint32_t)id << 2) | 2ull; op_args[k] = id; }
                else atomicOr(&fs->error, ERR_OPS_OVERFLOW);
            } else {
                const int k = atomicAdd(&status_out[0], 1);
                if (k < STATUS_KILL_CAP) status_out[MSG_HEADER_WORDS + k] = id;
                else atomicOr(&fs->error, ERR_REMOTE_RECORD0);
            }
        }
        __syncthreads();                             // (all reads of the arrival-order list are done)
        for (int e = n + tid; e < n_all; e += 256) sorted_id[start + e] = -1;
    } else {
        for (int e = tid; e < n; e += 256) ids[e] = sorted_id[start + e];
    }
    __syncthreads();
    // The ids in ascending order.  A cell's particles live in the slots of one segment, so the ids span a few
    // thousand values: a bitmap of the span in LDS (one atomicOr per id), a prefix of the words' population
    // counts, and every thread writes out the ids of its two words.  (Before: every id counted the smaller ones
    // among all of them, n/4 16-byte broadcast reads per thread -- at 256 ids per cell the LDS pipe's
    // 14 us of the kernel.)  Ids spread wider than the bitmap holds are ranked by counting as before.
    int lo = 0x7fffffff, hi = -1;
    for (int e = tid; e < n; e += 256) { const int v = ids[e]; lo = min(lo, v); hi = max(hi, v); }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { lo = min(lo, __shfl_xor(lo, sft)); hi = max(hi, __shfl_xor(hi, sft)); }
    if ((tid & 63) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    const int id_base = s_lo, span = s_hi - id_base + 1;
    if (span <= BITMAP_WORDS * 32) {
        const int words = (span + 31) >> 5;
        for (int w = tid; w < words; w += 256) bitmap[w] = 0;
        __syncthreads();
        for (int e = tid; e < n; e += 256) { const int b = ids[e] - id_base; atomicOr(&bitmap[b >> 5], 1u << (b & 31)); }
        __syncthreads();
        constexpr int WPT = BITMAP_WORDS / 256;                 // words per thread
        unsigned w[WPT];
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < WPT; i++) { w[i] = (WPT * tid + i < words) ? bitmap[WPT * tid + i] : 0u; cnt += __popc(w[i]); }
        const int incl = wave_incl_scan(cnt);
        if ((tid & 63) == 63) s_wt[tid >> 6] = incl;
        __syncthreads();
        int pos = incl - cnt;
        for (int k = 0; k < (tid >> 6); k++) pos += s_wt[k];
#pragma unroll
        for (int i = 0; i < WPT; i++)
            for (unsigned m = w[i]; m; m &= m - 1) ordered[pos++] = id_base + (WPT * tid + i) * 32 + (__ffs(m) - 1);
    } else {
        // pad to a multiple of 4 with INT_MAX so the ranking reads whole 16-byte LDS words
        for (int e = n + tid; e < ((n + 3) & ~3); e += 256) ids[e] = 0x7fffffff;
        __syncthreads();
        for (int e = tid; e < n; e += 256) {
            const int mine = ids[e];
            int rank = 0;
            const int4 *v = reinterpret_cast<const int4 *>(ids);
#pragma unroll 4
            for (int j = 0; j < (n + 3) / 4; j++) {
                const int4 q = v[j];
                rank += (q.x < mine) + (q.y < mine) + (q.z < mine) + (q.w < mine);
            }
            ordered[rank] = mine;
        }
    }
    __syncthreads();
    // (the small instance keeps what the halo lists need -- position, collision id, face bits -- in registers
    // instead of reading the rows back and redoing the three divisions)
    constexpr int KR = CAP == SMALL ? SMALL / 256 : 1;
    float hx[KR], hy[KR], hz[KR];
    int hid[KR], hm3[KR];
#pragma unroll
    for (int k = 0; k < KR; k++) hm3[k] = 0;
    bool wild_any = false;
    auto row = [&](int e, int k) {
        const int id = ordered[e], si = slot_index(P, id);
        // the particle's T_DATA row (written by the scatter pass for every live slot, before the overflow
        // check as in ps.cpp:1495-1500): x, y, z, w, age in one 24-byte read
        const uint2 *t = reinterpret_cast<const uint2 *>(tdata + (size_t)6 * si);
        const uint2 t0 = t[0], t1 = t[1], t2 = t[2];
        float4 p = make_float4(__uint_as_float(t0.y), __uint_as_float(t1.x), __uint_as_float(t1.y), __uint_as_float(t2.x));
        const float age = __uint_as_float(t2.y);
        // A kid is skipped by the reference's force loop and never collides (app_common.cu:240-243, 284-287);
        // here it stays in the lists with mass 0, so that r * 0 = +-0 leaves every sum as it was -- which needs r
        // to be a number.  A child born with the direction (0, 0, 0) has a velocity and, a step later, a position
        // that is not one (0/0, ps.cpp:1306-1333): in the snapshot a kid's position is the origin (its T_DATA row
        // and its own state keep what the reference has).
        if (age < P.kid_thr) p.x = p.y = p.z = 0.0f;
        if (e < P.max_per_cell) {
            sorted_id[start + e] = id;
            rank_of_slot[si] = start + e;
            const float w_eff = (age < P.kid_thr) ? 0.0f : (P.force_sign < 0.f ? -p.w : p.w);
            {   // the same four values as separate arrays: what the pair kernel streams
                const size_t cap = (size_t)P.sorted_cap;
                snap_soa[start + e] = p.x; snap_soa[cap + start + e] = p.y;
                snap_soa[2 * cap + start + e] = p.z; snap_soa[3 * cap + start + e] = w_eff;
            }
            snap_age[start + e] = age;
            // collision id: the slot id, or -1 for a body that can never collide (kid, over age)
            const bool collides = !(age < P.kid_thr) && !(age > P.life_thr);
            snap_cid[start + e] = collides ? id : -1;
            // (small instance: a candidate whose position is no number -- HALO_ALL -- is left to a pass of its own
            // below, so that the loop every body takes knows nothing of it: with the 26-neighbour case in here
            // the kernel took 10 us more, measured)
            const int m3 = !(halo_count && collides) ? 0 : CAP == SMALL ? halo_dirs_of_numbers(P, ci1, ci2, ci3, p.x, p.y, p.z)
                                                                        : halo_dirs(P, ci1, ci2, ci3, p.x, p.y, p.z);
            if (CAP == SMALL && halo_count && collides && !finite3(p.x, p.y, p.z)) wild_any = true;
            if (m3) {
                if (CAP == SMALL) {
#pragma unroll
                    for (int m = 1; m < 8; m++) { const int dir = halo_dir_of_subset(m3, m); if (dir >= 0) atomicAdd(&s_halo[dir], 1); }
                    hx[k] = p.x; hy[k] = p.y; hz[k] = p.z; hid[k] = id; hm3[k] = m3;
                } else
                    for_each_halo_dir(m3, [&](int dir) { atomicAdd(&s_halo[dir], 1); });
            }
        } else {
            sorted_id[start + e] = -1;
            cell_arr[si] = P.world > 1 ? -2 - cell_arr[si] : -1; pflags[si] = 0;       // (slab: the chunk-capacity walk still needs the cell, see chunk_cap_block)
            pos4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            vel4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            acc4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            atomicAdd(&(ctr + (blockIdx.x % COUNTER_COPIES))->cell_overflow_kills, 1ull);
            // freed with the already-reset segment (-1,-1): queue record 0 (ps.cpp:1523-1526).  On a
            // slab that does not hold that queue the slot id travels to its owner in the status message.
            if (owns_record(P, 0)) {
                const int k = atomicAdd(&fs->n_ops, 1);
                if (k < ops_cap) { op_keys[k] = ((uint64_t)(uint32_t)id << 2) | 2ull; op_args[k] = id; }
                else atomicOr(&fs->error, ERR_OPS_OVERFLOW);
            } else {
                const int k = atomicAdd(&status_out[0], 1);
                if (k < STATUS_KILL_CAP) status_out[MSG_HEADER_WORDS + k] = id;
                else atomicOr(&fs->error, ERR_REMOTE_RECORD0);
            }
        }
    };
    if (CAP == SMALL) {
#pragma unroll
        for (int k = 0; k < KR; k++) { const int e = tid + 256 * k; if (e < n) row(e, k); }
    } else {
        for (int e = tid; e < n; e += 256) row(e, 0);
    }
    if (!halo_count) return;
    const bool wild_cell = __syncthreads_or(wild_any);   // the snapshot rows of this cell are in memory, the directions counted
    if (CAP != SMALL) {
        list_in_neighbour_halos(P, c, start, min(n, P.max_per_cell), SnapSoa{snap_soa, (size_t)P.sorted_cap}, snap_cid, halo_count, halo_f, halo_id, s_halo, s_halo_base, true);
        return;
    }
    // the candidates whose position is no number: listed with all 26 neighbours (read back from the rows)
    auto for_each_wild = [&](auto fn) {
        const size_t cap = (size_t)P.sorted_cap;
        for (int e = tid; e < min(n, P.max_per_cell); e += 256) {
            const int id = snap_cid[start + e];
            const float x = snap_soa[start + e], y = snap_soa[cap + start + e], z = snap_soa[2 * cap + start + e];
            if (id >= 0 && !finite3(x, y, z)) fn(id, x, y, z);
        }
    };
    if (wild_cell) {
        for_each_wild([&](int, float, float, float) { for (int dir = 0; dir < 27; dir++) if (dir != 13) atomicAdd(&s_halo[dir], 1); });
        __syncthreads();
    }
    // room in each neighbour's list with one global atomic per direction, then the bodies (as list_in_neighbour_halos)
    if (tid < 27) {
        int base = -1;
        const int cnt = s_halo[tid];
        if (cnt > 0) {
            const int nc = halo_neighbour(P, ci1, ci2, ci3, tid);
            if (nc >= 0) base = atomicAdd(&halo_count[nc], cnt);
        }
        s_halo_base[tid] = base;
        s_halo[tid] = 0;
    }
    __syncthreads();
    const size_t plane = (size_t)P.n_local_cells * HALO_CAP;
#pragma unroll
    for (int k = 0; k < KR; k++) {
        if (!hm3[k]) continue;
#pragma unroll
        for (int m = 1; m < 8; m++) {
            const int dir = halo_dir_of_subset(hm3[k], m);
            if (dir < 0 || s_halo_base[dir] < 0) continue;
            const int kk = s_halo_base[dir] + atomicAdd(&s_halo[dir], 1);
            if (kk < HALO_CAP) {
                const size_t at = (size_t)halo_neighbour(P, ci1, ci2, ci3, dir) * HALO_CAP + kk;
                halo_f[at] = hx[k]; halo_f[plane + at] = hy[k]; halo_f[2 * plane + at] = hz[k];
                halo_id[at] = hid[k];
            }
        }
    }
    if (wild_cell)
        for_each_wild([&](int id, float x, float y, float z) {
            for (int dir = 0; dir < 27; dir++) {
                if (dir == 13 || s_halo_base[dir] < 0) continue;
                const int kk = s_halo_base[dir] + atomicAdd(&s_halo[dir], 1);
                if (kk < HALO_CAP) {
                    const size_t at = (size_t)halo_neighbour(P, ci1, ci2, ci3, dir) * HALO_CAP + kk;
                    halo_f[at] = x; halo_f[plane + at] = y; halo_f[2 * plane + at] = z;
                    halo_id[at] = id;
                }
            }
        });
}

// The instance for ordinary cells runs one workgroup per cell; the one for crowded cells (more than 1024 ids: a
// collapsing cloud) is launched every step too, with a few workgroups that leave at once unless the frame has
// such a cell (max_cell_raw, from k_scan) and otherwise stride over the cells.
template <int CAP>
__global__ __launch_bounds__(256) void k_sort_cells(DevParams P, const int *__restrict__ cell_start, int *__restrict__ sorted_id,
                                                     float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                     float *__restrict__ snap_soa, float *__restrict__ snap_age,
                                                     const uint32_t *__restrict__ tdata, int *__restrict__ rank_of_slot,
                                                     uint64_t *op_keys, int *op_args, int ops_cap,
                                                     int *__restrict__ halo_count, float *__restrict__ halo_f,
                                                     int *__restrict__ halo_id, int *__restrict__ snap_cid,
                                                     int *__restrict__ status_out, FrameScalars *fs, DevCounters *ctr)
{
    if (CAP == 1024) {
        sort_cell<CAP>(P, (int)blockIdx.x, cell_start, sorted_id, pos4, vel4, acc4, cell_arr, pflags, snap_soa, snap_age, tdata, rank_of_slot,
                       op_keys, op_args, ops_cap, halo_count, halo_f, halo_id, snap_cid, status_out, fs, ctr);
        return;
    }
    if (fs->max_cell_raw <= 1024) return;
    for (int c = blockIdx.x; c < P.n_own_cells; c += gridDim.x) {
        sort_cell<CAP>(P, c, cell_start, sorted_id, pos4, vel4, acc4, cell_arr, pflags, snap_soa, snap_age, tdata, rank_of_slot,
                       op_keys, op_args, ops_cap, halo_count, halo_f, halo_id, snap_cid, status_out, fs, ctr);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ pair kernel
// Correctly rounded fp32 sqrt and reciprocal without the range/denormal scaffolding
// the compiler wraps around them: valid for normal inputs well inside the exponent
// range (the host only selects them when eps2^3 .. (3 (2L)^2 + eps2)^3 lies in
// [2^-60, 2^60]).  Each is one hardware estimate (v_rsq_f32 / v_rcp_f32, 1 ulp) plus
// one residual correction, and each is checked against the compiler's correctly
// rounded form over EVERY float of [2^-62, 2^62] by psamd_selftest_math
// (tests/test_gpu_math.py): zero mismatches.
__device__ __forceinline__ float sqrt_rn_short(float a)
{
    const float r = __builtin_amdgcn_rsqf(a);
    const float g = a * r;                      // ~sqrt(a)
    const float h = 0.5f * r;                   // ~1 / (2 sqrt(a))
    const float d = __builtin_fmaf(-g, g, a);   // exact residual
    return __builtin_fmaf(d, h, g);
}

__device__ __forceinline__ float rcp_rn_newton(float q)
{
    const float x = __builtin_amdgcn_rcpf(q);
    const float e = __builtin_fmaf(-q, x, 1.0f);
    return __builtin_fmaf(e, x, x);
}

// RN(1 / RN(sqrt(a))): the reference's 1.0f / sqrtf(a), two roundings.
__device__ __forceinline__ float inv_sqrt_selected(float six)
{
    return rcp_rn_newton(sqrt_rn_short(six));
}

// One transcendental instead of two: the reciprocal's Newton step starts from the rsq estimate
// itself (r ~ 1/sqrt(a) ~ 1/s).  That is RN(1/s) for every float of the range EXCEPT where s has
// an all-ones mantissa (1/s lies a hair above a rounding tie and the step lands on the tie: 124
// inputs in [2^-62, 2^62]); there the residual e is exactly 2^-24, which is what `tie` reports so
// that the caller can redo the group with inv_sqrt_selected.  Checked for every float of the
// range by psamd_selftest_math: no mismatch that is not reported.  v_rcp_f32 costs 3.3 issue
// slots on gfx950 (profiles/r1_microbench_valu_rates.txt), the compare one.
__device__ __forceinline__ float inv_sqrt_guarded(float a, bool &tie)
{
    const float r = __builtin_amdgcn_rsqf(a);
    const float g = a * r, h = 0.5f * r;
    const float s = __builtin_fmaf(__builtin_fmaf(-g, g, a), h, g);
    const float e = __builtin_fmaf(-s, r, 1.0f);
    tie = tie || e == 0x1p-24f;
    return __builtin_fmaf(e, r, r);
}

// A tempting shortcut that is NOT exact, kept only so the self test can show it: start
// the reciprocal's Newton step from the rsq estimate (2h ~ 1/g) instead of a second
// transcendental.  124 of the 1.04e9 floats in range come out one ulp off.
__device__ __forceinline__ float inv_sqrt_one_transcendental(float a)
{
    const float r = __builtin_amdgcn_rsqf(a);
    float g = a * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    const float q = __builtin_fmaf(__builtin_fmaf(-g, g, a), h, g);
    float x = h + h;
    for (int k = 0; k < 2; k++) x = __builtin_fmaf(__builtin_fmaf(-q, x, 1.0f), x, x);
    return x;
}

// bodyBodyInteraction, app_common.cu:236-267, for a snapshot body q = (x,y,z,w_eff).
__device__ __forceinline__ float pair_exact(float xi, float yi, float zi, const float4 q, double eps2,
                                            float &ax, float &ay, float &az)
{
    const float rx = q.x - xi, ry = q.y - yi, rz = q.z - zi;
    const float d2 = rx * rx + ry * ry + rz * rz;
    const float dsq = (float)((double)d2 + eps2);      // EPS2 is a double literal
    const float six = dsq * dsq * dsq;
    const float inv = 1.0f / sqrtf(six);               // correctly rounded sqrt, then divide
    const float s = q.w * inv;
    ax += rx * s; ay += ry * s; az += rz * s;
    return d2;
}

// Same physics with fused multiply-adds and the hardware reciprocal square root:
// differs from the reference in the last bits (PSAMD_FLAG_FAST_MATH).
__device__ __forceinline__ float pair_fast(float xi, float yi, float zi, const float4 q, float eps2,
                                           float &ax, float &ay, float &az)
{
    const float rx = q.x - xi, ry = q.y - yi, rz = q.z - zi;
    const float d2 = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
    const float dsq = d2 + eps2;
    const float rinv = __builtin_amdgcn_rsqf(dsq);
    const float s = q.w * (rinv * rinv * rinv);
    ax = fmaf(rx, s, ax); ay = fmaf(ry, s, ay); az = fmaf(rz, s, az);
    return d2;
}

// bodyBodyCollision, app_common.cu:269-301, evaluated exactly for the few pairs whose
// squared distance passes the gate.  0 none, 1 survive (higher id), 2 kill (lower id).
__device__ __forceinline__ int collide_exact(const DevParams &P, float d2, float age_i, int id_i,
                                             float age_j, int id_j)
{
    const float dist = sqrtf(d2);
    if ((double)dist > P.coll_radius || (double)age_i < P.kid_age || (double)age_j < P.kid_age) return 0;
    if ((double)age_i > P.life || (double)age_j > P.life) return 0;
    if (id_i > id_j) return 1;
    if (id_i < id_j) return 2;
    return 0;
}

// Lean exact pair arithmetic for k_pairs<1>.  The reference adds the double literal EPS2
// in double and rounds to float; from eps_f32_from upwards a plain fp32 add gives the same
// bits (checked for every such float when the context is created).  A wave takes the
// slow branch only when one of its lanes holds a pair closer than `slow_below` =
// max(eps_f32_from, collision gate): there EPS2 is added in double and the exact collision
// rule is evaluated for the pairs inside the gate, so the common path carries neither.
struct PairCtx {
    float xi, yi, zi, age_i;
    int id_i, gi;
    bool scan;
};

typedef float v2f __attribute__((ext_vector_type(2)));

// Two pairs per instruction slot: gfx950 has packed fp32 add/mul/fma, and the SoA tile hands
// (x_j, x_j+1) over in one aligned register pair, so nothing is shuffled between registers.
// Every packed operation rounds each half exactly like its scalar form.
__device__ __forceinline__ v2f inv_sqrt_selected2(v2f six)
{
    v2f r; r.x = __builtin_amdgcn_rsqf(six.x); r.y = __builtin_amdgcn_rsqf(six.y);
    const v2f g = six * r, h = 0.5f * r;
    const v2f s = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, g, six), h, g);   // sqrt_rn_short
    v2f x; x.x = __builtin_amdgcn_rcpf(s.x); x.y = __builtin_amdgcn_rcpf(s.y);
    const v2f one = {1.0f, 1.0f};
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-s, x, one), x, x);          // rcp_rn_newton
}

// inv_sqrt_guarded on two pairs
__device__ __forceinline__ v2f inv_sqrt_guarded2(v2f six, bool &tie)
{
    v2f r; r.x = __builtin_amdgcn_rsqf(six.x); r.y = __builtin_amdgcn_rsqf(six.y);
    const v2f g = six * r, h = 0.5f * r;
    const v2f s = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, g, six), h, g);
    const v2f one = {1.0f, 1.0f};
    const v2f e = __builtin_elementwise_fma(-s, r, one);
    tie = tie || e.x == 0x1p-24f || e.y == 0x1p-24f;
    return __builtin_elementwise_fma(e, r, r);
}

// NQ pairs in two stages, so that a caller can start fetching the next group's bodies
// between them: distances first (the only use of the positions), then everything else.
template <int NQ>
struct PairRows {
    v2f rx[NQ / 2], ry[NQ / 2], rz[NQ / 2], d[NQ / 2];
    float dm;                                   // smallest d of the group
};

// SOFTENED: d = fma chain started at eps2 (fast math); else the reference's unfused r.r
template <int NQ, bool SOFTENED>
__device__ __forceinline__ void pairs_dist(const PairCtx &c, const v2f (&qx)[NQ / 2], const v2f (&qy)[NQ / 2],
                                           const v2f (&qz)[NQ / 2], float eps2, PairRows<NQ> &r)
{
    const v2f xi = {c.xi, c.xi}, yi = {c.yi, c.yi}, zi = {c.zi, c.zi}, eps = {eps2, eps2};
    r.dm = 3.0e38f;
#pragma unroll
    for (int i = 0; i < NQ / 2; i++) {
        r.rx[i] = qx[i] - xi; r.ry[i] = qy[i] - yi; r.rz[i] = qz[i] - zi;
        if (SOFTENED)
            r.d[i] = __builtin_elementwise_fma(r.rz[i], r.rz[i], __builtin_elementwise_fma(r.ry[i], r.ry[i], __builtin_elementwise_fma(r.rx[i], r.rx[i], eps)));
        else
            r.d[i] = r.rx[i] * r.rx[i] + r.ry[i] * r.ry[i] + r.rz[i] * r.rz[i];
        r.dm = fminf(fminf(r.dm, r.d[i].x), r.d[i].y);
    }
}

// ONE_T: one transcendental per pair (inv_sqrt_guarded2) -- fewer issue slots, for passes that are
// throughput-bound (four or more waves per SIMD: -3.7 % on the N = 2^20 force pass); the two-
// transcendental form has the shorter dependency chain and wins where a SIMD holds one or two waves
// (a 1/8 slab's tile walk: 0.57 against 0.64 ms).
template <int NQ, bool ONE_T>
__device__ __forceinline__ void pairs_finish_exact(const DevParams &P, const PairCtx &c, const PairRows<NQ> &r,
                                                   const v2f (&qw)[NQ / 2], int gj0,
                                                   const float *__restrict__ snap_age,
                                                   const int *__restrict__ sorted_id,
                                                   float &ax, float &ay, float &az, int &flag)
{
    constexpr int H = NQ / 2;
    v2f e[H];
    // One-pass stage (c.scan; a compile-time false in the two-pass force pass): a distance that is not a number --
    // the particle's own position or a body's is not one -- passes the reference's collision test, but the
    // group's minimum does not see it (fminf drops it): such a group takes the branch with the exact rule too.
    bool wild = false;
    if (c.scan) {
        v2f t = r.d[0];
#pragma unroll
        for (int i = 1; i < H; i++) t = t + r.d[i];
        const float tt = t.x + t.y;
        wild = tt != tt;
    }
    if (__any(r.dm < P.slow_below) || __any(wild)) {
#pragma unroll
        for (int i = 0; i < H; i++) {
            e[i].x = (float)((double)r.d[i].x + P.eps2);
            e[i].y = (float)((double)r.d[i].y + P.eps2);
        }
        if (c.scan && (wild || !(r.dm > P.coll_d2_gate))) {
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const float di = (i & 1) ? r.d[i >> 1].y : r.d[i >> 1].x;
                if (!(di > P.coll_d2_gate) && gj0 + i != c.gi)
                    flag = max(flag, collide_exact(P, di, c.age_i, c.id_i, snap_age[gj0 + i], sorted_id[gj0 + i]));
            }
        }
    } else {
        const v2f eps = {P.eps2f, P.eps2f};
#pragma unroll
        for (int i = 0; i < H; i++) e[i] = r.d[i] + eps;
    }
    v2f sc[H];
    if (!ONE_T) {
#pragma unroll
        for (int i = 0; i < H; i++) sc[i] = qw[i] * inv_sqrt_selected2(e[i] * e[i] * e[i]);
    } else {
        bool tie = false;
#pragma unroll
        for (int i = 0; i < H; i++) {
            sc[i] = inv_sqrt_guarded2(e[i] * e[i] * e[i], tie);
            // (keeps the step's last fma above the branch: sunk below it, its operands -- 16 VGPRs --
            // stay live across the branch and the kernel drops from 6 to 5 waves per SIMD)
            asm volatile("" : "+v"(sc[i]));
        }
        if (__any(tie)) {                               // about one group in 500
#pragma unroll
            for (int i = 0; i < H; i++) sc[i] = inv_sqrt_selected2(e[i] * e[i] * e[i]);
        }
#pragma unroll
        for (int i = 0; i < H; i++) sc[i] = qw[i] * sc[i];
    }
#pragma unroll
    for (int i = 0; i < H; i++) {                       // sums in list order
        const v2f px = r.rx[i] * sc[i], py = r.ry[i] * sc[i], pz = r.rz[i] * sc[i];
        ax += px.x; ay += py.x; az += pz.x;
        ax += px.y; ay += py.y; az += pz.y;
    }
}

// Fast-math finish (FMA + v_rsq) on softened distances.
template <int NQ>
__device__ __forceinline__ void pairs_finish_fast(const PairRows<NQ> &r, const v2f (&qw)[NQ / 2],
                                                  float &ax, float &ay, float &az)
{
    constexpr int H = NQ / 2;
    v2f sc[H];
#pragma unroll
    for (int i = 0; i < H; i++) {
        v2f q; q.x = __builtin_amdgcn_rsqf(r.d[i].x); q.y = __builtin_amdgcn_rsqf(r.d[i].y);
        sc[i] = qw[i] * (q * q * q);
    }
#pragma unroll
    for (int i = 0; i < H; i++) {
        ax = fmaf(r.rx[i].x, sc[i].x, ax); ay = fmaf(r.ry[i].x, sc[i].x, ay); az = fmaf(r.rz[i].x, sc[i].x, az);
        ax = fmaf(r.rx[i].y, sc[i].y, ax); ay = fmaf(r.ry[i].y, sc[i].y, ay); az = fmaf(r.rz[i].y, sc[i].y, az);
    }
}

#if defined(PSAMD_TWO_TRANSCENDENTALS)      // (A/B builds)
constexpr bool ONE_T_DEFAULT = false;
#else
constexpr bool ONE_T_DEFAULT = true;
#endif
template <int NQ, bool ONE_T = ONE_T_DEFAULT>
__device__ __forceinline__ void pairsN_exact_lean(const DevParams &P, const PairCtx &c, const v2f (&qx)[NQ / 2],
                                                  const v2f (&qy)[NQ / 2], const v2f (&qz)[NQ / 2],
                                                  const v2f (&qw)[NQ / 2], int gj0,
                                                  const float *__restrict__ snap_age,
                                                  const int *__restrict__ sorted_id,
                                                  float &ax, float &ay, float &az, int &flag)
{
    PairRows<NQ> r;
    pairs_dist<NQ, false>(c, qx, qy, qz, 0.f, r);
    pairs_finish_exact<NQ, ONE_T>(P, c, r, qw, gj0, snap_age, sorted_id, ax, ay, az, flag);
}

// returns the smallest softened squared distance (d2 + eps2) of the group, for the collision gate
template <int NQ>
__device__ __forceinline__ float pairsN_fast(const PairCtx &c, const v2f (&qx)[NQ / 2], const v2f (&qy)[NQ / 2],
                                             const v2f (&qz)[NQ / 2], const v2f (&qw)[NQ / 2], float eps2,
                                             float &ax, float &ay, float &az)
{
    PairRows<NQ> r;
    pairs_dist<NQ, true>(c, qx, qy, qz, eps2, r);
    pairs_finish_fast<NQ>(r, qw, ax, ay, az);
    return r.dm;
}

__device__ __forceinline__ void pair1_exact_lean(const DevParams &P, const PairCtx &c, const float4 q, int gj,
                                                 const float *__restrict__ snap_age,
                                                 const int *__restrict__ sorted_id,
                                                 float &ax, float &ay, float &az, int &flag)
{
    const float rx = q.x - c.xi, ry = q.y - c.yi, rz = q.z - c.zi;
    const float d2 = rx * rx + ry * ry + rz * rz;
    const float e = (float)((double)d2 + P.eps2);
    if (__any(c.scan && !(d2 > P.coll_d2_gate))) {
        if (c.scan && !(d2 > P.coll_d2_gate) && gj != c.gi)
            flag = max(flag, collide_exact(P, d2, c.age_i, c.id_i, snap_age[gj], sorted_id[gj]));
    }
    const float s = q.w * inv_sqrt_selected(e * e * e);
    ax += rx * s; ay += ry * s; az += rz * s;
}

// ------------------------------------------------------------------ two-pass pair stage
// The reference scans a particle's neighbours for collisions first and runs the force loop
// only if there was none (ps.cpp:1182-1263): a particle that dies or "survives" a collision
// this step is not integrated and its acceleration is never looked at.  In a dense cloud
// that is a large share (42 % in the first step of the N = 2^20 benchmark cloud).  The lean
// modes do the same: k_collide_cell settles every particle's flag from the few bodies that can
// reach it -- its own cell and the neighbours' bodies near the shared faces (the halo lists
// k_sort_cells filled) -- then k_build_active lists, per cell, the particles that still need
// a force, and the force pass walks the 27-cell stencil for those only.
//
// Bodies in the stencil of local cell (i1, i2, i3) that are no kids, counted by one wave.  For the particle
// whose own position is not a number: the lean force walks let a particle meet itself and the kids because
// r * 0 adds nothing -- not so when r is no number.  The reference skips both (ps.cpp:1258,
// app_common.cu:240-243): with no other body in the stencil the particle's sum is +0 (this count is 1:
// itself), with one it is no number either way.
__device__ __forceinline__ int stencil_adults(const DevParams &P, int i1, int i2, int i3, const int *__restrict__ cell_start,
                                              const float *__restrict__ snap_age)
{
    const int lane = threadIdx.x & 63;
    int total = 0;
    for (int k = 0; k < STENCIL; k++) {
        const int nc = __builtin_amdgcn_readfirstlane(local_cell(P, i3 + c_stencil[k][2], i1 + c_stencil[k][1], i2 + c_stencil[k][0]));
        if (nc < 0) continue;
        const int b = __builtin_amdgcn_readfirstlane(cell_start[nc]);
        const int n = __builtin_amdgcn_readfirstlane(min(cell_start[nc + 1] - b, P.max_per_cell));
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            total += __popcll(__ballot(j < n && !(snap_age[b + (j < n ? j : 0)] < P.kid_thr)));
        }
    }
    return total;
}

// One particle against `n` bodies given as arrays (wave-uniform pointers, so the loads are
// scalar loads).  bodyBodyCollision (app_common.cu:269-301) without a branch: the reference's
// test (double)sqrtf(r.r) > COLLISION_RADIUS is, sqrtf being correctly rounded and monotone,
// r.r > coll_d2_max for a float found by bisection when the context is created; a body that can
// never collide (kid, over age) carries cid = -1, otherwise its slot id; and "flag = max over
// the hits of (id_i > id_j ? 1 : 2)" is two lane masks: met someone with a higher id (2, the
// lower id dies), met someone with a lower one (1).  The particle itself drops out because
// neither id comparison holds for it.
__device__ __forceinline__ void collide_scan(const DevParams &P, float xi, float yi, float zi, int id_i, bool scan,
                                             const float *__restrict__ bx, const float *__restrict__ by,
                                             const float *__restrict__ bz, const int *__restrict__ bcid, int n,
                                             unsigned long long &hi_mask, unsigned long long &lo_mask)
{
    // hi_mask / lo_mask: lanes that met a body with a higher / lower id (wave-uniform words: the
    // bookkeeping is scalar work).  A group's sixteen bodies AND their ids arrive in one batch of scalar
    // loads; a body's test is its distance arithmetic and one compare, the two id compares happen only
    // for the body some lane is within reach of (about one in eight at the benchmark's density), behind
    // a scalar branch.  (Before: a group minimum first, then -- nearly every group has a hit -- a loop of
    // sixteen compare-and-branch steps and a scalar load of the hit's id that the walk had to wait
    // for; that bookkeeping cost as much as the arithmetic.  Two groups of loads in flight were
    // tried and were slower.)
    constexpr int NB = 16;
    const v2f x2 = {xi, xi}, y2 = {yi, yi}, z2 = {zi, zi};
    const float dmax = P.coll_d2_max;
    const unsigned uid = (unsigned)id_i;
    const unsigned long long scan_mask = __builtin_amdgcn_ballot_w64(scan);
    auto hit = [&](float d2, int cj) {
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(!(d2 > dmax)) & scan_mask;
        if (hm) {
            hi_mask |= hm & __builtin_amdgcn_ballot_w64(cj > id_i);
            lo_mask |= hm & __builtin_amdgcn_ballot_w64((unsigned)cj < uid);      // (a body that never collides carries -1: not below any id)
        }
    };
    int j = 0;
    // (Two groups of eight in flight -- the next group's loads issued before the current one is worked
    // through, since the wave spends half its cycles parked at s_waitcnt -- were measured again with the
    // ids in the batch: 189 us against 147 for flags + plan.  The double set of bodies costs SGPR spills.)
    for (; j + NB <= n; j += NB) {
        // all the group's loads and distances first (one batch of scalar loads, one wait), then the tests
        int cid[NB];
        v2f d[NB / 2];
#pragma unroll
        for (int i = 0; i < NB; i++) cid[i] = bcid[j + i];
#pragma unroll
        for (int i = 0; i < NB / 2; i++) {
            const v2f rx = v2f{bx[j + 2 * i], bx[j + 2 * i + 1]} - x2, ry = v2f{by[j + 2 * i], by[j + 2 * i + 1]} - y2,
                      rz = v2f{bz[j + 2 * i], bz[j + 2 * i + 1]} - z2;
            d[i] = rx * rx + ry * ry + rz * rz;
        }
#pragma unroll
        for (int i = 0; i < NB / 2; i++) { asm volatile("" : "+v"(d[i])); }      // (keeps the tests below the arithmetic: the loads stay one batch)
#pragma unroll
        for (int i = 0; i < NB / 2; i++) { hit(d[i].x, cid[2 * i]); hit(d[i].y, cid[2 * i + 1]); }
    }
    for (; j < n; j++) {
        const float rx = bx[j] - xi, ry = by[j] - yi, rz = bz[j] - zi;
        hit(rx * rx + ry * ry + rz * rz, bcid[j]);
    }
}

// The collision flags of every particle of the computed cells and, for the particles that will not be
// integrated or feel no force (kids), the final force4 record.  One workgroup per cell, with the
// candidates culled first.  A collision needs the two within COLLISION_RADIUS (0.4 against a 5.0 cell): of
// the ~400 bodies a cell's particle could meet (its cell's and the halo list's) a handful are near enough
// to be worth the arithmetic.  The workgroup bins those bodies (the ones that can collide at all: cid >= 0)
// on a grid of up to 10^3 bins over the cell's box grown by the halo reach -- a counting sort in LDS:
// census with the body's rank in its bin from the atomic's return, prefix, scatter of (x, y, z, id) rows,
// the bodies held in registers between the passes -- and a particle then tests the bodies of its bin and
// the bins around it only: nine runs (a row of three bins along x is one run of the sorted rows), nine
// bodies in all at the benchmark's density.  A bin is wider than the reach, so two bodies within it of
// each other are never more than one bin apart on any axis (the bin coordinate is a monotone function of
// the position, clamped into the grid); the test itself is the arithmetic of collide_scan on the same
// operands, and "any hit with a higher / a lower id" does not depend on the order the candidates come
// in: the flags are the same bits.  A cell with more bodies than the LDS rows hold, or whose halo list
// overflowed, takes collide_scan over everything.
// (Until round 3 this was one wave per 64-particle slice running collide_scan over all ~400 bodies, 123 us
// at N = 2^20; a workgroup per cell with the 400 bodies in LDS read back as broadcast rows was 133 us --
// a broadcast ds_read_b128 still occupies the LDS pipe for its 64 lanes.  With the bins: 41 us, of which
// the runs are 18.  Steps on the way, flags + plan: 146 us -> 98 (bins) -> 91 (bodies kept in registers,
// run bounds read in one batch) -> 69 (two bodies a turn, ids by max / min instead of a branch at a hit)
// -> 67 (the particle's own position and id from the binning registers); profiles/r3_ab_collide.txt.)
constexpr int COLL_NB = 10;
template <int CAP>
__global__ __launch_bounds__(256, CAP <= 1024 ? 6 : 3) void k_collide_cell(DevParams P, const int *__restrict__ cell_start,
                                                      const float *__restrict__ snap_soa, const float *__restrict__ snap_age,
                                                      const int *__restrict__ sorted_id, const int *__restrict__ snap_cid,
                                                      const int *__restrict__ halo_count, const float *__restrict__ halo_f,
                                                      const int *__restrict__ halo_id, int *__restrict__ active_list,
                                                      int *__restrict__ active_count, int *__restrict__ task_cost,
                                                      float4 *__restrict__ force4)
{
    constexpr int KB = CAP / 256;                                 // bodies a thread bins (held in registers between the passes)
    __shared__ float4 s_body[CAP];
    __shared__ int s_bin[COLL_NB * COLL_NB * COLL_NB + 1];
    __shared__ int s_wtot[4];
    const int c = comp_cell(P, blockIdx.x);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = __builtin_amdgcn_readfirstlane(cell_start[c]);
    const int cnt = __builtin_amdgcn_readfirstlane(min(cell_start[c + 1] - base, P.max_per_cell));
    if (cnt <= 0) return;
    const int nh = __builtin_amdgcn_readfirstlane(halo_count[c]);
    const size_t cap = (size_t)P.sorted_cap;
    const size_t hat = (size_t)c * HALO_CAP, hplane = (size_t)P.n_local_cells * HALO_CAP;
    int i1, i2, i3;
    cell_coords(P, c, i1, i2, i3);
    // the bins: nb per axis over [-reach, cell + reach) in the cell's own coordinates
    const float cs = (float)P.cell_size, reach = P.halo_reach * 1.01f + 1e-3f, box = cs + 2.0f * reach;
    const int nb = max(1, min(COLL_NB, (int)(box / (reach * 1.05f))));
    const float per_unit = (float)nb / box;
    const float ox = ((float)i2 - (float)(P.G / 2)) * cs - reach, oy = ((float)(P.G / 2) - (float)i1) * cs + reach,
                oz = ((float)(P.G / 2) - (float)i3) * cs + reach;      // u = x - ox, oy - y, oz - z: offsets into the grown box
    auto bin1 = [&](float u) { return max(0, min(nb - 1, (int)(u * per_unit))); };
    bool binned = nh <= HALO_CAP && cnt + nh <= CAP;
    const int nbins = nb * nb * nb;
    // what one force task of this cell walks: the population of its stencil (the last wave, while the others' loads fly)
    if (wv == 3) {
        int n = 0;
        if (lane < STENCIL) {
            const int nc = local_cell(P, i3 + c_stencil[lane][2], i1 + c_stencil[lane][1], i2 + c_stencil[lane][0]);
            if (nc >= 0) n = min(cell_start[nc + 1] - cell_start[nc], P.max_per_cell);
        }
        n = wave_incl_scan(n);
        if (lane == 63) task_cost[c] = n;
    }
    float4 q[KB];
    if (binned) {
        const int nbody = cnt + nh;
        // all the loads in one batch (the coordinates do not wait for the ids), the bins zeroed meanwhile
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int e = tid + 256 * k;
            const bool own = e < cnt;
            q[k] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (e < nbody) {
                q[k].w = __int_as_float(own ? snap_cid[base + e] : halo_id[hat + (e - cnt)]);
                q[k].x = own ? snap_soa[base + e] : halo_f[hat + (e - cnt)];
                q[k].y = own ? snap_soa[cap + base + e] : halo_f[hplane + hat + (e - cnt)];
                q[k].z = own ? snap_soa[2 * cap + base + e] : halo_f[2 * hplane + hat + (e - cnt)];
            }
        }
        // A candidate whose position is not a number passes the reference's distance test against every particle
        // that scans it (halo_dirs): no bins for this cell, collide_scan meets it with everything.
        bool wild = false;
#pragma unroll
        for (int k = 0; k < KB; k++) wild |= __float_as_int(q[k].w) >= 0 && !finite3(q[k].x, q[k].y, q[k].z);
        for (int b = tid; b <= nbins; b += 256) s_bin[b] = 0;
        if (__syncthreads_or(wild)) binned = false;
    }
    if (binned) {
        int bin[KB], rank[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) {
            bin[k] = -1; rank[k] = 0;
            if (__float_as_int(q[k].w) >= 0) {
                bin[k] = (bin1(oz - q[k].z) * nb + bin1(oy - q[k].y)) * nb + bin1(q[k].x - ox);
                rank[k] = atomicAdd(&s_bin[bin[k]], 1);
            }
        }
        __syncthreads();
        // exclusive prefix over the bins: a run of bins per thread, the runs' totals through the waves
        const int per = (nbins + 255) / 256;
        const int b0 = min(nbins, tid * per), b1 = min(nbins, b0 + per);
        int mine = 0;
        for (int b = b0; b < b1; b++) mine += s_bin[b];
        const int incl = wave_incl_scan(mine);
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int run = incl - mine;
        for (int k = 0; k < wv; k++) run += s_wtot[k];
        for (int b = b0; b < b1; b++) { const int n = s_bin[b]; s_bin[b] = run; run += n; }
        if (tid == 255) s_bin[nbins] = run;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KB; k++)
            if (bin[k] >= 0) s_body[s_bin[bin[k]] + rank[k]] = q[k];
        __syncthreads();
    }
    const float dmax = P.coll_d2_max;
    // the flag, the final force4 record of the particles the force pass does not visit, and the list of the ones it does
    // (flag 0 and not a kid), packed at active_list[cell_start[c] ...] in whatever order the cell's waves arrive
    auto finish = [&](bool valid, int gi, bool dead, bool kid, bool met_higher, bool met_lower, bool alone = false) {
        int flag = met_higher ? 2 : met_lower ? 1 : 0;
        if (dead) flag = 2;                                          // ps.cpp:1183
        if (valid) force4[gi] = make_float4(0.f, 0.f, 0.f, __int_as_float(flag));   // final unless the force pass overwrites it
        const bool on = valid && flag == 0 && !kid && !alone;       // (alone: stencil_adults)
        const unsigned long long m = __ballot(on);
        if (m) {
            int off = 0;
            if (lane == 0) off = atomicAdd(&active_count[c], __popcll(m));
            off = __builtin_amdgcn_readfirstlane(off);
            if (on) active_list[base + off + __popcll(m & ((1ull << lane) - 1ull))] = gi;
        }
    };
    if (binned) {
        // thread tid's k-th body is the cell's particle tid + 256 k (the cell's own come first): position and id are
        // in registers already; only a particle that cannot collide needs its age looked up (dead or kid?)
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int first = wv * 64 + 256 * k;
            if (first >= cnt) break;
            const bool valid = lane < cnt - first;
            const int gi = base + first + (valid ? lane : 0);
            const int id_i = __float_as_int(q[k].w);
            const bool scan = valid && id_i >= 0;
            bool dead = false, kid = false, met_higher = false, met_lower = false;
            if (valid && id_i < 0) { const float age_i = snap_age[gi]; dead = age_i > P.life_thr; kid = age_i < P.kid_thr; }
            if (scan) {
                const float xi = q[k].x, yi = q[k].y, zi = q[k].z;
                const unsigned uid = (unsigned)id_i;
                const int bx = bin1(xi - ox), by = bin1(oy - yi), bz = bin1(oz - zi);
                const int x0 = max(bx - 1, 0), x1 = min(bx + 1, nb - 1);
                // the nine runs' bounds first (one batch of LDS reads), then the runs
                int j0[9], j1[9];
#pragma unroll
                for (int r = 0; r < 9; r++) {
                    const int z = bz + r / 3 - 1, y = by + r % 3 - 1;
                    const bool in = z >= 0 && z < nb && y >= 0 && y < nb;
                    const int row = (z * nb + y) * nb;
                    j0[r] = in ? s_bin[row + x0] : 0;
                    j1[r] = in ? s_bin[row + x1 + 1] : 0;
                }
                // two bodies a turn (an odd run's last body twice: the result is an OR over the hits), no branch
                // at a hit: the highest id met as a signed number and the lowest as an unsigned one say, against
                // the particle's own, whether there was one above and one below (a miss counts as id -1: neither)
                const v2f x2 = {xi, xi}, y2 = {yi, yi}, z2 = {zi, zi};
                int hi = -1;
                unsigned lo = ~0u;
#pragma unroll
                for (int r = 0; r < 9; r++)
                    for (int j = j0[r]; j < j1[r]; j += 2) {
                        const float4 qa = s_body[j], qb = s_body[min(j + 1, j1[r] - 1)];
                        const v2f rx = v2f{qa.x, qb.x} - x2, ry = v2f{qa.y, qb.y} - y2, rz = v2f{qa.z, qb.z} - z2;
                        const v2f d2 = rx * rx + ry * ry + rz * rz;
                        const int ca = !(d2.x > dmax) ? __float_as_int(qa.w) : -1, cb = !(d2.y > dmax) ? __float_as_int(qb.w) : -1;
                        hi = max(hi, max(ca, cb));
                        lo = min(lo, min((unsigned)ca, (unsigned)cb));
                    }
                met_higher = hi > id_i;
                met_lower = lo < uid;
            }
            finish(valid, gi, dead, kid, met_higher, met_lower);
        }
    } else {
        for (int first = wv * 64; first < cnt; first += 256) {
            const bool valid = lane < cnt - first;
            const int gi = base + first + (valid ? lane : 0);
            const float xi = snap_soa[gi], yi = snap_soa[cap + gi], zi = snap_soa[2 * cap + gi];
            const float age_i = snap_age[gi];
            const int id_i = sorted_id[gi];
            const bool dead = age_i > P.life_thr, kid = age_i < P.kid_thr;
            const bool scan = valid && !dead && !kid;
            unsigned long long hi_mask = 0, lo_mask = 0;
            collide_scan(P, xi, yi, zi, id_i, scan, snap_soa + base, snap_soa + cap + base, snap_soa + 2 * cap + base,
                         snap_cid + base, cnt, hi_mask, lo_mask);
            // (a particle whose own position is not a number passes the distance test against EVERY body of its
            // stencil, not only the ones near the faces: its wave walks the whole stencil)
            if (nh <= HALO_CAP && !__any(scan && !finite3(xi, yi, zi))) {
                collide_scan(P, xi, yi, zi, id_i, scan, halo_f + hat, halo_f + hplane + hat, halo_f + 2 * hplane + hat, halo_id + hat, nh,
                             hi_mask, lo_mask);
            } else {
                // the halo list overflowed (denser than the container admits in steady state): whole stencil
                for (int k = 1; k < 27; k++) {
                    const int nc = __builtin_amdgcn_readfirstlane(local_cell(P, i3 + c_stencil[k][2], i1 + c_stencil[k][1], i2 + c_stencil[k][0]));
                    if (nc < 0) continue;
                    const int nbase = __builtin_amdgcn_readfirstlane(cell_start[nc]);
                    const int n = __builtin_amdgcn_readfirstlane(min(cell_start[nc + 1] - nbase, P.max_per_cell));
                    collide_scan(P, xi, yi, zi, id_i, scan, snap_soa + nbase, snap_soa + cap + nbase, snap_soa + 2 * cap + nbase, snap_cid + nbase, n,
                                 hi_mask, lo_mask);
                }
            }
            // (a particle whose position is not a number and that has no other adult in its stencil: its sum is +0)
            bool alone = false;
            if (__any(scan && !finite3(xi, yi, zi))) {
                const int adults = stencil_adults(P, i1, i2, i3, cell_start, snap_age);   // (all lanes: the count is a wave's work)
                alone = scan && !finite3(xi, yi, zi) && adults <= 1;
            }
            finish(valid, gi, dead, kid, (hi_mask >> lane) & 1ull, (lo_mask >> lane) & 1ull, alone);
        }
    }
}

// The plan of the balanced force pass, one launch of eight workgroups (one per XCD run of wave
// slots).  Every workgroup works out, for itself, in LDS:
//   (1) the prefix over the computed cells (the lent ones first: their results travel back to the
//       rank that owns them) of the 64-slices of the active lists and of what those tasks walk
//       (a task of cell c walks task_cost[c] bodies, the population of the cell's stencil);
//       with `merge`, only full slices become ordinary tasks and the leftovers (a cell's last,
//       partly filled slice: 20 of 64 lanes on average once the collided particles are gone) are
//       packed, up to four cells to a wave, into merged tasks;
//   (2) where every wave slot of ITS run starts: the pass's work is the list of (task, stencil
//       step) units -- task-major, 27 steps per task -- a unit costs the bodies of the neighbour
//       cell it visits, and wave slot s takes the units from wave_pos[s] up to wave_pos[s + 1]:
//       equal shares of the cost, cut at unit boundaries.  The eight runs start at whole tasks, so a
//       task that is cut is always continued by a workgroup of the same run.
// The task list, the packs and the frame scalars are the same whichever workgroup writes them; each
// writes a share.  (These were three launches, k_build_active / k_active_tasks / k_split_tasks, 60 us
// of mostly one-workgroup latency on the step's critical path; the prefixes are cheap enough to
// be recomputed eight times.)
// merge: 0 every slice is an ordinary task; 1 the packs are the merged tasks of k_pairs_merged (run
// beside the balanced pass); 2 the packs are tasks of the balanced pass itself (tile walk): pack m is
// task n_tasks2 + m, with one virtual "cell" ncomp + m in the prefix arrays.
constexpr int PLAN_LDS = 6144;        // prefix entries (computed cells + virtual pack cells + 1) kept in LDS
__global__ __launch_bounds__(1024) void k_plan_force(DevParams P, int nw, int merge, const int *__restrict__ cell_start_g,
                                                     const int *__restrict__ active_count, const int *__restrict__ task_cost,
                                                     int *__restrict__ task_list2, int *__restrict__ ctask_start_g,
                                                     long long *__restrict__ cost_start_g, int4 *__restrict__ merged_tasks,
                                                     long long *__restrict__ wave_pos, FrameScalars *fs, unsigned long long *trace)
{
#ifdef PSAMD_PLAN_TRACE    // diagnostic build: time stamps (100 MHz) of workgroup x's phases in trace[8 x ...]
#define PT(i) do { if (threadIdx.x == 0) trace[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PT(i) do {} while (0)
#endif
    PT(0);
    __shared__ long long s_cost[PLAN_LDS + 1];
    __shared__ int s_task[PLAN_LDS + 1];
    __shared__ int s_ac[PLAN_LDS];                     // active_count | task_cost << 13 of the j-th computed cell
    __shared__ long long wave_tot[16], wave_cost[16], wave_pcost[16];
    __shared__ long long s_run[2];
    __shared__ long long s_runcost[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, x = blockIdx.x;
    const int ncomp = comp_count(P);
    const bool ext = merge == 2;
    const bool in_lds = (ext ? 2 * ncomp : ncomp) + 1 <= PLAN_LDS && P.max_per_cell < (1 << 13);
    long long *cost_start = in_lds ? s_cost : cost_start_g;
    int *ctask_start = in_lds ? s_task : ctask_start_g;
    if (in_lds) for (int j = tid; j < ncomp; j += 1024) { const int c = comp_cell(P, j), n = active_count[c]; s_ac[j] = n | ((n ? task_cost[c] : 0) << 13); }
    __syncthreads();
    PT(1);
    auto act_of = [&](int j) { return in_lds ? (s_ac[j] & 0x1fff) : active_count[comp_cell(P, j)]; };
    auto cost_of = [&](int j) { return in_lds ? (s_ac[j] >> 13) : task_cost[comp_cell(P, j)]; };

    // ---- (1) prefixes, task list, packs ----
    const int per = (ncomp + 1023) / 1024;
    const int c0 = min(ncomp, tid * per), c1 = min(ncomp, c0 + per);
    // the leftovers are packed greedily, in cell order, a run of cells per thread: six (two packs of
    // three 20-lane leftovers) where there are threads enough -- each step of the greedy walk is a
    // dependent LDS round trip, and this walk is done twice
    const int pper = max(6, (ncomp + 1023) / 1024);
    const int p0 = min(ncomp, tid * pper), p1 = min(ncomp, p0 + pper);
    // out / cost_out (may be null): the packs and, per pack, what its wave walks (its longest stencil)
    auto pack = [&](int4 *out, long long *cost_out, long long cost_base, long long *cost_sum) -> int {
        int npack = 0, used = 0, ng = 0, pc = 0;
        long long acc = 0;
        int4 cur = make_int4(-1, -1, -1, -1);
        auto flush = [&]() {
            if (out) out[npack] = cur;
            if (cost_out) cost_out[npack] = cost_base + acc;
            acc += pc; npack++;
            cur = make_int4(-1, -1, -1, -1); used = 0; ng = 0; pc = 0;
        };
        for (int j = p0; j < p1; j++) {
            const int r = act_of(j) & 63;
            if (r == 0) continue;
            const int c = comp_cell(P, j);
            if (ng == 4 || used + r > 64) flush();
            if (ng == 0) cur.x = c; else if (ng == 1) cur.y = c; else if (ng == 2) cur.z = c; else cur.w = c;
            ng++; used += r; pc = max(pc, cost_of(j));
        }
        if (ng) flush();
        if (cost_sum) *cost_sum = acc;
        return npack;
    };
    long long mine = 0, mycost = 0, mypcost = 0;   // tasks (low word) and packs (high word); bodies the tasks walk; ... the packs walk
    for (int j = c0; j < c1; j++) {
        const int n = act_of(j), nt = merge ? (n >> 6) : ((n + 63) >> 6);
        mine += nt;
        mycost += (long long)nt * cost_of(j);
    }
    if (merge) mine |= (long long)pack(nullptr, nullptr, 0, &mypcost) << 32;
    long long incl = mine, cincl = mycost, pincl = mypcost;
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(incl, d), oc = __shfl_up(cincl, d), op = __shfl_up(pincl, d);
        if (lane >= d) { incl += o; cincl += oc; pincl += op; }
    }
    if (lane == 63) { wave_tot[wv] = incl; wave_cost[wv] = cincl; wave_pcost[wv] = pincl; }
    __syncthreads();
    long long run2 = incl - mine, total2 = 0, crun = cincl - mycost, ctotal = 0, prun = pincl - mypcost, ptotal = 0;
    for (int k = 0; k < 16; k++) {
        if (k < wv) { run2 += wave_tot[k]; crun += wave_cost[k]; prun += wave_pcost[k]; }
        total2 += wave_tot[k]; ctotal += wave_cost[k]; ptotal += wave_pcost[k];
    }
    int run = (int)(run2 & 0xffffffffll);
    const int total = (int)(total2 & 0xffffffffll), npacks = (int)(total2 >> 32);
    PT(2);
    const bool my_share = (tid & 7) == x;           // the lists in memory: each workgroup writes an eighth
    for (int j = c0; j < c1; j++) {
        const int n = merge ? (act_of(j) >> 6) : ((act_of(j) + 63) >> 6);
        ctask_start[j] = run; cost_start[j] = crun;
        if (my_share && n) { const int c = comp_cell(P, j); for (int sl = 0; sl < n; sl++) task_list2[run + sl] = c * P.slices + sl; }
        run += n;
        crun += (long long)n * cost_of(j);
    }
    if (merge) {
        const int m0 = (int)(run2 >> 32);
        const int np = pack(x == 0 ? merged_tasks + m0 : nullptr, ext ? cost_start + ncomp + m0 : nullptr, ctotal + prun, nullptr);
        if (ext) for (int m = 0; m < np; m++) ctask_start[ncomp + m0 + m] = total + m0 + m;
    }
    const int ncells = ncomp, nent = ncomp + (ext ? npacks : 0), ntask = total + (ext ? npacks : 0);
    const long long T = ctotal + (ext ? ptotal : 0);
    if (tid == 0) {
        ctask_start[nent] = ntask;
        cost_start[nent] = T;
        if (x == 0) { fs->n_tasks2 = total; fs->n_merged = npacks; fs->cost_total = T; }
    }
    if (!in_lds) __threadfence();                    // (every workgroup wrote the same values; this one reads its own)
    __syncthreads();
    PT(3);
    if (nw <= 0) return;                             // (unbalanced pass: only the lists were wanted)

    // ---- (2) the wave slots of run x ----
    // A position in the pass's work is (task, cost already walked inside the task): which stencil step
    // that is depends on the populations of the task's stencil, which the wave that starts (or stops)
    // there looks up anyway -- k_pairs_balanced turns the residual into a step.  (Walking the 27 counts
    // here, per wave slot, was most of this kernel's 40 us.)  For a merged pack the residual IS the
    // step (its steps are taken as equally long), marked by bit 30.  whole = round up to the next task start.
    auto pos_at = [&](long long v, bool whole) -> long long {
        if (v >= T) return (long long)ntask << 32;
        int a = 0, b = nent - 1;                          // last entry whose tasks start at or before v
        while (a < b) { const int m = (a + b + 1) >> 1; if (cost_start[m] <= v) a = m; else b = m - 1; }
        const int nt = ctask_start[a + 1] - ctask_start[a];
        if (a >= ncells) {                                // a merged pack: one task
            const long long S = cost_start[a + 1] - cost_start[a], off = v - cost_start[a];
            const int k = S > 0 ? (int)min((long long)(STENCIL - 1), off * STENCIL / S) : 0;
            if (whole) return (long long)(ctask_start[a] + (off > 0 ? 1 : 0)) << 32;
            return ((long long)ctask_start[a] << 32) | (long long)(k | (1 << 30));
        }
        const int S = cost_of(a);
        if (nt == 0 || S <= 0) return (long long)ctask_start[a + 1] << 32;     // (v < T: cannot be the last cell)
        const long long off = v - cost_start[a];
        const int q = (int)min((long long)(nt - 1), off / S);
        const int r = (int)(off - (long long)q * S);
        const int t = ctask_start[a] + q;
        if (whole) return (long long)(t + (r > 0 ? 1 : 0)) << 32;
        return ((long long)t << 32) | (long long)r;
    };
    auto cost_of_task_start = [&](int t) -> long long {
        if (t >= ntask) return T;
        int a = 0, b = nent - 1;
        while (a < b) { const int mm = (a + b + 1) >> 1; if (ctask_start[mm] <= t) a = mm; else b = mm - 1; }
        if (a >= ncells) return cost_start[a];             // a merged pack is one task
        return cost_start[a] + (long long)(t - ctask_start[a]) * cost_of(a);
    };
    const int m = nw >> 3;                                // wave slots per XCD run (nw is a multiple of 32)
    if (tid < 2) {
        s_run[tid] = pos_at(T * (x + tid) / 8, true);
        s_runcost[tid] = cost_of_task_start((int)(s_run[tid] >> 32));
    }
    __syncthreads();
    PT(4);
    const long long run_lo = s_run[0], run_hi = s_run[1];
    const long long lo = s_runcost[0], hi = s_runcost[1];
    for (int j = tid; j < m; j += 1024)                   // equal shares of the run's own cost range
        wave_pos[x * m + j] = j == 0 ? run_lo : max(run_lo, min(run_hi, pos_at(lo + (hi - lo) * j / m, false)));
    if (x == 7 && tid == 0) wave_pos[nw] = run_hi;        // = (ntask, 0)
    __syncthreads();
    PT(5);
#undef PT
}

// wave_pos -> wave_unit: one WAVE per wave-slot boundary.  The stencil step of a position (task, cost
// already walked inside the task) is the number of leading stencil cells the residual covers whole:
// 27 lanes look the cells' populations up, one scan, one ballot.  (One THREAD per boundary walking
// the 27 counts serially -- some 2000 instructions -- was the bulk of the old split kernel.)
__global__ __launch_bounds__(256) void k_resolve_steps(DevParams P, int nw, const int *__restrict__ cell_start,
                                                       const int *__restrict__ task_list, const long long *__restrict__ wave_pos,
                                                       int *__restrict__ wave_unit)
{
    const int s = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63);
    if (s > nw) return;
    const long long pos = wave_pos[s];
    const int t = __builtin_amdgcn_readfirstlane((int)(pos >> 32)), r = __builtin_amdgcn_readfirstlane((int)(pos & 0xffffffffll));
    int k = 0;
    if (r & (1 << 30)) k = r & 63;                           // a merged pack: the residual is the step
    else if (r > 0) {
        const int c = __builtin_amdgcn_readfirstlane(task_list[t]) / P.slices;
        int i1, i2, i3, cnt = 0;
        cell_coords(P, c, i1, i2, i3);
        if (lane < STENCIL) {
            const int nc = local_cell(P, i3 + c_stencil[lane][2], i1 + c_stencil[lane][1], i2 + c_stencil[lane][0]);
            if (nc >= 0) cnt = min(cell_start[nc + 1] - cell_start[nc], P.max_per_cell);
        }
        const int cum = wave_incl_scan(cnt);
        k = __popcll(__ballot(lane < STENCIL - 1 && cum <= r));
    }
    if (lane == 0) wave_unit[s] = t * STENCIL + k;
}

// One wave = 64 consecutive particles of one cell (four independent waves per workgroup).
// Neighbour cells are visited in the reference's stencil order, their bodies in list
// order, and every lane adds each body to its own particle's sum: each particle sees
// exactly the reference's sequence of fp32 additions (ps.cpp:1247-1259).
// MODE 0: exact with the compiler's correctly rounded sqrt/divide (any EPS2);
//      1: exact with the short sqrt/reciprocal above, NQ pairs per slow-branch test;
//      2: fast math (FMA + v_rsq), not bit-exact.
// SHARDED: the launch covers only this rank's run of the task list.
//
// Modes 1 and 2 never stage neighbour data at all.  It is the same for all 64 lanes, the
// ranges are wave-uniform, so the loads are scalar loads (s_load_dwordx8 from the SoA
// snapshot, straight out of L2 into SGPRs) and the packed fp32 instructions take the SGPR
// pairs as operands: no LDS, no vector registers for the bodies.  (An LDS tile read with
// ds_read_b128 by four waves per CU kept the LDS pipe ~70 % busy -- 16 cycles per wave
// read, scripts/microbench/lds_groups.hip -- and cost 4 % more time.)
// (Round 4 tried a third way -- every row of 16 lanes holds 16 bodies in VGPRs and the arithmetic takes them through
// DPP, `v_sub_f32_dpp rx, tile_x, xi row_newbcast:j`: no LDS, no scalar loads, the compiler fuses every broadcast.
// Bit-identical and 9-17 % slower everywhere: a DPP-modified v_sub / v_mul issues at half rate on gfx950.
// profiles/r4_ab_dpp_walk.txt, commit 2592be9.)
// Mode 0, the fallback for softening lengths outside the lean range, streams 64-body
// tiles through 1 KiB of LDS per wave.  No s_barrier: a wave only ever touches its own
// tile, and a wave's LDS operations complete in issue order, so a compiler-level fence
// is all the ordering needed.
#ifndef PSAMD_BALANCED_WAVES
#define PSAMD_BALANCED_WAVES 7      // resident waves per SIMD the scalar-walk force pass is built for (70 VGPRs; measured, exact / tolerance arithmetic: 6 waves 2.15 / 1.25 ms, 7 waves 2.11 / 1.22 ms)
#endif
#define PS_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

#ifdef PSAMD_WAVE_TRACE   // diagnostic build only: when and where did this wave run
#define PS_TRACE_BEGIN() const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime()
#define PS_TRACE_END() do { if ((threadIdx.x & 63) == 0) { \
        unsigned long long *t_ = trace + (size_t)3 * (blockIdx.x * 4 + (threadIdx.x >> 6)); \
        t_[0] = trace_t0; t_[1] = __builtin_amdgcn_s_memrealtime(); \
        t_[2] = ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf) << 32)   /* XCC_ID */ \
                | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); } } while (0)               /* HW_ID */
#else
#define PS_TRACE_BEGIN() do {} while (0)
#define PS_TRACE_END() do {} while (0)
#endif

// One task: 64 consecutive particles of one cell against the cell's stencil.
// Hand-off of a task's partial sums between the wave that walked the first stencil steps and
// the one that continues (balanced force pass).  Follows the guide's inter-workgroup recipe
// (cdna_hip_programming.md, Guideline 16): the payload is stored write-through with agent-scope
// atomic stores, the storing wave drains its stores, ONE lane raises the flag with an agent-scope
// atomic store; the consumer polls that one word relaxed and reads the payload with agent-scope
// atomic loads (they bypass the CU's L1, so no acquire fence is needed).  The flags are zeroed
// with the frame, before the launch.
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

// The flag word carries the stencil step the published sums stand at, so that a task cut in
// three or more pieces hands on correctly at every cut (each consumer waits for ITS step).
__device__ __forceinline__ void handoff_publish(float4 *slot, float ax, float ay, float az, int flag, bool valid, int *ready, int step)
{
    if (valid) {
        gu64 *p = (gu64 *)(unsigned long long *)slot;
        __hip_atomic_store(p, ((unsigned long long)__float_as_uint(ay) << 32) | __float_as_uint(ax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, ((unsigned long long)(unsigned)flag << 32) | __float_as_uint(az), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store((gu32 *)(unsigned int *)ready, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// false: the flag never came (bounded spin; the caller raises a sticky error)
__device__ __forceinline__ bool handoff_consume(const float4 *slot, float &ax, float &ay, float &az, int &flag, bool valid, const int *ready, int step)
{
    int ok = 0;
    if ((threadIdx.x & 63) == 0) {
        for (unsigned spins = 0; spins < (1u << 22); spins++) {
            if (__hip_atomic_load((gu32 *)(unsigned int *)ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)step) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // compiler-only: the loads below stay below the poll
    if (ok && valid) {
        gu64 *p = (gu64 *)(unsigned long long *)slot;
        const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ax = __uint_as_float((unsigned)a); ay = __uint_as_float((unsigned)(a >> 32));
        az = __uint_as_float((unsigned)b); flag = (int)(unsigned)(b >> 32);
    }
    return ok != 0;
}

// Stencil steps [k0, k1) of a task.  resume: the sums of steps < k0 come from the wave that
// walked them (ready != nullptr); a walk that stops before step 27 publishes its sums instead
// of finishing the particle.  The whole task is k0 = 0, k1 = 27, ready = nullptr.
// SETTLED: the collision flags are known already (two-pass mode: the balanced pass) -- nothing tracks distances for them
template <int MODE, int NQ, bool ALLP = false, bool SETTLED = false>
__device__ __forceinline__ void pairs_task(const DevParams &P, const int *__restrict__ cell_start,
                                           const SnapSoa snap4, const float *__restrict__ snap_soa,
                                           const float *__restrict__ snap_age, const int *__restrict__ sorted_id,
                                           float4 *__restrict__ force4, int task,
                                           float4 *tile, unsigned long long *trace,
                                           const int *__restrict__ active_list = nullptr,
                                           const int *__restrict__ active_count = nullptr,
                                           int k0 = 0, int k1 = STENCIL, int *ready = nullptr, FrameScalars *fs = nullptr,
                                           const FarCells far = FarCells(), int part = 0, int task_no = 0,
                                           const float *__restrict__ far_buf = nullptr, const int *__restrict__ far_start = nullptr,
                                           const int *__restrict__ far_n = nullptr)
{
    PS_TRACE_BEGIN();
    const int c = task / P.slices, slice = task - c * P.slices;
    const int base = cell_start[c];
    // two-pass mode: the slice is cut from the cell's list of particles that need a force
    const int cnt = active_list ? active_count[c] : min(cell_start[c + 1] - base, P.max_per_cell);
    const int first = slice * 64;
    if (first >= cnt) return;
    const int nvalid = min(64, cnt - first);
    const int lane = threadIdx.x & 63;
    const bool valid = lane < nvalid;
    const int gi = active_list ? active_list[base + first + (valid ? lane : 0)] : base + first + (valid ? lane : 0);
    const float4 me = snap4[gi];
    const float age_i = snap_age[gi];
    const int id_i = sorted_id[gi];
    const bool dead = age_i > P.life_thr;                      // ps.cpp:1183
    const bool kid = age_i < P.kid_thr;
    const bool scan = SETTLED ? false : (valid && !dead && !kid && !active_list);   // two-pass mode: flags are settled already

    int i1, i2, i3;
    cell_coords(P, c, i1, i2, i3);
    float ax = 0.f, ay = 0.f, az = 0.f;
    int flag = 0;
    const float eps2f = (float)P.eps2;

    // Lane k (< 27) looks up neighbour cell k of the stencil once: its range in the
    // sorted order, or an empty range if it lies outside the grid.
    int my_nb = 0, my_cnt = 0;
    if (lane < 27) {
        const int nc = local_cell(P, i3 + c_stencil[lane][2], i1 + c_stencil[lane][1], i2 + c_stencil[lane][0]);
        if (nc >= 0) {
            my_nb = cell_start[nc];
            my_cnt = min(cell_start[nc + 1] - my_nb, P.max_per_cell);
        }
    }
    if (MODE != 0) {
        const PairCtx ctx = {me.x, me.y, me.z, age_i, id_i, gi, scan};
        const size_t cap = (size_t)P.sorted_cap;
        if (k0 > 0 && !handoff_consume(force4 + gi, ax, ay, az, flag, valid, ready, k0)) {
            if (lane == 0) atomicOr(&fs->error, ERR_HANDOFF_TIMEOUT);
        }
        // The bodies [nb, nb + n) of one cell, from four planes of a snapshot (wave-uniform pointers: scalar loads).
        auto walk_cell = [&](const float *__restrict__ sx, const float *__restrict__ sy, const float *__restrict__ sz,
                             const float *__restrict__ sw, int nb, int n) {
            float dmin = 3.0e38f;
            int jj = 0;
            // NQ bodies per group.  (Fetching the next group between the distance stage and
            // the rest -- scalar loads return out of order, so it cannot go out any earlier --
            // was measured 3 % slower for the exact arithmetic on a full GPU and no faster
            // on a 1/8 share.)
            for (; jj + NQ <= n; jj += NQ) {
                v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], qw[NQ / 2];
#pragma unroll
                for (int i = 0; i < NQ / 2; i++) {
                    qx[i] = v2f{sx[jj + 2 * i], sx[jj + 2 * i + 1]};
                    qy[i] = v2f{sy[jj + 2 * i], sy[jj + 2 * i + 1]};
                    qz[i] = v2f{sz[jj + 2 * i], sz[jj + 2 * i + 1]};
                    qw[i] = v2f{sw[jj + 2 * i], sw[jj + 2 * i + 1]};
                }
                // (The compiler lets the masses' load sink to its use, behind the reciprocal square roots; pinned up
                // here with the coordinates' loads -- four in one batch -- the pass took the same time, 2.13 ms.)
                if (MODE == 1)
                    pairsN_exact_lean<NQ>(P, ctx, qx, qy, qz, qw, nb + jj, snap_age, sorted_id, ax, ay, az, flag);
                else
                    dmin = fminf(dmin, pairsN_fast<NQ>(ctx, qx, qy, qz, qw, eps2f, ax, ay, az));
            }
            for (; jj < n; jj++) {
                const float4 q = make_float4(sx[jj], sy[jj], sz[jj], sw[jj]);
                if (MODE == 1)
                    pair1_exact_lean(P, ctx, q, nb + jj, snap_age, sorted_id, ax, ay, az, flag);
                else
                    dmin = fminf(dmin, pair_fast(me.x, me.y, me.z, q, eps2f, ax, ay, az) + eps2f);
            }
            // fast math, rare: someone in this cell is within the (widened) collision gate of
            // one of my lanes; the exact rule is then evaluated on unfused distances
            const float gate_soft = (P.coll_d2_gate + eps2f) * 1.0001f;
            if (MODE == 2 && __any(scan && !(dmin > gate_soft))) {
                if (scan && !(dmin > gate_soft)) {
                    for (int j = 0; j < n; j++) {
                        const float rx = sx[j] - me.x, ry = sy[j] - me.y, rz = sz[j] - me.z;
                        const float d2 = rx * rx + ry * ry + rz * rz;
                        if (!(d2 > P.coll_d2_gate) && nb + j != gi)
                            flag = max(flag, collide_exact(P, d2, age_i, id_i, snap_age[nb + j], sorted_id[nb + j]));
                    }
                }
            }
        };
        // the stencil, in the reference's order (all-pairs mode: part 0 only)
        if (!ALLP || part == 0)
            for (int k = k0; k < k1; k++) {
                const int nb = __builtin_amdgcn_readlane(my_nb, k), n = __builtin_amdgcn_readlane(my_cnt, k);
                const float *sx = snap_soa + nb;
                walk_cell(sx, sx + cap, sx + 2 * cap, sx + 3 * cap, nb, n);
            }
        // All-pairs mode (ALLP, not in the reference): then every other cell in GLOBAL index order -- this
        // wave's part of them, the 64-cell blocks [blk_lo, blk_hi).  A far cell's bodies are summed on their
        // own and the cell's sum added to the particle's: an fp32 sum of a quarter of a million terms in one
        // chain would carry 4e-5 of rounding (measured at N = 2^18); the stencil's chain is the reference's
        // and stays as it is.  The bodies come from far_buf: the own snapshot on one GPU (local cell ==
        // global cell), the all-gathered snapshot of all ranks otherwise -- a pointer of its own, not a
        // choice between two, or the compiler cannot keep the loads scalar.
        if (ALLP) {
            const int nblk = (P.num_cells_global + 63) >> 6;
            const int blk_lo = nblk * part / ALLP_PARTS, blk_hi = nblk * (part + 1) / ALLP_PARTS;
            const size_t plane = (size_t)far.plane;
            for (int blk = blk_lo; blk < blk_hi; blk++) {
                // the block's 64 cell ranges in one vector load (a scalar load per cell, and the body loads
                // behind it, were two dependent round trips for 64 bodies of work); 0 bodies: a stencil cell
                const int c2 = blk * 64 + lane, GG = P.G * P.G;
                int far_nb = 0, far_cnt = 0;
                if (c2 < P.num_cells_global) {
                    const int j3 = c2 / GG, rem = c2 - j3 * GG, j1 = rem / P.G, j2 = rem - j1 * P.G;
                    if (!(abs(j1 - i1) <= 1 && abs(j2 - i2) <= 1 && abs(j3 - i3) <= 1)) {
                        far_nb = far_start[c2];
                        far_cnt = far_n ? far_n[c2] : min(far_start[c2 + 1] - far_nb, P.max_per_cell);
                    }
                }
                for (int j = 0; j < 64; j++) {
                    const int n = __builtin_amdgcn_readlane(far_cnt, j);
                    if (n == 0) continue;
                    const int nb = __builtin_amdgcn_readlane(far_nb, j);
                    const float near_x = ax, near_y = ay, near_z = az;
                    ax = 0.f; ay = 0.f; az = 0.f;
                    const float *sx = far_buf + nb;
                    walk_cell(sx, sx + plane, sx + 2 * plane, sx + 3 * plane, nb, n);
                    ax = near_x + ax; ay = near_y + ay; az = near_z + az;
                }
            }
        }
    } else {
        // Generic exact mode: tiles of 64 snapshot entries, in stencil order then list order.
        // The next tile's global load is issued before the current tile is consumed.  The lean
        // modes let a particle meet itself (r = 0 adds +0, exactly nothing) because
        // 1/sqrt(eps2^3) is finite on the range they are allowed on; this one also serves
        // softening lengths where it is not, so it skips the self pair explicitly, as the
        // reference does by id (ps.cpp:1258), and a kid neighbour too (app_common.cu:240: ai
        // comes back unchanged; its zeroed mass times an infinite 1/r^3 would be a NaN).
        int k = 0, t0 = 0;
        int nb = __shfl(my_nb, 0), ncnt = __shfl(my_cnt, 0);
        while (ncnt == 0 && ++k < 27) { nb = __shfl(my_nb, k); ncnt = __shfl(my_cnt, k); }
        bool have = k < 27;
        float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have && lane < min(64, ncnt)) pre = snap4[nb + lane];
        while (have) {
            const int c_nb = nb, c_t0 = t0, n = min(64, ncnt - t0);
            PS_WAVE_SYNC();                           // previous tile fully consumed
            if (lane < n) tile[lane] = pre;
            PS_WAVE_SYNC();
            t0 += 64;                                 // advance to the next non-empty tile
            if (t0 >= ncnt) {
                t0 = 0; ncnt = 0;
                while (ncnt == 0 && ++k < 27) { nb = __shfl(my_nb, k); ncnt = __shfl(my_cnt, k); }
            }
            have = k < 27;
            // issued after the fences (they drain outstanding loads), consumed a tile later
            if (have && lane < min(64, ncnt - t0)) pre = snap4[nb + t0 + lane];
            float dmin = 3.0e38f, dsum = 0.0f;          // (dsum: a distance that is not a number passes the collision test; fminf drops it)
#pragma unroll 4
            for (int jj = 0; jj < n; jj++) {
                if (c_nb + c_t0 + jj == gi) continue;
                const float4 q = tile[jj];
                if (q.w == 0.0f) {                     // kid (or massless) neighbour: no force term, still a distance
                    const float rx = q.x - me.x, ry = q.y - me.y, rz = q.z - me.z;
                    const float d2 = rx * rx + ry * ry + rz * rz;
                    dmin = fminf(dmin, d2); dsum += d2;
                    continue;
                }
                const float d2 = pair_exact(me.x, me.y, me.z, q, P.eps2, ax, ay, az);
                dmin = fminf(dmin, d2); dsum += d2;
            }
            // rare: someone in this tile is within the collision gate of one of my lanes
            const bool close = scan && (!(dmin > P.coll_d2_gate) || dsum != dsum);
            if (__any(close)) {
                if (close) {
                    for (int jj = 0; jj < n; jj++) {
                        const float4 q = tile[jj];
                        const float rx = q.x - me.x, ry = q.y - me.y, rz = q.z - me.z;
                        const float d2 = rx * rx + ry * ry + rz * rz;
                        const int gj = c_nb + c_t0 + jj;
                        if (!(d2 > P.coll_d2_gate) && gj != gi)
                            flag = max(flag, collide_exact(P, d2, age_i, id_i, snap_age[gj], sorted_id[gj]));
                    }
                }
            }
        }
    }
    if (MODE != 0 && k1 < STENCIL) {             // not the end of the walk: hand the sums on
        handoff_publish(force4 + gi, ax, ay, az, flag, valid, ready, k1);
        PS_TRACE_END();
        return;
    }
    if (ALLP) {                                  // a partial sum: k_allpairs_combine finishes the particle
        if (valid) far.part_acc[(size_t)part * far.part_plane + (size_t)task_no * 64 + lane] = make_float4(ax, ay, az, 0.f);
        return;
    }
    if (dead) flag = 2;
    if (kid) { ax = 0.f; ay = 0.f; az = 0.f; }   // every term is skipped for a kid (app_common.cu:240)
    if (MODE != 0 && !SETTLED) {
        // one-pass lean stage: a particle whose own position is not a number met itself and the kids (stencil_adults;
        // the two-pass stage settles this in k_collide_cell, the generic mode skips both explicitly)
        const bool lost = valid && !kid && !finite3(me.x, me.y, me.z);
        if (__any(lost)) {
            const int adults = stencil_adults(P, i1, i2, i3, cell_start, snap_age);       // (all lanes: the count is a wave's work)
            if (lost && adults <= 1) { ax = 0.f; ay = 0.f; az = 0.f; }
        }
    }
    if (valid) force4[gi] = make_float4(ax, ay, az, __int_as_float(flag));
    PS_TRACE_END();
}

template <int MODE, int NQ, bool ALLP>
__global__ __launch_bounds__(256) void k_pairs(DevParams P, const int *__restrict__ cell_start,
                                               const SnapSoa snap4,
                                               const float *__restrict__ snap_soa,
                                               const float *__restrict__ snap_age,
                                               const int *__restrict__ sorted_id,
                                               const int *__restrict__ task_list,
                                               float4 *__restrict__ force4,
                                               FrameScalars *fs, unsigned long long *trace,
                                               const int *__restrict__ active_list, const int *__restrict__ active_count,
                                               const FarCells far, const float *__restrict__ far_buf,
                                               const int *__restrict__ far_start, const int *__restrict__ far_n)
{
    // Workgroups of four INDEPENDENT waves (no workgroup barrier anywhere): the hardware
    // deals a workgroup's waves over the four SIMDs of its CU and workgroups over the
    // CUs, which keeps even a small share (a few waves per CU) evenly spread.
    __shared__ float4 tiles[MODE == 0 ? 4 : 1][MODE == 0 ? 64 : 1];   // mode 0 only
    const int wave = threadIdx.x >> 6;
    // The work list holds only non-empty (cell, slice) tasks, cell-major.  Workgroups are dealt
    // round-robin over the eight XCDs (b and b + 8 share an L2), so workgroup b takes its four
    // tasks from XCD (b & 7)'s contiguous eighth of the list: neighbouring cells' snapshots then
    // sit in that XCD's L2.
    // (Eighths of equal WORK instead of equal length -- the outer planes of the grid have
    // fewer neighbours, so the two XCDs holding them go idle for the last sixth of the
    // launch -- were tried: the XCDs then finish together, yet the launch was only 1 %
    // shorter and the extra prefix sum cost k_scan 10 us.)
    const int ntask = active_list ? fs->n_tasks2 : fs->n_tasks;
    const int nitem = ALLP ? ntask * ALLP_PARTS : ntask;         // all-pairs: a wave per (task, part)
    const int nwg = (nitem + 3) >> 2;
    if ((int)blockIdx.x >= nwg) return;
    const int slot = xcd_contiguous(blockIdx.x, nwg) * 4 + wave;
    if (slot >= nitem) return;
    const int t = ALLP ? slot / ALLP_PARTS : slot, part = ALLP ? slot - t * ALLP_PARTS : 0;
    // (an all-pairs context always runs the two-pass stage: the flags are settled)
    pairs_task<MODE, NQ, ALLP, ALLP>(P, cell_start, snap4, snap_soa, snap_age, sorted_id, force4,
                                     task_list[t], tiles[MODE == 0 ? wave : 0], trace, active_list, active_count, 0, STENCIL, nullptr, fs, far, part, t,
                                     far_buf, far_start, far_n);
}

// All-pairs: a particle's acceleration = ((stencil chain + part 0's far cells) + part 1) + ... + part 15,
// the same association on one GPU and on any number of ranks.  One thread per (task, lane).
__global__ void k_allpairs_combine(DevParams P, const int *__restrict__ cell_start, const int *__restrict__ task_list,
                                   const int *__restrict__ active_list, const int *__restrict__ active_count,
                                   const FarCells far, float4 *__restrict__ force4, const FrameScalars *__restrict__ fs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x, t = i >> 6, lane = i & 63;
    if (t >= fs->n_tasks2) return;
    const int task = task_list[t], c = task / P.slices, slice = task - c * P.slices;
    const int first = slice * 64, cnt = active_count[c];
    if (first + lane >= cnt) return;
    const int gi = active_list[cell_start[c] + first + lane];
    float4 a = far.part_acc[(size_t)t * 64 + lane];
#pragma unroll
    for (int p = 1; p < ALLP_PARTS; p++) {
        const float4 b = far.part_acc[(size_t)p * far.part_plane + (size_t)t * 64 + lane];
        a.x += b.x; a.y += b.y; a.z += b.z;
    }
    force4[gi] = make_float4(a.x, a.y, a.z, __int_as_float(0));     // (on the active list: flag 0, not a kid)
}

constexpr int MERGE_TILE = 4 * 64 + 4;          // floats per lane group: x[64] y[64] z[64] w[64] + skew

// The same walk for a wave that has its SIMD (almost) to itself -- a slab of a multi-GPU run has
// about 1.5 force tasks per SIMD.  There the scalar-load walk of pairs_task is latency-bound (one
// wave cannot cover its own s_load round trips: 1.6x slower per task, PSAMD_WAVES sweep in
// DESIGN.md), so the bodies come as 64-body tiles instead: one vector load per lane, issued a
// whole tile ahead (vector loads retire in order, so they pipeline), through LDS (SoA, no
// barrier: a wave reads only its own tiles and its LDS operations complete in order), read back
// as broadcast 16-byte rows.  Same arithmetic, same order: short last tiles are padded with
// massless bodies far outside the box (r * 0 = +-0 added to a sum that started at +0 changes
// nothing, as for kids).  Two-pass mode only (flags are settled), lean arithmetic.
//
// A wave serves up to four lane GROUPS, each a run of one cell's particles with its own stencil
// and its own tile (the groups' tiles skewed by 16 bytes onto different banks): one group of up
// to 64 lanes = an ordinary (cell, slice) task; several = the partly filled last slices of up to
// four cells packed into one wave (a cell's list of ~148 particles fills two slices and a third
// of another).  All groups walk stencil step k together, tile by tile, for as many rows as the
// longest of their lists.
struct TileGroups {
    int ng;
    int cell[4], first[4], count[4];      // group g: particles active_list[cell_start[cell] + first ..][0 .. count)
};

// NG: how many groups the code is built for (1: an ordinary task, nothing per-group left in it; 4: a pack)
template <int MODE, int NQ, int NG, bool ONE_T>
__device__ __forceinline__ void pairs_task_tile(const DevParams &P, const int *__restrict__ cell_start,
                                                const SnapSoa snap4, float4 *__restrict__ force4,
                                                const TileGroups &G, float *tile, const int *__restrict__ active_list,
                                                int k0, int k1, int *ready, FrameScalars *fs)
{
    const int lane = threadIdx.x & 63;
    int off[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; g++) off[g + 1] = off[g] + ((g < NG && g < G.ng) ? G.count[g] : 0);
    const int g = NG == 1 ? 0 : (lane >= off[1]) + (lane >= off[2]) + (lane >= off[3]);       // a lane past the last group: 3, invalid
    const bool valid = lane < off[4];
    const int gc = valid ? (g == 0 ? G.cell[0] : g == 1 ? G.cell[1] : g == 2 ? G.cell[2] : G.cell[3]) : G.cell[0];
    const int gf = valid ? (g == 0 ? G.first[0] : g == 1 ? G.first[1] : g == 2 ? G.first[2] : G.first[3]) : G.first[0];
    const int l = valid ? lane - (g == 0 ? off[0] : g == 1 ? off[1] : g == 2 ? off[2] : off[3]) : 0;
    const int gi = active_list[cell_start[gc] + gf + l];
    const float4 me = snap4[gi];
    const float eps2f = (float)P.eps2;
    // neighbour ranges of all groups: entry e = group * 27 + stencil step, held by lane e % 64
    int tab_nb[2] = {0, 0}, tab_cnt[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < (NG == 1 ? 1 : 2); r++) {
        const int e = lane + 64 * r, eg = e / STENCIL, ek = e - eg * STENCIL;
        const int ec = (eg < NG && eg < G.ng) ? (eg == 0 ? G.cell[0] : eg == 1 ? G.cell[1] : eg == 2 ? G.cell[2] : G.cell[3]) : -1;
        if (ec >= 0) {
            int i1, i2, i3;
            cell_coords(P, ec, i1, i2, i3);
            const int nc = local_cell(P, i3 + c_stencil[ek][2], i1 + c_stencil[ek][1], i2 + c_stencil[ek][0]);
            if (nc >= 0) {
                tab_nb[r] = cell_start[nc];
                tab_cnt[r] = min(cell_start[nc + 1] - tab_nb[r], P.max_per_cell);
            }
        }
    }
    float ax = 0.f, ay = 0.f, az = 0.f;
    int flag = 0;
    const PairCtx ctx = {me.x, me.y, me.z, 0.f, 0, gi, false};
    if (k0 > 0 && !handoff_consume(force4 + gi, ax, ay, az, flag, valid, ready, k0)) {
        if (lane == 0) atomicOr(&fs->error, ERR_HANDOFF_TIMEOUT);
    }
    const float far = 1.0e6f;                                       // padding body, mass 0
    const float *tx = tile + (valid ? g : 0) * MERGE_TILE, *ty = tx + 64, *tz = tx + 128, *tw = tx + 192;
    int nbs[4] = {0, 0, 0, 0}, cnts[4] = {0, 0, 0, 0};
    // ranges of stencil step k for every group; returns the longest list
    auto step_ranges = [&](int k) -> int {
        int longest = 0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) {
            const int e = gg * STENCIL + k;
            nbs[gg] = __builtin_amdgcn_readlane(e < 64 ? tab_nb[0] : tab_nb[1], e & 63);
            cnts[gg] = gg < G.ng ? __builtin_amdgcn_readlane(e < 64 ? tab_cnt[0] : tab_cnt[1], e & 63) : 0;
            longest = max(longest, cnts[gg]);
        }
        return longest;
    };
    float4 pre[NG];
    auto fetch = [&](int t0) {                                      // this lane's body of every group's tile at row t0
#pragma unroll
        for (int gg = 0; gg < NG; gg++) {
            pre[gg] = make_float4(far, far, far, 0.f);
            if (gg < G.ng && lane < cnts[gg] - t0) pre[gg] = snap4[nbs[gg] + t0 + lane];
        }
    };
    // first non-empty step from k0 on, its first tiles fetched ahead
    int k = k0, t0 = 0, longest = 0;
    while (k < k1 && (longest = step_ranges(k)) == 0) k++;
    bool have = k < k1;
    if (have) fetch(0);
    while (have) {
        const int n = (min(64, longest - t0) + NQ - 1) & ~(NQ - 1);
        PS_WAVE_SYNC();                               // previous tiles fully consumed
#pragma unroll
        for (int gg = 0; gg < NG; gg++)
            if (gg < G.ng) {
                float *t = tile + gg * MERGE_TILE + lane;
                t[0] = pre[gg].x; t[64] = pre[gg].y; t[128] = pre[gg].z; t[192] = pre[gg].w;
            }
        PS_WAVE_SYNC();
        t0 += 64;                                     // advance to the next non-empty row of tiles
        if (t0 >= longest) {
            t0 = 0; longest = 0; k++;
            while (k < k1 && (longest = step_ranges(k)) == 0) k++;
        }
        have = k < k1;
        // issued after the fences (they drain outstanding loads), consumed a tile later
        if (have) fetch(t0);
        float dmin = 3.0e38f;
        for (int jj = 0; jj < n; jj += NQ) {
            v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], qw[NQ / 2];   // 16-byte LDS reads, NQ is a multiple of 4
#pragma unroll
            for (int i = 0; i < NQ / 2; i += 2) {
                const float4 vx = *reinterpret_cast<const float4 *>(tx + jj + 2 * i);
                const float4 vy = *reinterpret_cast<const float4 *>(ty + jj + 2 * i);
                const float4 vz = *reinterpret_cast<const float4 *>(tz + jj + 2 * i);
                const float4 vw = *reinterpret_cast<const float4 *>(tw + jj + 2 * i);
                qx[i] = v2f{vx.x, vx.y}; qx[i + 1] = v2f{vx.z, vx.w};
                qy[i] = v2f{vy.x, vy.y}; qy[i + 1] = v2f{vy.z, vy.w};
                qz[i] = v2f{vz.x, vz.y}; qz[i + 1] = v2f{vz.z, vz.w};
                qw[i] = v2f{vw.x, vw.y}; qw[i + 1] = v2f{vw.z, vw.w};
            }
            if (MODE == 1)
                pairsN_exact_lean<NQ, ONE_T>(P, ctx, qx, qy, qz, qw, 0, nullptr, nullptr, ax, ay, az, flag);
            else
                dmin = fminf(dmin, pairsN_fast<NQ>(ctx, qx, qy, qz, qw, eps2f, ax, ay, az));
        }
    }
    if (k1 < STENCIL) { handoff_publish(force4 + gi, ax, ay, az, flag, valid, ready, k1); return; }
    if (valid) force4[gi] = make_float4(ax, ay, az, __int_as_float(flag));
}

// The force pass, balanced: `nw` waves (all resident), wave slot s walks the (task, stencil step)
// units from wave_pos[s] up to wave_pos[s + 1] -- the same number of bodies for every wave
// (k_split_tasks).  Most of a wave's share is whole tasks; the task its share ends in is started
// FIRST (steps 0 .. k-1, sums published), then the whole tasks, and LAST the task its share
// begins in is finished from the sums the previous wave slot published at the very start of its
// own work -- so nobody waits in practice, and a particle's sum is still one serial chain of
// fp32 additions in the reference's order.  A share that lies inside one task (few tasks, many
// waves) is one middle piece: consume, walk, publish.
// Wave slots are dealt XCD by XCD like the tasks of k_pairs; k_split_tasks starts every XCD's
// run at a whole task, so the wave that continues a task runs in a workgroup that was
// dispatched no later (block b - 8) or is the same workgroup.
// WALK 0: scalar-load walk, ordinary tasks only (packs, if any, run in k_pairs_merged beside it);
//      1: tile walk for everything, packs of partial slices included (few waves per SIMD);
//      2: scalar-load walk for the ordinary tasks, tile walk for the packs, all in one balanced list.
template <int MODE, int NQ>
__device__ __forceinline__ void merged_pack_task(const DevParams &P, const int *__restrict__ cell_start,
                                                 const SnapSoa snap4,
                                                 const int *__restrict__ active_list,
                                                 const int *__restrict__ active_count,
                                                 const int4 *__restrict__ merged_tasks,
                                                 float4 *__restrict__ force4, int slot, float *tile);

// nmb (WALK 0, a multiple of 8 so that the XCD dealing is undisturbed): the first nmb workgroups of the
// launch serve the merged packs of partly filled slices instead (merged_pack_task) -- dispatched first,
// their waves are the oldest on their SIMDs and are served first, which is what lets these long,
// stall-prone waves finish well inside the pass.  (As a kernel of their own on a second stream they
// needed a head start to get that: forked at the same moment as the balanced pass they ended with it,
// and the stage took 0.1 ms longer.)
template <int MODE, int NQ, int WALK>
__global__ __launch_bounds__(256, WALK == 0 ? PSAMD_BALANCED_WAVES : 4) void k_pairs_balanced(DevParams P, const int *__restrict__ cell_start,
                                                        const SnapSoa snap4,
                                                        const float *__restrict__ snap_soa,
                                                        const float *__restrict__ snap_age,
                                                        const int *__restrict__ sorted_id,
                                                        const int *__restrict__ task_list,
                                                        float4 *__restrict__ force4,
                                                        FrameScalars *fs, unsigned long long *trace,
                                                        const int *__restrict__ active_list, const int *__restrict__ active_count,
                                                        const int *__restrict__ wave_unit, int *__restrict__ task_ready,
                                                        const int4 *__restrict__ merged_tasks, int nmb)
{
    __shared__ __attribute__((aligned(16))) float tiles[4][4 * MERGE_TILE];   // up to four 1-KiB tiles per wave
    const int wave = threadIdx.x >> 6;
    if (WALK == 0 && (int)blockIdx.x < nmb) {
        const int pack = blockIdx.x * 4 + wave;
        if (pack < fs->n_merged) merged_pack_task<MODE, (NQ > 4 ? 4 : NQ)>(P, cell_start, snap4, active_list, active_count, merged_tasks, force4, pack, tiles[wave]);   // (4 bodies per group: the 8-wide form costs this kernel its sixth wave per SIMD)
        return;
    }
    const int slot = xcd_contiguous((int)blockIdx.x - nmb, (int)gridDim.x - nmb) * 4 + wave;
    const int ub = __builtin_amdgcn_readfirstlane(wave_unit[slot]), ue = __builtin_amdgcn_readfirstlane(wave_unit[slot + 1]);
    if (ue <= ub) return;
    const int tb = ub / STENCIL, lb = ub - tb * STENCIL;            // first unit: task tb, step lb
    const int tl = (ue - 1) / STENCIL, le = ue - tl * STENCIL;      // last task tl, its steps [.., le)
    // one call site, so one copy of the walk: the pieces in the order they are done
    const bool single = tb == tl;
    const int has_head = (!single && le < STENCIL) ? 1 : 0, has_tail = (!single && lb > 0) ? 1 : 0;
    const int first_whole = tb + has_tail, last_whole = tl + (has_head ? 0 : 1);      // tasks walked whole: [first, last)
    const int nwhole = single ? 0 : last_whole - first_whole;
    const int pieces = single ? 1 : has_head + nwhole + has_tail;
    for (int i = 0; i < pieces; i++) {
        int t, k0 = 0, k1 = STENCIL;
        if (single) { t = tb; k0 = lb; k1 = le; }
        else if (has_head && i == 0) { t = tl; k1 = le; }                 // the head of the last task first: publish early
        else if (i - has_head < nwhole) t = first_whole + (i - has_head);
        else { t = tb; k0 = lb; }                                         // the tail of the first task last: its head was published long ago
        const int nord = fs->n_tasks2;
        if (WALK == 1 || (WALK == 2 && t >= nord)) {
            // task t: an ordinary (cell, slice) task, or -- past them -- merged pack t - n_tasks2
            TileGroups G;
            if (t < nord) {
                const int task = task_list[t], c = task / P.slices, slice = task - c * P.slices;
                G.ng = 1; G.cell[0] = c; G.first[0] = slice * 64; G.count[0] = min(64, active_count[c] - slice * 64);
                G.cell[1] = G.cell[2] = G.cell[3] = c; G.first[1] = G.first[2] = G.first[3] = 0; G.count[1] = G.count[2] = G.count[3] = 0;
            } else {
                const int4 pk = merged_tasks[t - nord];
                const int cells[4] = {pk.x, pk.y, pk.z, pk.w};
                G.ng = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool on = cells[q] >= 0;
                    G.cell[q] = on ? cells[q] : pk.x;
                    G.first[q] = on ? (active_count[cells[q]] & ~63) : 0;
                    G.count[q] = on ? (active_count[cells[q]] & 63) : 0;
                    if (on) G.ng = q + 1;
                }
            }
            if (t < nord) pairs_task_tile<MODE, NQ, 1, WALK != 1>(P, cell_start, snap4, force4, G, tiles[wave], active_list, k0, k1, task_ready + t, fs);
            else pairs_task_tile<MODE, NQ, 4, WALK != 1>(P, cell_start, snap4, force4, G, tiles[wave], active_list, k0, k1, task_ready + t, fs);
        } else
            pairs_task<MODE, NQ, false, true>(P, cell_start, snap4, snap_soa, snap_age, sorted_id, force4, task_list[t], nullptr, trace,
                                              active_list, active_count, k0, k1, task_ready + t, fs);
    }
}

// Merged task of the two-pass force pass: the partly filled last slices of up to four cells
// share one wave, each cell's particles in their own run of lanes.  Every lane group has its
// own stencil, so the bodies cannot come as scalar operands here: each group's current 64
// bodies sit in its own LDS tile (SoA, the groups' tiles skewed by 16 bytes so that they use
// different banks -- scripts/microbench/lds_groups.hip) and a lane reads its group's tile.
// All groups walk stencil step k together, tile by tile, for as many rows as the longest of
// their lists; shorter lists are padded with massless bodies far outside the box: such a
// row adds r * 0 = +-0 to a sum that started at +0 (bit-identical, as for kids).  Launched
// on its own (different register budget from k_pairs).

template <int MODE, int NQ>
__device__ __forceinline__ void merged_pack_task(const DevParams &P, const int *__restrict__ cell_start,
                                                 const SnapSoa snap4,
                                                 const int *__restrict__ active_list,
                                                 const int *__restrict__ active_count,
                                                 const int4 *__restrict__ merged_tasks,
                                                 float4 *__restrict__ force4, int slot, float *tile)
{
    const int lane = threadIdx.x & 63;
    // (Raising these waves' issue priority -- they run one per SIMD among six of the balanced
    // pass -- was tried: s_setprio(3) ended them 0.6 ms earlier and the
    // balanced pass 0.5 ms later, 2.26 -> 2.48 ms for the stage.)
    const int4 pk = merged_tasks[slot];
    const int cells[4] = {pk.x, pk.y, pk.z, pk.w};
    // lane ranges of the groups
    int off[5] = {0, 0, 0, 0, 0}, ng = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = cells[k] >= 0 ? (active_count[cells[k]] & 63) : 0;
        off[k + 1] = off[k] + r;
        if (cells[k] >= 0) ng = k + 1;
    }
    const int g = (lane >= off[1]) + (lane >= off[2]) + (lane >= off[3]);       // a lane past the last group: 3, invalid
    const bool valid = lane < off[4];
    const int c = valid ? (g == 0 ? cells[0] : g == 1 ? cells[1] : g == 2 ? cells[2] : cells[3]) : cells[0];
    const int l = valid ? lane - (g == 0 ? off[0] : g == 1 ? off[1] : g == 2 ? off[2] : off[3]) : 0;
    const int gi = active_list[cell_start[c] + (active_count[c] & ~63) + l];
    const float4 me = snap4[gi];
    const float eps2f = (float)P.eps2;

    // neighbour ranges of all groups: entry e = group * 27 + stencil step, held by lane e % 64
    int tab_nb[2] = {0, 0}, tab_cnt[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int e = lane + 64 * r, eg = e / 27, ek = e - eg * 27;
        const int ec = eg == 0 ? cells[0] : eg == 1 ? cells[1] : eg == 2 ? cells[2] : eg == 3 ? cells[3] : -1;
        if (ec >= 0) {
            int i1, i2, i3;
            cell_coords(P, ec, i1, i2, i3);
            const int nc = local_cell(P, i3 + c_stencil[ek][2], i1 + c_stencil[ek][1], i2 + c_stencil[ek][0]);
            if (nc >= 0) {
                tab_nb[r] = cell_start[nc];
                tab_cnt[r] = min(cell_start[nc + 1] - tab_nb[r], P.max_per_cell);
            }
        }
    }
    const float *tx = tile + (valid ? g : 0) * MERGE_TILE, *ty = tx + 64, *tz = tx + 128, *tw = tx + 192;
    const float far = 1.0e6f;                                       // padding body, mass 0
    const PairCtx ctx = {me.x, me.y, me.z, 0.f, 0, gi, false};
    float ax = 0.f, ay = 0.f, az = 0.f;
    int flag = 0;
    for (int k = 0; k < 27; k++) {
        int nbs[4], cnts[4], longest = 0;
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            const int e = gg * 27 + k;
            nbs[gg] = __builtin_amdgcn_readlane(e < 64 ? tab_nb[0] : tab_nb[1], e & 63);
            cnts[gg] = gg < ng ? __builtin_amdgcn_readlane(e < 64 ? tab_cnt[0] : tab_cnt[1], e & 63) : 0;
            longest = max(longest, cnts[gg]);
        }
        for (int t0 = 0; t0 < longest; t0 += 64) {
            PS_WAVE_SYNC();                                         // previous tiles fully consumed
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                if (gg < ng) {
                    float4 v = make_float4(far, far, far, 0.f);
                    if (lane < cnts[gg] - t0) v = snap4[nbs[gg] + t0 + lane];
                    float *t = tile + gg * MERGE_TILE + lane;
                    t[0] = v.x; t[64] = v.y; t[128] = v.z; t[192] = v.w;
                }
            }
            PS_WAVE_SYNC();
            const int n = (min(64, longest - t0) + NQ - 1) & ~(NQ - 1);
            float dmin = 3.0e38f;
            for (int jj = 0; jj < n; jj += NQ) {
                v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], qw[NQ / 2];   // 16-byte LDS reads, NQ is a multiple of 4
#pragma unroll
                for (int i = 0; i < NQ / 2; i += 2) {
                    const float4 vx = *reinterpret_cast<const float4 *>(tx + jj + 2 * i);
                    const float4 vy = *reinterpret_cast<const float4 *>(ty + jj + 2 * i);
                    const float4 vz = *reinterpret_cast<const float4 *>(tz + jj + 2 * i);
                    const float4 vw = *reinterpret_cast<const float4 *>(tw + jj + 2 * i);
                    qx[i] = v2f{vx.x, vx.y}; qx[i + 1] = v2f{vx.z, vx.w};
                    qy[i] = v2f{vy.x, vy.y}; qy[i + 1] = v2f{vy.z, vy.w};
                    qz[i] = v2f{vz.x, vz.y}; qz[i + 1] = v2f{vz.z, vz.w};
                    qw[i] = v2f{vw.x, vw.y}; qw[i + 1] = v2f{vw.z, vw.w};
                }
                if (MODE == 1)
                    pairsN_exact_lean<NQ>(P, ctx, qx, qy, qz, qw, 0, nullptr, nullptr, ax, ay, az, flag);
                else
                    dmin = fminf(dmin, pairsN_fast<NQ>(ctx, qx, qy, qz, qw, eps2f, ax, ay, az));
            }
        }
    }
    if (valid) force4[gi] = make_float4(ax, ay, az, 0.f);
}

template <int MODE, int NQ>
__global__ __launch_bounds__(256, 6) void k_pairs_merged(DevParams P, const int *__restrict__ cell_start,
                                                      const SnapSoa snap4,
                                                      const int *__restrict__ active_list,
                                                      const int *__restrict__ active_count,
                                                      const int4 *__restrict__ merged_tasks,
                                                      float4 *__restrict__ force4, const FrameScalars *__restrict__ fs)
{
    __shared__ __attribute__((aligned(16))) float tiles[4][4 * MERGE_TILE];
    const int wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= fs->n_merged) return;
    merged_pack_task<MODE, NQ>(P, cell_start, snap4, active_list, active_count, merged_tasks, force4, slot, tiles[wave]);
}

// ------------------------------------------------------------------ apply
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


// Death, survival, integration, wrap and re-hash for every particle of the frame
// (ps.cpp:1182-1242, 1261-1302), one thread per owned SLOT so that the particle arrays stream
// through coalesced (live slots are dense at the head of every segment); only the
// force record is gathered through the slot's rank in the sorted order.  Lifecycle side
// effects that depend on the reference's serial order (free-slot queues) are emitted
// as (key, arg) queue operations and MoveRec records and replayed afterwards.  A particle
// (or a child) whose new segment belongs to a neighbour rank leaves through the outbox:
// that rank's queue hands out its slot, in the same serial order.
// What one slot's update leaves for the list-writing phase of k_apply.
struct ApplyEmit {
    int id, new_cell, new_rec, old_chunk;
    unsigned bits;          // 1 killed, 2 born, 4 relocate, 8 remote, 16 up
};

// ITEMS slots per thread (item `it` of workgroup b is slot (b * ITEMS + it) * 1024 + tid: coalesced).
// Measured on the full N = 2^20 container: 1 / 2 / 4 slots per thread 44 / 55 / 50 us, and 256-thread
// workgroups 71 us -- neither the per-workgroup list reservation (one same-address atomic each) nor
// the workgroup count is what bounds it; one slot per thread in 1024-thread workgroups stays.
template <int ITEMS>
__global__ __launch_bounds__(1024) void k_apply(DevParams P, SegLayout S, const StepState *__restrict__ stp,
                                                const int *__restrict__ rank_of_slot,
                                                const float4 *__restrict__ force4,
                                                float4 *pos4, float4 *vel4, float4 *acc4,
                                                int *cell_arr, uint8_t *pflags,
                                                const CellInfo *__restrict__ celltab,
                                                uint64_t *op_keys, int *op_args, int ops_cap,
                                                MoveRec *moves, int moves_cap,
                                                Outboxes out,
                                                const int *__restrict__ chunk_count, const uint8_t *__restrict__ chunk_skip,
                                                FrameScalars *fs, DevCounters *ctr)
{
    __shared__ int s_ops, s_moves, s_base_ops, s_base_moves;
    __shared__ unsigned int s_cnt[4];
    if (threadIdx.x == 0) { s_ops = 0; s_moves = 0; s_cnt[0] = s_cnt[1] = s_cnt[2] = s_cnt[3] = 0; }
    const int chunk_over = fs->chunk_over;
    const int step = (P.flags & PSAMD_FLAG_EXPLOSIONS) ? stp->step : 0;       // keys the explosion RNG, nothing else
    int old_cells[ITEMS];
    bool any_active = false;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const int si = (blockIdx.x * ITEMS + it) * 1024 + (int)threadIdx.x;       // storage index of the slot
        int oc = -1;
        if (si < P.slots_total) oc = cell_arr[si];
        if (oc <= -2) { cell_arr[si] = -1; oc = -1; }            // a slot the cell-overflow rule reset this frame (slab encoding)
        // free slots (and the ones the cell-overflow rule just killed) have cell == -1
        bool act = oc >= 0 && oc < P.num_cells_global;
        // a particle past the capacity of its chunk's list is not in calc_forces' loop (k_chunk_cap)
        if (chunk_over && act && chunk_count[celltab[oc].chunk] > P.max_per_chunk && chunk_skip[si]) act = false;
        old_cells[it] = act ? oc : -1;
        any_active |= act;
    }
    if (!__syncthreads_or(any_active)) return;                    // nothing alive in this workgroup

    ApplyEmit em[ITEMS];
    int n_op = 0, n_mv = 0;
    unsigned cnt_moved = 0, cnt_surv = 0, cnt_age = 0, cnt_coll = 0;
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const int si = (blockIdx.x * ITEMS + it) * 1024 + (int)threadIdx.x;
        const int old_cell = old_cells[it];
        const bool active = old_cell >= 0;
        const int id = active ? slot_of_index(P, si) : 0;             // slot == particle id
        const int gi = active ? rank_of_slot[si] : 0;

    int flag = 0, new_cell = 0;
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    // the particle's own state is asked for together with its force record (its address needs
    // nothing but the slot), not after the flag in that record has come back
    float4 p = f, v = f;
    float fert = 0.f;
    if (active) { p = pos4[si]; v = vel4[si]; fert = acc4[si].w; f = force4[gi]; flag = __float_as_int(f.w); }
    const CellInfo old_ci = active ? celltab[old_cell] : CellInfo{0, 1, 0, 0};

    const bool killed = active && flag == 2, survived = active && flag == 1, moved = active && flag == 0;
    bool died_of_age = false, born = false, relocate = false;
    int new_rec = 0;

    if (killed) {                                        // kill, ps.cpp:1211-1235
        died_of_age = v.w > P.life_thr;
        cell_arr[si] = -1; pflags[si] = 0;
        pos4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
        vel4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
        acc4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if (survived) {                               // survive_particle, app.cu:271-283
        vel4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
        acc4[si] = make_float4(0.f, 0.f, 0.f, fert);
        pflags[si] = 0;
    } else if (moved) {
        float axv = f.x, ayv = f.y, azv = f.z;
        const float t = P.t;
        if (P.drag > 0.f) { axv -= P.drag * v.x; ayv -= P.drag * v.y; azv -= P.drag * v.z; }    // not in the reference
        // dx = v*t (fp32) + 0.5*a*t*t (double, left to right), rounded once (ps.cpp:1274-1276)
        float dx = (float)((double)(v.x * t) + ((0.5 * (double)axv) * (double)t) * (double)t);
        float dy = (float)((double)(v.y * t) + ((0.5 * (double)ayv) * (double)t) * (double)t);
        float dz = (float)((double)(v.z * t) + ((0.5 * (double)azv) * (double)t) * (double)t);
        if (P.flags & PSAMD_FLAG_EULER) { dx = v.x * t; dy = v.y * t; dz = v.z * t; }           // not in the reference
        dx = clamp_mag(dx, P.dmax); dy = clamp_mag(dy, P.dmax); dz = clamp_mag(dz, P.dmax);
        float rx = p.x + dx, ry = p.y + dy, rz = p.z + dz;

        // set_pos_t, app.cu:117-158: double floor, periodic wrap one grid length at a time
        const int G = P.G;
        const double cs = P.cell_size;
        // (the reference's host path converts with cvttsd2si: a value that is no number, or out of int's range, comes
        // out as INT_MIN there -- 0 on this hardware -- and the wrap loop below then walks it to (2^31 mod G
        // related) cell indices: 0 for G = 16, 4 for G = 12, 7 for G = 15.  Same arithmetic here.)
        auto to_int = [](double d) { return (d >= -2147483648.0 && d < 2147483648.0) ? (int)d : (int)0x80000000; };
        int i1 = to_int(floor((-1.0 * (double)ry) / cs) + (double)(G / 2));
        int i2 = to_int(floor((1.0 * (double)rx) / cs) + (double)(G / 2));
        int i3 = to_int(floor((-1.0 * (double)rz) / cs) + (double)(G / 2));
        for (int guard = 0; guard < 64 &&
             !((i1 >= 0 && i1 < G) && (i2 >= 0 && i2 < G) && (i3 >= 0 && i3 < G)); guard++) {
            if (!(i1 >= 0 && i1 < G)) { const int o = i1; i1 = (i1 + G) % G; ry = (float)((double)ry + (-1.0 * (double)(i1 - o) * cs)); }
            if (!(i2 >= 0 && i2 < G)) { const int o = i2; i2 = (i2 + G) % G; rx = (float)((double)rx + ((double)(i2 - o) * cs)); }
            if (!(i3 >= 0 && i3 < G)) { const int o = i3; i3 = (i3 + G) % G; rz = (float)((double)rz + (-1.0 * (double)(i3 - o) * cs)); }
        }
        i1 = min(max(i1, 0), G - 1); i2 = min(max(i2, 0), G - 1); i3 = min(max(i3, 0), G - 1); // non-finite input only
        new_cell = i3 * G * G + i1 * G + i2;

        float vx = v.x + axv * t, vy = v.y + ayv * t, vz = v.z + azv * t;   // ps.cpp:1289-1296
        vx = clamp_mag(vx, P.vmax); vy = clamp_mag(vy, P.vmax); vz = clamp_mag(vz, P.vmax);
        const float age = v.w + t;                                         // ps.cpp:1302
        uint8_t pf = pflags[si];
        const CellInfo new_ci = celltab[new_cell];
        new_rec = segment_record(S, new_ci.seg_type, new_ci.seg_tid);

        // explosion, ps.cpp:1306-1333, with a counter-based RNG keyed on (seed, step, id)
        if ((P.flags & PSAMD_FLAG_EXPLOSIONS) && (age >= fert) && !(pf & 1)) {
            const uint64_t h0 = splitmix64(P.seed ^ ((uint64_t)(uint32_t)step << 32) ^ (uint64_t)(uint32_t)id);
            const uint64_t h1 = splitmix64(h0), h2 = splitmix64(h1);
            const int r0 = (int)((double)(h0 >> 11) * (1.0 / 9007199254740992.0) * 100.0) - 50;
            const int r1 = (int)((double)(h1 >> 11) * (1.0 / 9007199254740992.0) * 100.0) - 50;
            const int r2 = (int)((double)(h2 >> 11) * (1.0 / 9007199254740992.0) * 100.0) - 50;
            float ux = (float)r0, uy = (float)r1, uz = (float)r2;
            const float mag = sqrtf((float)((double)(ux * ux) + (double)(uy * uy) + (double)(uz * uz)));
            ux /= mag; uy /= mag; uz /= mag;
            vx = (float)((double)ux * P.expl_speed);
            vy = (float)((double)uy * P.expl_speed);
            vz = (float)((double)uz * P.expl_speed);
            pf |= 1;
            born = true;
        }
        pos4[si] = make_float4(rx, ry, rz, p.w);
        vel4[si] = make_float4(vx, vy, vz, age);
        acc4[si] = make_float4(axv, ayv, azv, fert);
        cell_arr[si] = new_cell;
        pflags[si] = pf;
        // segment change => the particle must move to a slot of the new segment
        // (set_pos_x raises seg_fault, app.cu:178-185; handled at ps.cpp:1335-1374)
        relocate = (new_ci.seg_type != old_ci.seg_type || new_ci.seg_tid != old_ci.seg_tid);
    }

    // Does the new segment's queue live on a neighbour rank?  A step moves a particle by at most
    // CELL_SIZE (MAX_DX), i.e. one cell layer -- or TWO when the rounded sum lands exactly on the
    // far face (a particle one ulp below a face, moved by exactly +CELL_SIZE); the box is periodic,
    // so "above" the top layer is layer 0 on the ring's next rank.  One or two layers up the ring:
    // the record goes up; one or two down: down.  (Every rank computes at least two layers; whether
    // the neighbour really owns the record is checked where the record arrives.)
    const bool remote_ = (born || relocate) && P.world > 1 && !owns_record(P, new_rec);
    const int GG = P.G * P.G;
    const int layers_up = ((new_cell / GG) - (old_cell / GG) + P.G) % P.G;
    bool up_ = remote_ && (layers_up == 1 || layers_up == 2);
    // (A jump of more layers than that -- a particle whose position stopped being a number is filed under cell 0
    // wherever it was -- is routed by who holds the record, below; no route: ERR_FOREIGN_CELL.)
    // Whose queue is it?  The neighbour's in the direction of travel as a rule; the OTHER neighbour's in a
    // ring of two or three (the same rank, or the rank two further round); and when a two-layer jump flies
    // over a rank whose whole state is one layer, the rank beyond it: that record travels in the hop-two
    // outbox, straight to rank +-2 (the reference relocates to any segment, ps.cpp:1335-1374).
    // A record for a rank further away than that -- a particle whose position stopped being a number is filed under
    // one fixed cell wherever it was (see the conversion below) -- goes into the far outbox, which every rank
    // receives (an all-gather in the transfer phase; worlds of four or more with births on).
    bool hop2_ = false, far_ = false;
    if (remote_ && !nbr_owns_record(P, up_ ? 1 : 0, new_rec)) {
        const bool near2 = layers_up == 1 || layers_up == 2 || layers_up >= P.G - 2;
        if (nbr_owns_record(P, up_ ? 0 : 1, new_rec)) up_ = !up_;
        else if (P.xfer2_cap > 0 && (near2 || P.far_cap <= 0)) hop2_ = true;
        else if (P.far_cap > 0) far_ = true;
        else atomicOr(&fs->error, ERR_FOREIGN_CELL);
    }

        em[it].id = id; em[it].new_cell = new_cell; em[it].new_rec = new_rec; em[it].old_chunk = old_ci.chunk;
        em[it].bits = (killed ? 1u : 0u) | (born ? 2u : 0u) | (relocate ? 4u : 0u) | (remote_ ? 8u : 0u) | (up_ ? 16u : 0u) | (hop2_ ? 32u : 0u) | (far_ ? 64u : 0u);
        n_op += (killed ? 1 : 0) + ((born && !remote_) ? 1 : 0) + (relocate ? (remote_ ? 1 : 2) : 0);
        n_mv += (born ? 1 : 0) + (relocate ? 1 : 0);
        cnt_moved += (unsigned)__popcll(__ballot(moved)); cnt_surv += (unsigned)__popcll(__ballot(survived));
        cnt_age += (unsigned)__popcll(__ballot(killed && died_of_age)); cnt_coll += (unsigned)__popcll(__ballot(killed && !died_of_age));
    }

    // Event counters and list space: wave -> workgroup (LDS) -> one global atomic per
    // workgroup.  Queue operations: kill -> insert; birth -> remove; relocation ->
    // remove + insert.  Moves: one record per birth / relocation.  A remove on a neighbour's
    // queue is not a local operation: it travels in the outbox.  A thread's operations are
    // consecutive in the lists (their order there is immaterial: they are bucketed by key).
    const int lane = (int)__lane_id();
    const int op_incl = wave_incl_scan(n_op), mv_incl = wave_incl_scan(n_mv);
    __syncthreads();                                     // s_* zeroed
    int wave_ops = 0, wave_moves = 0;
    if (lane == 63) {
        if (op_incl) wave_ops = atomicAdd(&s_ops, op_incl);
        if (mv_incl) wave_moves = atomicAdd(&s_moves, mv_incl);
        if (cnt_moved) atomicAdd(&s_cnt[0], cnt_moved);
        if (cnt_surv) atomicAdd(&s_cnt[1], cnt_surv);
        if (cnt_age) atomicAdd(&s_cnt[2], cnt_age);
        if (cnt_coll) atomicAdd(&s_cnt[3], cnt_coll);
    }
    wave_ops = __shfl(wave_ops, 63); wave_moves = __shfl(wave_moves, 63);
    __syncthreads();
    if (threadIdx.x == 0) {
        DevCounters *mine = ctr + (blockIdx.x % COUNTER_COPIES);
        if (s_cnt[0]) atomicAdd(&mine->integrated, (unsigned long long)s_cnt[0]);
        if (s_cnt[1]) atomicAdd(&mine->survives, (unsigned long long)s_cnt[1]);
        if (s_cnt[2]) atomicAdd(&mine->deaths_age, (unsigned long long)s_cnt[2]);
        if (s_cnt[3]) atomicAdd(&mine->deaths_collision, (unsigned long long)s_cnt[3]);
        if (s_ops | s_moves) {
            // n_ops (low word) and n_moves (high word) grow with a single 64-bit atomic
            const unsigned long long both = ((unsigned long long)(unsigned)s_moves << 32) | (unsigned)s_ops;
            const unsigned long long old = atomicAdd((unsigned long long *)&fs->n_ops, both);
            s_base_ops = (int)(old & 0xffffffffull); s_base_moves = (int)(old >> 32);
        }
    }
    __syncthreads();
    int k = s_base_ops + wave_ops + op_incl - n_op;
    int m = s_base_moves + wave_moves + mv_incl - n_mv;
    const bool room = k + n_op <= ops_cap && m + n_mv <= moves_cap;
    if (!room && (n_op || n_mv)) atomicOr(&fs->error, ERR_OPS_OVERFLOW);
#pragma unroll
    for (int it = 0; it < ITEMS; it++) {
        const unsigned bits = (room && (n_op || n_mv)) ? em[it].bits : 0u;
        const bool killed = bits & 1u, born = bits & 2u, relocate = bits & 4u, remote = bits & 8u, up = bits & 16u, hop2 = bits & 32u, far = bits & 64u;
        const int id = em[it].id, new_cell = em[it].new_cell;
        const int own_r = segment_record_of_slot(S, id);
        const int box = far ? 4 : (up ? 1 : 0) + (hop2 ? 2 : 0);          // which outbox a departure of this particle goes to
        // Outbox entries are reserved per wave and direction: one atomic on the message's counter for all
        // of a wave's departures (a rank whose layer empties into its neighbour -- the box surface on the
        // last rank -- made tens of thousands of same-address atomics here, one per particle: 80 us).
        int out_base[2] = {0, 0};               // this lane's outbox entry for [0] a relocation, [1] a birth
        if (P.world > 1) {
#pragma unroll
            for (int dir = 0; dir < 5; dir++) {
                if (dir >= 2 && (dir == 4 ? P.far_cap : P.xfer2_cap) <= 0) continue;
                const bool mine = remote && box == dir;
                const int want = mine ? ((born ? 1 : 0) + (relocate ? 1 : 0)) : 0;
                if (__any(want > 0)) {
                    const int incl = wave_incl_scan(want);
                    int base = 0;
                    if (lane == 63) base = atomicAdd(&fs->n_out[dir], incl);
                    base = __shfl(base, 63) + incl - want;
                    if (mine) { out_base[1] = base; out_base[0] = base + (born ? 1 : 0); }      // birth first, as the records are written
                }
            }
        }
        if (!(bits & 7u)) continue;
        const uint64_t key = ((uint64_t)(uint32_t)(em[it].old_chunk + 1) << P.key_chunk_shift) | ((uint64_t)(uint32_t)id << 2);
        const uint64_t own_rec = (uint64_t)(uint32_t)own_r << P.key_rec_shift;
        const uint64_t dst_rec = (uint64_t)(uint32_t)em[it].new_rec << P.key_rec_shift;
        // a departure: reserve its outbox entry and put the key there; k_moves_stage adds the state
        auto depart = [&](int kind, uint64_t sub) -> int {
            const int o = out_base[kind];
            if (o >= (far ? P.far_cap : hop2 ? P.xfer2_cap : P.xfer_cap)) { atomicOr(&fs->error, ERR_HALO_OVERFLOW); return -1; }
            XferRec *x = out.o[box] + o;
            x->key = dst_rec | key | sub; x->new_cell = new_cell; x->kind = kind;
            return o;
        };
        if (killed) { op_keys[k] = own_rec | key | 2ull; op_args[k] = id; k++; }
        if (born) {
            if (remote) moves[m] = {id, depart(1, 0ull), 1 | MOVE_OUT | (up ? MOVE_UP : 0) | (hop2 ? MOVE_HOP2 : 0) | (far ? MOVE_FAR : 0), new_cell};
            else {
                moves[m] = {id, -1, 1, new_cell};
                op_keys[k] = dst_rec | key | 0ull; op_args[k] = m;
                k++;
            }
            m++;
        }
        if (relocate) {
            if (remote) moves[m] = {id, depart(0, 1ull), 0 | MOVE_OUT | (up ? MOVE_UP : 0) | (hop2 ? MOVE_HOP2 : 0) | (far ? MOVE_FAR : 0), new_cell};
            else {
                moves[m] = {id, -1, 0, new_cell};
                op_keys[k] = dst_rec | key | 1ull; op_args[k] = m;
                k++;
            }
            m++;
            op_keys[k] = own_rec | key | 2ull; op_args[k] = id; k++;
        }
    }
}

// ------------------------------------------------------------------ lifecycle replay
// The path for a queue with more operations in one step than k_replay_bucket sorts in LDS (a
// collapsing cloud; 1024 particles per cell): the step's operations arrive sorted by key (record-major,
// rocPRIM radix sort of all keys), one workgroup per queue finds its run by binary search and
// replays it on the circular FIFO as q_insert / q_remove would (app_common.cu:305-376).  Like the
// bucketed replay it does so in CLOSED FORM when prefix sums of the +1 / -1 sequence show that the
// queue neither runs empty nor fills up during the step -- the k-th remove takes logical element k,
// the k-th insert becomes logical element count0 + k -- streaming the run through in chunks of one
// operation per thread (three passes: count and check, removes, queue update); only otherwise one lane
// walks the list (on a copy of the segment in LDS when it fits).  `scratch` (n_ops ints; the unsorted
// argument array, free once the sort has run) holds the run's insert arguments in order.
constexpr int RSORT_THREADS = 1024;
__global__ __launch_bounds__(RSORT_THREADS) void k_replay(DevParams P, int n_ops,
                                                 const uint64_t *__restrict__ keys,
                                                 const int *__restrict__ args, int *__restrict__ scratch,
                                                 QueueInfo *qinfo, int *queue, MoveRec *moves,
                                                 DevCounters *ctr)
{
    __shared__ int window[QUEUE_WINDOW];
    __shared__ int op_arg[REPLAY_CHUNK];
    __shared__ unsigned char op_sub[REPLAY_CHUNK];
    __shared__ int wave_tot[RSORT_THREADS / 64];
    __shared__ int s_carry, s_bad;
    const int rec = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // [lo, hi) = operations whose key carries this record
    const uint64_t klo = (uint64_t)(uint32_t)rec << P.key_rec_shift;
    const uint64_t khi = (uint64_t)(uint32_t)(rec + 1) << P.key_rec_shift;
    int lo = 0, hi = n_ops;
    { int a = 0, b = n_ops; while (a < b) { const int m = (a + b) >> 1; if (keys[m] < klo) a = m + 1; else b = m; } lo = a; }
    { int a = lo, b = n_ops; while (a < b) { const int m = (a + b) >> 1; if (keys[m] < khi) a = m + 1; else b = m; } hi = a; }
    if (hi == lo) return;

    QueueInfo q = qinfo[rec];
    // the queue array is stored like the slots: only the owned segments, back to back
    queue += slot_index(P, q.rloc) - q.rloc;
    const int count0 = q.count, size = q.seg_size;
    unsigned long long lost = 0, reloc = 0, births = 0, births_failed = 0;

    // prefix of (inserts | removes << 16) over one chunk of RSORT_THREADS operations, carried from chunk to chunk
    auto chunk_scan = [&](int c0, int &sub, int &arg, int &ins_b, int &rem_b) {
        const int e = c0 + tid;
        sub = -1; arg = 0;
        if (e < hi) { sub = (int)(keys[e] & 3ull); arg = args[e]; }
        const int v = sub < 0 ? 0 : (sub == 2 ? 1 : (1 << 16));
        const int incl = wave_incl_scan(v);
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int o = s_carry;
        for (int k = 0; k < wv; k++) o += wave_tot[k];
        const int excl = o + incl - v;
        ins_b = excl & 0xffff; rem_b = excl >> 16;
        __syncthreads();
        if (tid == RSORT_THREADS - 1) s_carry = o + incl;
        __syncthreads();
    };
    // (counts per chunk fit 16 bits; the carry is kept as two ints packed the same way only while the run is
    // shorter than 65536 operations of either kind -- longer runs take the serial walk)
    const bool packable = hi - lo < 65536;
    if (tid == 0) { s_carry = 0; s_bad = (count0 <= 0 || !packable) ? 1 : 0; }
    __syncthreads();
    if (packable) {
        for (int c0 = lo; c0 < hi; c0 += RSORT_THREADS) {
            int sub, arg, ins_b, rem_b;
            chunk_scan(c0, sub, arg, ins_b, rem_b);
            if (sub >= 0) {
                const int c = count0 + ins_b - rem_b;
                if (sub == 2) { if (!(c < size)) s_bad = 1; scratch[lo + ins_b] = arg; }
                else if (!(c >= 2)) s_bad = 1;
            }
        }
    }
    __syncthreads();
    const int I = s_carry & 0xffff, R = s_carry >> 16;
    if (!s_bad) {
        int *seg = queue + q.rloc;
        const int F = q.front - q.rloc;                // offset of logical element 0
        __syncthreads();
        if (tid == 0) s_carry = 0;
        __syncthreads();
        for (int c0 = lo; c0 < hi; c0 += RSORT_THREADS) {
            int sub, arg, ins_b, rem_b;
            chunk_scan(c0, sub, arg, ins_b, rem_b);
            if (sub >= 0 && sub != 2) {
                const int item = (rem_b < count0) ? seg[(F + rem_b) % size] : scratch[lo + rem_b - count0];
                moves[arg].dst = item;
                if (sub == 1) reloc++; else births++;
            }
        }
        __syncthreads();
        for (int r = tid; r < R; r += RSORT_THREADS) seg[(F + r) % size] = -1;             // every removed element
        __syncthreads();
        for (int k = tid; k < I; k += RSORT_THREADS)                                        // inserts that stayed
            if (count0 + k >= R) seg[(F + count0 + k) % size] = scratch[lo + k];
        if (tid == 0) {
            q.count = count0 + I - R;
            q.front = q.rloc + (F + R) % size;
            q.rear = q.rloc + (F + count0 + I - 1) % size;
            qinfo[rec] = q;
        }
    } else {
        const bool in_lds = q.seg_size <= QUEUE_WINDOW;
        if (in_lds) for (int e = tid; e < q.seg_size; e += RSORT_THREADS) window[e] = queue[q.rloc + e];
        for (int c0 = lo; c0 < hi; c0 += REPLAY_CHUNK) {
            const int n = min(REPLAY_CHUNK, hi - c0);
            __syncthreads();
            for (int e = tid; e < n; e += RSORT_THREADS) {
                op_arg[e] = args[c0 + e];
                op_sub[e] = (unsigned char)(keys[c0 + e] & 3ull);
            }
            __syncthreads();
            if (tid == 0) {
                for (int e = 0; e < n; e++) {
                    const int sub = op_sub[e], arg = op_arg[e];
                    if (sub == 2) {                                // q_insert(arg)
                        if (q.count == q.seg_size) continue;
                        if (q.count == 0) { q.front = q.rloc; q.rear = q.rloc; }
                        else if (q.rear == q.rloc + q.seg_size - 1) q.rear = q.rloc;
                        else q.rear++;
                        q.count++;
                        if (in_lds) window[q.rear - q.rloc] = arg; else queue[q.rear] = arg;
                    } else {                                       // q_remove -> moves[arg].dst
                        int item = -1;
                        if (q.count > 0) {
                            const int pos = q.front;
                            if (q.count == 1) { q.front = -1; q.rear = -1; }
                            else if (q.front == q.rloc + q.seg_size - 1) q.front = q.rloc;
                            else q.front++;
                            q.count--;
                            if (in_lds) { item = window[pos - q.rloc]; window[pos - q.rloc] = -1; }
                            else { item = queue[pos]; queue[pos] = -1; }
                        }
                        moves[arg].dst = item;
                        if (sub == 1) { if (item >= 0) reloc++; else lost++; }
                        else { if (item >= 0) births++; else births_failed++; }
                    }
                }
            }
        }
        __syncthreads();
        if (in_lds) for (int e = tid; e < q.seg_size; e += RSORT_THREADS) queue[q.rloc + e] = window[e];
        if (tid == 0) qinfo[rec] = q;
    }
    DevCounters *mine = ctr + (blockIdx.x % COUNTER_COPIES);
    if (reloc) atomicAdd(&mine->relocations, reloc);
    if (lost) atomicAdd(&mine->relocations_lost, lost);
    if (births) atomicAdd(&mine->births, births);
    if (births_failed) atomicAdd(&mine->births_failed, births_failed);
}

// ---- fast path: bucket the operations by queue record, then one workgroup per record
// sorts its (<= BUCKET_MAX) operations in LDS and replays them in parallel ----------

// ops per record (rec_count and rec_cursor are zeroed with the frame); n_ops is still on the device at
// this point.  (Counting where the operations are made, inside k_apply, was tried twice: a
// workgroup-wide LDS histogram cost that kernel 21 us -- two more barriers per 1024-thread workgroup --
// and per-wave aggregated global atomics 80 us: the 729 counters share 46 cache lines and same-line
// atomics are served one at a time.  This kernel takes 5 us.)
__global__ __launch_bounds__(1024) void k_ops_hist(const uint64_t *__restrict__ keys, const FrameScalars *fs,
                                                    int ops_cap, int rec_shift, int nrec, int *rec_count)
{
    __shared__ int h[LDS_CELLS];
    const int n = min(fs->n_ops, ops_cap), tid = threadIdx.x;
    if ((long long)blockIdx.x * SLOTS_PER_WG >= n) return;
    const bool lds = nrec <= LDS_CELLS;
    if (lds) { for (int r = tid; r < nrec; r += 1024) h[r] = 0; __syncthreads(); }
    for (long long b0 = (long long)blockIdx.x * SLOTS_PER_WG; b0 < n; b0 += (long long)gridDim.x * SLOTS_PER_WG)
        for (int i = tid; i < SLOTS_PER_WG; i += 1024) {
            const long long e = b0 + i;
            if (e < n) {
                const int r = (int)(keys[e] >> rec_shift);
                if (lds) atomicAdd(&h[r], 1); else atomicAdd(&rec_count[r], 1);
            }
        }
    if (lds) {
        __syncthreads();
        for (int r = tid; r < nrec; r += 1024) if (h[r]) atomicAdd(&rec_count[r], h[r]);
    }
}

// The step's scalars for the host (live count, sticky errors, the sizes of the operation lists): the
// workgroup that settles the last of them, the longest bucket, writes the record straight into the
// host's pinned copy, and the step's number behind it once the record is out -- the host polls that
// word.  (It was a 100-byte device-to-host copy command and an event between this kernel and the replay:
// a launch of its own and an idle gap of ~6 us on the step's critical path.)  Called by all threads of
// one workgroup.
__device__ __forceinline__ void publish_scalars(const FrameScalars *fs, FrameScalars *fs_host, int longest, StepState *st)
{
    constexpr int WORDS = (int)(sizeof(FrameScalars) / sizeof(int)), SKIP = (int)(offsetof(FrameScalars, max_bucket) / sizeof(int)),
                  SEQ = (int)(offsetof(FrameScalars, seq) / sizeof(int));
    static_assert(sizeof(FrameScalars) % sizeof(int) == 0, "copied word by word");
    const int *src = reinterpret_cast<const int *>(fs);
    int *dst = reinterpret_cast<int *>(fs_host);
    for (int i = threadIdx.x; i < WORDS; i += blockDim.x)
        if (i != SEQ) dst[i] = i == SKIP ? longest : src[i];              // (max_bucket is being written by this very workgroup)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // the record's number: one more than the last one this context handed out (the host counts along); and the
        // step this record closes is over as far as its number goes: the next frame's reset makes it step + 1
        const int seq = st->seq + 1;
        st->seq = seq; st->pending = 1;
        __hip_atomic_store(&fs_host->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// exclusive prefix of rec_count and its maximum, for configurations with more queue records than
// k_ops_scatter scans for itself in LDS
__global__ __launch_bounds__(1024) void k_ops_scan(int nrec, const int *__restrict__ rec_count,
                                                    int *__restrict__ rec_start, FrameScalars *fs, FrameScalars *fs_host, StepState *st)
{
    __shared__ int wave_tot[16];
    __shared__ int carry_s, max_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { carry_s = 0; max_s = 0; }
    __syncthreads();
    int mymax = 0;
    for (int base = 0; base < nrec; base += 1024) {
        const int r = base + tid;
        const int v = (r < nrec) ? rec_count[r] : 0;
        mymax = max(mymax, v);
        const int incl = wave_incl_scan(v);
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < wv; k++) woff += wave_tot[k];
        const int excl = carry_s + woff + incl - v;
        if (r < nrec) rec_start[r] = excl;
        __syncthreads();
        if (tid == 1023) carry_s = excl + v;
        __syncthreads();
    }
    atomicMax(&max_s, mymax);
    __syncthreads();
    if (tid == 0) { rec_start[nrec] = carry_s; fs->max_bucket = max_s; }
    publish_scalars(fs, fs_host, max_s, st);
}

// The life-cycle kernels below are launched BEFORE the host has read the step's counts back
// (the grid covers the most the step can have produced): they take the counts from the
// frame scalars themselves, and stand down when a queue's list is too long for the bucketed
// replay -- the host then runs the sort-based path once it has seen the counts.
__device__ __forceinline__ bool lifecycle_deferred(const FrameScalars *fs) { return fs->max_bucket > BUCKET_MAX; }

// Bucket the operations by queue record.  SCAN: every workgroup first works out the buckets' starts
// for itself (an exclusive prefix of rec_count in LDS: a few hundred records) instead of waiting
// for a one-workgroup kernel to do it; workgroup 0 also leaves them in rec_start for the replay and
// publishes the longest bucket.  Grid-stride over the operations: the grid is sized from a bound of
// the live count, whatever the step really produced is covered.
template <bool SCAN>
__global__ __launch_bounds__(1024) void k_ops_scatter(const uint64_t *__restrict__ keys, const int *__restrict__ args,
                                                       FrameScalars *fs, FrameScalars *fs_host, StepState *st, int ops_cap, int rec_shift, int nrec,
                                                       const int *__restrict__ rec_count, int *__restrict__ rec_start,
                                                       int *__restrict__ rec_cursor,
                                                       uint64_t *__restrict__ keys_out, int *__restrict__ args_out)
{
    __shared__ int h[LDS_CELLS];
    __shared__ int s_start[SCAN ? LDS_CELLS + 1 : 1];
    __shared__ int wave_tot[16];
    __shared__ int max_s;
    const int n = min(fs->n_ops, ops_cap);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // (SCAN: the last workgroup -- as a rule one with no operations of its own, the grid is sized from a bound --
    // stays for the scan and hands the step's scalars to the host, see publish_scalars)
    const bool publisher = SCAN && blockIdx.x == gridDim.x - 1;
    if ((long long)blockIdx.x * SLOTS_PER_WG >= n && (!SCAN || (blockIdx.x != 0 && !publisher))) return;
    const int *start = rec_start;
    if (SCAN) {
        if (tid == 0) max_s = 0;
        const int per = (nrec + 1023) / 1024, r0 = min(nrec, tid * per), r1 = min(nrec, r0 + per);
        int mine = 0, mymax = 0;
        for (int r = r0; r < r1; r++) { const int v = rec_count[r]; mine += v; mymax = max(mymax, v); }
        const int incl = wave_incl_scan(mine);
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        if (mymax) atomicMax(&max_s, mymax);
        int run = incl - mine, total = 0;
        for (int k = 0; k < 16; k++) { if (k < wv) run += wave_tot[k]; total += wave_tot[k]; }
        for (int r = r0; r < r1; r++) { s_start[r] = run; run += rec_count[r]; }
        if (tid == 0) s_start[nrec] = total;
        __syncthreads();
        const int longest = max_s;
        if (blockIdx.x == 0) {
            for (int r = tid; r <= nrec; r += 1024) rec_start[r] = s_start[r];
            if (tid == 0) fs->max_bucket = longest;
        }
        if (publisher) publish_scalars(fs, fs_host, longest, st);
        if (longest > BUCKET_MAX) return;                       // (lifecycle_deferred, from this workgroup's own scan)
        start = s_start;
    } else if (lifecycle_deferred(fs)) return;
    const bool lds = nrec <= LDS_CELLS;
    for (long long base = (long long)blockIdx.x * SLOTS_PER_WG; base < n; base += (long long)gridDim.x * SLOTS_PER_WG) {
        int mine[SLOTS_PER_WG / 1024];
        __syncthreads();
        if (lds) { for (int r = tid; r < nrec; r += 1024) h[r] = 0; __syncthreads(); }
#pragma unroll
        for (int i = 0; i < SLOTS_PER_WG / 1024; i++) {
            const long long e = base + i * 1024 + tid;
            mine[i] = (e < n) ? (int)(keys[e] >> rec_shift) : -1;
            if (lds && mine[i] >= 0) atomicAdd(&h[mine[i]], 1);
        }
        if (lds) {
            __syncthreads();
            for (int r = tid; r < nrec; r += 1024) { const int v = h[r]; if (v) h[r] = start[r] + atomicAdd(&rec_cursor[r], v); }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < SLOTS_PER_WG / 1024; i++)
            if (mine[i] >= 0) {
                const long long e = base + i * 1024 + tid;
                const int pos = lds ? atomicAdd(&h[mine[i]], 1) : start[mine[i]] + atomicAdd(&rec_cursor[mine[i]], 1);
                keys_out[pos] = keys[e]; args_out[pos] = args[e];
            }
    }
}

// One workgroup per queue record with at most BUCKET_MAX operations: rank them by key in
// LDS, then replay.  When the queue provably neither runs empty nor fills up during the
// step (prefix sums of +1/-1 over the sorted operations), every operation's effect on
// the circular FIFO has a closed form -- the k-th remove takes logical element k, the
// k-th insert becomes logical element count0 + k -- and all of them are applied at once;
// otherwise one lane walks the list exactly as q_insert / q_remove do.
constexpr int REPLAY_THREADS = 512;

// Relocation phase 1 for move record m, run by the workgroups of the replay launch past the queue
// records (nothing here depends on the replay, so it rides along instead of being two launches).
// One GPU: read the moving particle (copy_particle, ps.cpp:1363) or the parent of a child to be
// born into the staging area, and reset_particle the slot a relocation vacates (ps.cpp:1367).  A
// parent that also relocates this step has two records, written side by side by its k_apply
// thread (birth, then relocation): the relocation's thread stages for both and then resets, the
// birth's thread stands back -- so no record reads a slot another thread zeroes.
// Slab: everything local was staged when the outboxes were closed (k_moves_stage); only the reset is left.
__device__ __forceinline__ void moves_stage_reset(const DevParams &P, int m, MoveRec *moves, const FrameScalars *__restrict__ fs,
                                                  float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                  float4 *stage)
{
    if (lifecycle_deferred(fs)) return;
    const int n = fs->n_moves;
    if (m >= n) return;
    const MoveRec r = moves[m];
    if (r.kind & MOVE_IN) return;                         // arrived from a neighbour: staged on arrival, vacates nothing here
    const int kind = r.kind & 0xff;
    const int si = slot_index(P, r.src);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.world > 1) {
        if (kind == 0) { cell_arr[si] = -1; pflags[si] = 0; pos4[si] = zero; vel4[si] = zero; acc4[si] = zero; }
        return;
    }
    if (kind == 1) {
        if (m + 1 < n) { const MoveRec nx = moves[m + 1]; if (nx.src == r.src && (nx.kind & 0xff) == 0) return; }
        float4 *s = stage + (size_t)3 * m;
        s[0] = pos4[si]; s[1] = vel4[si]; s[2] = acc4[si];
        return;
    }
    const float4 p = pos4[si], v = vel4[si], a = acc4[si];
    float4 *s = stage + (size_t)3 * m;
    s[0] = p; s[1] = v; s[2] = a;
    if (pflags[si]) moves[m].kind = MOVE_PARENT;          // is_parent travels in bit 8
    if (m > 0) {
        const MoveRec pv = moves[m - 1];
        if (pv.src == r.src && (pv.kind & 0xff) == 1) { float4 *b = stage + (size_t)3 * (m - 1); b[0] = p; b[1] = v; b[2] = a; }
    }
    cell_arr[si] = -1; pflags[si] = 0; pos4[si] = zero; vel4[si] = zero; acc4[si] = zero;
}

// CAP: the longest list this instance holds in LDS.  Two instances are launched back to back: CAP = 2048
// (27 KB of LDS: five workgroups per CU, every queue of the usual step at once; counting rank) serves
// the queues with up to 2048 operations, CAP = BUCKET_MAX (104 KB, one workgroup per CU; bitonic
// network) the longer lists -- its workgroups leave at once where there is none.  (One instance sized
// for the longest list ran one workgroup per CU for every queue: 66 us instead of 36 for the usual step.)
template <int CAP>
__device__ __forceinline__ void replay_record(const DevParams &P, const int rec, const int *__restrict__ rec_start,
                                                        const uint64_t *__restrict__ keys,
                                                        const int *__restrict__ args,
                                                        QueueInfo *qinfo, int *queue, MoveRec *moves,
                                                        DevCounters *ctr, const FrameScalars *__restrict__ fs,
                                                        unsigned long long *trace,
                                                        float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                        float4 *stage)
{
#ifdef PSAMD_REPLAY_TRACE
    unsigned long long tk[6]; int ti = 0;
#define RT() do { if (threadIdx.x == 0 && ti < 6) tk[ti++] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RT() do {} while (0)
#endif
    RT();
    // keys + args while sorting; afterwards the same bytes hold ins_arg (closed form) or the
    // copy of the segment the serial walk works on
    constexpr int RANK_MAX = 2048;
    static_assert(CAP == RANK_MAX || CAP == BUCKET_MAX, "two instances: short lists, long lists");
    constexpr int KEY_BYTES = (CAP + 64) * 8, SORT_BYTES = KEY_BYTES + CAP * 4;
    constexpr int RAW_BYTES = SORT_BYTES;
    constexpr int WINDOW_SLOTS = KEY_BYTES / 4;         // largest segment the serial walk copies into the key area (4224 / 16512 slots)
    __shared__ __attribute__((aligned(16))) unsigned char raw[RAW_BYTES];
    uint64_t *kbuf = reinterpret_cast<uint64_t *>(raw);
    int *abuf = reinterpret_cast<int *>(raw + KEY_BYTES);
    int *window = reinterpret_cast<int *>(raw);
    // (the sorted args stay where the sort left them, behind the keys: the segment copy of the serial
    // walk and the insert list of the closed form both fit in the key area in front of them)
    static_assert(CAP * 4 <= KEY_BYTES, "abuf must survive the reuse of the key area");
    __shared__ unsigned char s_sub[CAP];
    constexpr int NT = REPLAY_THREADS;
    __shared__ int wave_tot[NT / 64];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (lifecycle_deferred(fs)) return;
    const int start = rec_start[rec];
    const int n = min(rec_start[rec + 1] - start, BUCKET_MAX);
    if (n == 0 || (CAP == RANK_MAX ? n > RANK_MAX : n <= RANK_MAX)) return;       // (the other instance's)
    QueueInfo q = qinfo[rec];
    const bool in_lds = q.seg_size <= WINDOW_SLOTS;
    queue += slot_index(P, q.rloc) - q.rloc;           // owned segments only, back to back
    // Inside one bucket the record bits of the keys are all the same: what is sorted is (chunk, id, sub) with the
    // operation's place in the bucket packed in below it -- one 8-byte word per operation, its argument fetched
    // through that place once the order is known.  (With the arguments carried along as a second array every
    // exchange moved 24 bytes instead of 16; the sort is bound by LDS bandwidth, five workgroups to a CU.)
    constexpr int IDX_BITS = CAP == RANK_MAX ? 11 : 13;
    static_assert((1 << IDX_BITS) >= CAP, "an operation's place in the bucket must fit");
    const bool packed_keys = P.key_rec_shift + IDX_BITS <= 64;              // (else, a geometry with > 2^51 (chunk, id) pairs: keys and arguments side by side)
    const uint64_t low_mask = P.key_rec_shift >= 64 ? ~0ull : ((1ull << P.key_rec_shift) - 1ull);
    for (int e = tid; e < n; e += NT) {
        const uint64_t k = keys[start + e];
        kbuf[e] = packed_keys ? (((k & low_mask) << IDX_BITS) | (uint64_t)e) : k;
        abuf[e] = args[start + e];
    }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    RT();
    // bitonic sort in LDS, padded to a power of two with +inf keys.  (Ranking by counting --
    // every thread compares its keys with all of them, two per 16-byte broadcast read, no barriers -- was
    // tried for the short lists: LDS-bandwidth-bound, 65 us against the network's 36 for the usual step.)
    int np = 2;
    while (np < n) np <<= 1;
    for (int e = n + tid; e < np; e += NT) { kbuf[e] = ~0ull; abuf[e] = -1; }
    __syncthreads();
    auto sort = [&](auto with_args) {
        for (int k = 2; k <= np; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (np >> 1); t += NT) {
                    // t-th compare-exchange pair of this stage: e has bit j clear
                    const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const int partner = e | j;
                    const uint64_t a = kbuf[e], b = kbuf[partner];
                    const bool up = (e & k) == 0;
                    if ((a > b) == up) {
                        kbuf[e] = b; kbuf[partner] = a;
                        if (decltype(with_args)::value) { const int x = abuf[e]; abuf[e] = abuf[partner]; abuf[partner] = x; }
                    }
                }
                // For j <= 64 both elements of pair p lie in the 128-element chunk p >> 6, and all 64
                // pairs of a chunk belong to one wave (p = t + m * NT, NT a multiple of 64): such
                // stages need no workgroup barrier, only the wave's own order -- 56 of the 66 stages
                // at 2048 operations, and the barriers were what a long list cost.
                const int next_j = j > 1 ? (j >> 1) : k;              // the next k starts at j = k
                if (j > 64 || next_j > 64) __syncthreads();
                else PS_WAVE_SYNC();
            }
    };
    if (packed_keys) sort(std::false_type{}); else sort(std::true_type{});
    __syncthreads();
    if (packed_keys) {
        // the arguments into the order of the keys: through registers, the array is permuted in place
        constexpr int PER = CAP / NT;
        int av[PER];
#pragma unroll
        for (int m = 0; m < PER; m++) {
            const int e = tid + m * NT;
            av[m] = 0;
            if (e < n) {
                const uint64_t k = kbuf[e];
                av[m] = abuf[(int)(k & ((1ull << IDX_BITS) - 1ull))];
                s_sub[e] = (unsigned char)((k >> IDX_BITS) & 3ull);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < PER; m++) { const int e = tid + m * NT; if (e < n) abuf[e] = av[m]; }
    } else {
        for (int e = tid; e < n; e += NT) s_sub[e] = (unsigned char)(kbuf[e] & 3ull);
    }
    __syncthreads();
    const int *s_arg = abuf;
    RT();
    int *ins_arg = (int *)kbuf;                        // keys no longer needed

    // prefix counts of inserts / removes before each of my (up to 8 consecutive) operations
    const int per = (n + NT - 1) / NT, e0 = tid * per, e1 = min(n, e0 + per);
    int my_ins = 0, my_rem = 0;
    for (int e = e0; e < e1; e++) { if (s_sub[e] == 2) my_ins++; else my_rem++; }
    const int packed = my_ins | (my_rem << 16);
    const int incl = wave_incl_scan(packed);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int off = 0, total = 0;
    for (int k = 0; k < NT / 64; k++) { if (k < wv) off += wave_tot[k]; total += wave_tot[k]; }
    const int excl = off + incl - packed;
    int ins_b = excl & 0xffff, rem_b = excl >> 16;
    const int I = total & 0xffff, R = total >> 16;
    const int count0 = q.count, size = q.seg_size;
    {   // would any operation meet an empty or a full queue?
        int ib = ins_b, rb = rem_b;
        bool bad = (count0 <= 0);
        for (int e = e0; e < e1; e++) {
            const int c = count0 + ib - rb;
            if (s_sub[e] == 2) { bad |= !(c < size); ib++; } else { bad |= !(c >= 2); rb++; }
        }
        if (bad) s_bad = 1;
    }
    __syncthreads();
    RT();
    unsigned long long lost = 0, reloc = 0, births = 0, births_failed = 0;
    if (s_bad) {
        // rare (a queue about to run empty or fill up): one lane walks the list exactly as
        // q_insert / q_remove do, on a copy of the segment in LDS when it fits
        if (in_lds) { for (int e = tid; e < q.seg_size; e += NT) window[e] = queue[q.rloc + e]; __syncthreads(); }
        if (tid == 0) {
            for (int e = 0; e < n; e++) {
                const int sub = s_sub[e], arg = s_arg[e];
                if (sub == 2) {                                // q_insert(arg), app_common.cu:346-376
                    if (q.count == q.seg_size) continue;
                    if (q.count == 0) { q.front = q.rloc; q.rear = q.rloc; }
                    else if (q.rear == q.rloc + q.seg_size - 1) q.rear = q.rloc;
                    else q.rear++;
                    q.count++;
                    if (in_lds) window[q.rear - q.rloc] = arg; else queue[q.rear] = arg;
                } else {                                       // q_remove, app_common.cu:305-339
                    int item = -1;
                    if (q.count > 0) {
                        const int pos = q.front;
                        if (q.count == 1) { q.front = -1; q.rear = -1; }
                        else if (q.front == q.rloc + q.seg_size - 1) q.front = q.rloc;
                        else q.front++;
                        q.count--;
                        if (in_lds) { item = window[pos - q.rloc]; window[pos - q.rloc] = -1; }
                        else { item = queue[pos]; queue[pos] = -1; }
                    }
                    moves[arg].dst = item;
                    if (sub == 1) { if (item >= 0) reloc++; else lost++; }
                    else { if (item >= 0) births++; else births_failed++; }
                }
            }
        }
        if (in_lds) { __syncthreads(); for (int e = tid; e < q.seg_size; e += NT) queue[q.rloc + e] = window[e]; }
    } else {
        // closed form, straight on the queue in global memory: only the R + I touched entries move
        int *seg = queue + q.rloc;
        const int F = q.front - q.rloc;                // offset of logical element 0
        for (int e = e0, ib = ins_b; e < e1; e++) if (s_sub[e] == 2) ins_arg[ib++] = s_arg[e];
        __syncthreads();
        for (int e = e0, rb = rem_b; e < e1; e++)
            if (s_sub[e] != 2) {
                const int item = (rb < count0) ? seg[(F + rb) % size] : ins_arg[rb - count0];
                moves[s_arg[e]].dst = item;
                if (s_sub[e] == 1) reloc++; else births++;
                rb++;
            }
        __syncthreads();
        for (int r = tid; r < R; r += NT) seg[(F + r) % size] = -1;             // every removed element
        __syncthreads();
        for (int k = tid; k < I; k += NT)                                        // inserts that stayed
            if (count0 + k >= R) seg[(F + count0 + k) % size] = ins_arg[k];
        if (tid == 0) {
            q.count = count0 + I - R;
            q.front = q.rloc + (F + R) % size;
            q.rear = q.rloc + (F + count0 + I - 1) % size;
        }
    }
    __syncthreads();
    RT();
    if (tid == 0) qinfo[rec] = q;
    DevCounters *mine = ctr + (blockIdx.x % COUNTER_COPIES);
    reloc = (unsigned long long)wave_incl_scan((int)reloc); births = (unsigned long long)wave_incl_scan((int)births);
    lost = (unsigned long long)wave_incl_scan((int)lost); births_failed = (unsigned long long)wave_incl_scan((int)births_failed);
    if (lane == 63) {
        if (reloc) atomicAdd(&mine->relocations, reloc);
        if (births) atomicAdd(&mine->births, births);
        if (lost) atomicAdd(&mine->relocations_lost, lost);
        if (births_failed) atomicAdd(&mine->births_failed, births_failed);
    }
    RT();
#ifdef PSAMD_REPLAY_TRACE
    if (threadIdx.x == 0) { for (int i = 0; i < 6; i++) trace[(size_t)8 * rec + i] = tk[i]; trace[(size_t)8 * rec + 6] = (unsigned long long)n; }
#endif
#undef RT
}

// The instance for the usual lists runs one workgroup per queue record (and the first relocation phase in the
// workgroups past them); the one for long lists is launched every step too, with a few workgroups that leave at
// once unless some queue got more than 2048 operations this step (max_bucket) and otherwise stride over the records.
template <int CAP>
__global__ __launch_bounds__(REPLAY_THREADS) void k_replay_bucket(DevParams P, int nrec, const int *__restrict__ rec_start,
                                                        const uint64_t *__restrict__ keys, const int *__restrict__ args,
                                                        QueueInfo *qinfo, int *queue, MoveRec *moves,
                                                        DevCounters *ctr, const FrameScalars *__restrict__ fs,
                                                        unsigned long long *trace,
                                                        float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                        float4 *stage)
{
    if (CAP == 2048) {
        if ((int)blockIdx.x >= nrec) {
            moves_stage_reset(P, ((int)blockIdx.x - nrec) * REPLAY_THREADS + (int)threadIdx.x, moves, fs, pos4, vel4, acc4, cell_arr, pflags, stage);
            return;
        }
        replay_record<CAP>(P, (int)blockIdx.x, rec_start, keys, args, qinfo, queue, moves, ctr, fs, trace, pos4, vel4, acc4, cell_arr, pflags, stage);
        return;
    }
    if (fs->max_bucket <= 2048) return;
    for (int rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
        replay_record<CAP>(P, rec, rec_start, keys, args, qinfo, queue, moves, ctr, fs, trace, pos4, vel4, acc4, cell_arr, pflags, stage);
        __syncthreads();
    }
}

// Relocation phase 1a: read every moving particle (copy_particle, ps.cpp:1363) and
// every parent of a child to be born.  Read-only on the particle arrays, so a
// parent that also relocates this step is seen intact by both of its records.  A record
// that leaves for a neighbour rank (MOVE_OUT) gets its state written into the outbox entry
// k_apply reserved; one that arrived from a neighbour (MOVE_IN) was staged on arrival.
__global__ void k_moves_stage(DevParams P, MoveRec *moves, int n_host, const FrameScalars *__restrict__ fs,
                              const float4 *pos4, const float4 *vel4, const float4 *acc4,
                              const uint8_t *pflags, float4 *stage, Outboxes out, OutboxMsgs msgs)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < 5 && msgs.m[m]) {          // slab, closing the outboxes: the headers of the relocation messages
        int *h = msgs.m[m];
        h[0] = min(fs->n_out[m], m < 2 ? P.xfer_cap : m < 4 ? P.xfer2_cap : P.far_cap); h[1] = 0; h[2] = fs->error;
        if (m == 4) h[3] = FAR_MAGIC;      // (the receivers take it off again: a far outbox that was not all-gathered this step is noticed)
    }
    if (n_host < 0 && lifecycle_deferred(fs)) return;
    const int n = n_host < 0 ? fs->n_moves : n_host;
    if (m >= n) return;
    const MoveRec r = moves[m];
    if (r.kind & MOVE_IN) return;
    const int si = slot_index(P, r.src);
    if (r.kind & MOVE_OUT) {
        if (r.dst < 0) return;                          // the outbox was full (error already raised)
        XferRec *x = out.o[(r.kind & MOVE_FAR) ? 4 : ((r.kind & MOVE_UP) ? 1 : 0) + ((r.kind & MOVE_HOP2) ? 2 : 0)] + r.dst;
        const float4 p = pos4[si], v = vel4[si], a = acc4[si];
        x->pos[0] = p.x; x->pos[1] = p.y; x->pos[2] = p.z; x->pos[3] = p.w;
        x->vel[0] = v.x; x->vel[1] = v.y; x->vel[2] = v.z; x->vel[3] = v.w;
        x->acc[0] = a.x; x->acc[1] = a.y; x->acc[2] = a.z; x->acc[3] = a.w;
        if ((r.kind & 0xff) == 0 && pflags[si]) x->kind |= MOVE_PARENT;
        return;
    }
    float4 *s = stage + (size_t)3 * m;
    s[0] = pos4[si]; s[1] = vel4[si]; s[2] = acc4[si];
    if (r.kind == 0 && pflags[si]) moves[m].kind = MOVE_PARENT;  // is_parent travels in bit 8
}

// Relocation phase 1b: reset_particle on the vacated slots (ps.cpp:1367).
__global__ void k_moves_reset(DevParams P, const MoveRec *__restrict__ moves, int n_host, const FrameScalars *__restrict__ fs,
                              float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_host < 0 && lifecycle_deferred(fs)) return;
    const int n = n_host < 0 ? fs->n_moves : n_host;
    if (m >= n) return;
    const MoveRec r = moves[m];
    if ((r.kind & 0xff) != 0 || (r.kind & MOVE_IN)) return;     // births and arrivals vacate nothing here
    const int si = slot_index(P, r.src);
    cell_arr[si] = -1; pflags[si] = 0;
    pos4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
    vel4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
    acc4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// Relocation phase 2: drop each particle into the slot the queue replay assigned.
__global__ void k_moves_commit(DevParams P, const StepState *__restrict__ stp, const MoveRec *__restrict__ moves, int n_host, const FrameScalars *__restrict__ fs,
                               float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr,
                               uint8_t *pflags, const float4 *__restrict__ stage)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (n_host < 0 && lifecycle_deferred(fs)) return;
    const int n = n_host < 0 ? fs->n_moves : n_host;
    if (m >= n) return;
    const MoveRec r = moves[m];
    if (r.dst < 0 || (r.kind & MOVE_OUT)) return;
    const float4 *s = stage + (size_t)3 * m;
    const int di = slot_index(P, r.dst);
    if ((r.kind & 0xff) == 0) {
        pos4[di] = s[0]; vel4[di] = s[1]; acc4[di] = s[2];
        cell_arr[di] = r.new_cell;
        pflags[di] = (r.kind & MOVE_PARENT) ? 1 : 0;
    } else {
        // create_particle_s (app.cu:189-208): child at the parent's position, opposite
        // velocity, age 0, fresh fertility age from the counter-based RNG
        const uint64_t h0 = splitmix64(P.seed ^ ((uint64_t)(uint32_t)stp->step << 32) ^ (uint64_t)(uint32_t)r.src);
        const uint64_t h3 = splitmix64(splitmix64(splitmix64(h0)));
        const double u = (double)(h3 >> 11) * (1.0 / 9007199254740992.0);
        const float fert = (float)((double)P.fert_lo + u * (double)(P.fert_hi - P.fert_lo));
        const float4 pp = s[0], pv = s[1];
        pos4[di] = make_float4(pp.x, pp.y, pp.z, P.w_default);
        vel4[di] = make_float4((float)(-1.0 * (double)pv.x), (float)(-1.0 * (double)pv.y),
                                  (float)(-1.0 * (double)pv.z), 0.0f);
        acc4[di] = make_float4(0.f, 0.f, 0.f, fert);
        cell_arr[di] = r.new_cell;
        pflags[di] = 0;
    }
}

// ------------------------------------------------------------------ slab exchange
// Messages are arrays of 32-bit words that start with MSG_HEADER_WORDS ints: [0] cells or
// records carried, [1] bodies carried, [2] sticky error bits of the sender.  Their sizes are
// fixed when the context is created (halo_cap_cell bodies per cell, xfer_cap records), so
// the transport never has to negotiate a length.
//
// Snapshot of own cell layers for a neighbour (its halo layer and the layers it computes for
// this rank): header, one count per cell, then x[], y[], z[], w_eff[], age[], id[] of
// `cap` = cells * halo_cap_cell words each, bodies packed cell-major in list order.
//
// Exclusive prefix of min(count(c0 + j), limit) over j < ncell by one workgroup of 1024
// threads: off[j], off[ncell] = total.  `count` is a callable.
template <typename F>
__device__ __forceinline__ void block_prefix_1024(int ncell, F count, int *__restrict__ off, int *wave_tot, int *carry)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) *carry = 0;
    __syncthreads();
    for (int b = 0; b < ncell; b += 1024) {
        const int j = b + tid;
        const int n = j < ncell ? count(j) : 0;
        const int incl = wave_incl_scan(n);
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int o = *carry;
        for (int k = 0; k < wv; k++) o += wave_tot[k];
        if (j < ncell) off[j] = o + incl - n;
        __syncthreads();
        if (tid == 1023) *carry = o + incl;
        __syncthreads();
    }
    if (tid == 0) off[ncell] = *carry;
    __syncthreads();
}

// (both directions -- the snapshot for the rank below and the one for the rank above -- in one launch: blockIdx.x / a block range selects)
struct HaloOut { int c0, ncell; int *msg; int *pack_off; };
struct HaloOut2 { HaloOut h[2]; int n; };

__global__ __launch_bounds__(1024) void k_halo_prefix_out(DevParams P, HaloOut2 H, const int *__restrict__ cell_start, FrameScalars *fs)
{
    __shared__ int wave_tot[16];
    __shared__ int carry;
    const HaloOut h = H.h[blockIdx.x];
    const int c0 = h.c0, ncell = h.ncell;
    int *msg = h.msg, *pack_off = h.pack_off;
    int *counts = msg + MSG_HEADER_WORDS;
    const int lim = min(P.max_per_cell, P.halo_cap_cell);
    bool over = false;
    block_prefix_1024(ncell, [&](int j) {
        const int n = min(cell_start[c0 + j + 1] - cell_start[c0 + j], P.max_per_cell);
        over |= n > lim;
        counts[j] = min(n, lim);
        return min(n, lim); }, pack_off, wave_tot, &carry);
    if (over) atomicOr(&fs->error, ERR_HALO_OVERFLOW);
    __syncthreads();
    if (threadIdx.x == 0) { msg[0] = ncell; msg[1] = pack_off[ncell]; msg[2] = fs->error; }
}

// one workgroup per cell of the messages; the first one also closes the rank's status record (everything
// the build stage can raise has been raised by now)
__global__ __launch_bounds__(256) void k_halo_bodies_out(DevParams P, HaloOut2 H, const int *__restrict__ cell_start,
                                                          const SnapSoa snap4,
                                                          const float *__restrict__ snap_age, const int *__restrict__ sorted_id,
                                                          int *__restrict__ status_out, const FrameScalars *__restrict__ fs)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && status_out) { status_out[1] = fs->error; status_out[2] = fs->live; }
    int j = blockIdx.x, k = 0;
    if (j >= H.h[0].ncell) { j -= H.h[0].ncell; k = 1; }
    if (k >= H.n) return;
    const HaloOut h = H.h[k];
    const int c0 = h.c0, ncell = h.ncell;
    const int *pack_off = h.pack_off;
    int *msg = h.msg;
    const size_t cap = (size_t)ncell * P.halo_cap_cell;
    float *body = reinterpret_cast<float *>(msg + MSG_HEADER_WORDS + ncell);
    const int src = cell_start[c0 + j], dst = pack_off[j], n = pack_off[j + 1] - dst;
    for (int e = threadIdx.x; e < n; e += 256) {
        const float4 q = snap4[src + e];
        body[dst + e] = q.x; body[cap + dst + e] = q.y; body[2 * cap + dst + e] = q.z; body[3 * cap + dst + e] = q.w;
        body[4 * cap + dst + e] = snap_age[src + e];
        reinterpret_cast<int *>(body)[5 * cap + dst + e] = sorted_id[src + e];
    }
}

// The other end: a message's cells become the local cells of one or two remote regions:
// the first `split` cells those of region r0 (local cells from c0), the rest those of region
// r1 (from c1; absent when split == ncell).  Writes cell_start for those cells and the gap
// cell after each region, and -- when r1 holds lent layers, which this rank computes --
// appends their slices to the collide work list.  One workgroup per message (the one from the rank
// below and the one from the rank above in one launch).
struct HaloIn { int c0, c1, ncell, split, s0, s1, lent; const int *msg; int *unpack_off; };
struct HaloIn2 { HaloIn h[2]; int n; };

__global__ __launch_bounds__(1024) void k_halo_prefix_in(DevParams P, HaloIn2 H, int *__restrict__ cell_start, int *__restrict__ task_list,
                                                          FrameScalars *fs)
{
    __shared__ int wave_tot[16];
    __shared__ int carry, task_s;
    const HaloIn h = H.h[blockIdx.x];
    const int c0 = h.c0, c1 = h.c1, ncell = h.ncell, split = h.split, s0 = h.s0, s1 = h.s1, lent = h.lent;
    const int *msg = h.msg;
    int *unpack_off = h.unpack_off;
    const int tid = threadIdx.x;
    const int *counts = msg + MSG_HEADER_WORDS;
    if (msg[0] != ncell) {                      // not the message this rank was planned to get: leave the regions empty
        if (tid == 0) atomicOr(&fs->error, ERR_SLAB_MISMATCH);
        for (int j = tid; j <= ncell; j += 1024) unpack_off[j] = 0;
        for (int j = tid; j <= split; j += 1024) cell_start[c0 + j] = s0;
        for (int j = tid; j <= ncell - split && split < ncell; j += 1024) cell_start[c1 + j] = s1;
        return;
    }
    if (tid == 0) { task_s = fs->n_tasks; if (msg[2]) atomicOr(&fs->error, msg[2]); }
    const int lim = min(P.max_per_cell, P.halo_cap_cell);
    block_prefix_1024(ncell, [&](int j) { return min(max(counts[j], 0), lim); }, unpack_off, wave_tot, &carry);
    const int nfirst = unpack_off[split], total = unpack_off[ncell];
    for (int j = tid; j < ncell; j += 1024) {
        const int lc = j < split ? c0 + j : c1 + (j - split);
        cell_start[lc] = j < split ? s0 + unpack_off[j] : s1 + unpack_off[j] - nfirst;
        const int n = unpack_off[j + 1] - unpack_off[j];
        if (lent && j >= split && n > 0) {            // the lent cells' slices join the collide work list
            const int ns = (n + 63) >> 6;
            const int t0 = atomicAdd(&task_s, ns);
            for (int sl = 0; sl < ns; sl++) task_list[t0 + sl] = lc * P.slices + sl;
        }
    }
    __syncthreads();
    if (tid == 0) {
        cell_start[c0 + split] = s0 + nfirst;                                // gap cell after the first region
        if (split < ncell) cell_start[c1 + (ncell - split)] = s1 + total - nfirst;   // ... and after the second
        if (lent) { fs->n_tasks = task_s; fs->n_lent = total - nfirst; }
    }
}

__global__ __launch_bounds__(256) void k_halo_bodies_in(DevParams P, HaloIn2 H,
                                                         const int *__restrict__ cell_start,
                                                         float *__restrict__ snap_soa, float *__restrict__ snap_age,
                                                         int *__restrict__ sorted_id, int *__restrict__ snap_cid)
{
    int j = blockIdx.x, k = 0;
    if (j >= H.h[0].ncell) { j -= H.h[0].ncell; k = 1; }
    const HaloIn h = H.h[k];
    const int c0 = h.c0, c1 = h.c1, ncell = h.ncell, split = h.split;
    const int *msg = h.msg, *unpack_off = h.unpack_off;
    const size_t cap = (size_t)ncell * P.halo_cap_cell, sc = (size_t)P.sorted_cap;
    const float *body = reinterpret_cast<const float *>(msg + MSG_HEADER_WORDS + ncell);
    const int lc = j < split ? c0 + j : c1 + (j - split);
    const int src = unpack_off[j], n = unpack_off[j + 1] - src, dst = cell_start[lc];
    for (int e = threadIdx.x; e < n; e += 256) {
        const float x = body[src + e], y = body[cap + src + e], z = body[2 * cap + src + e], w = body[3 * cap + src + e],
                    age = body[4 * cap + src + e];
        const int id = reinterpret_cast<const int *>(body)[5 * cap + src + e];
        snap_soa[dst + e] = x; snap_soa[sc + dst + e] = y; snap_soa[2 * sc + dst + e] = z; snap_soa[3 * sc + dst + e] = w;
        snap_age[dst + e] = age;
        sorted_id[dst + e] = id;
        snap_cid[dst + e] = (!(age < P.kid_thr) && !(age > P.life_thr)) ? id : -1;
    }
}

// Remote cells list their bodies in the halos of the cells around them, like k_sort_cells
// does for the own cells.  One workgroup per remote cell: the local cells [lo[i], hi[i]) of up to three regions.
struct CellRanges3 { int lo[3], hi[3]; };
__global__ __launch_bounds__(256) void k_remote_halo_lists(DevParams P, CellRanges3 R, const int *__restrict__ cell_start,
                                                           const SnapSoa snap4, const int *__restrict__ snap_cid,
                                                           int *__restrict__ halo_count, float *__restrict__ halo_f,
                                                           int *__restrict__ halo_id)
{
    __shared__ int s_halo[27], s_halo_base[27];
    int b = blockIdx.x, c = -1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int n = R.hi[i] - R.lo[i];
        if (c < 0 && n > 0) { if (b < n) c = R.lo[i] + b; else b -= n; }
    }
    if (c < 0) return;
    const int start = cell_start[c], n = min(cell_start[c + 1] - start, P.max_per_cell);
    list_in_neighbour_halos(P, c, start, max(n, 0), snap4, snap_cid, halo_count, halo_f, halo_id, s_halo, s_halo_base, false);
}

// The force records of the lent layers go back as header + float4[bodies], in the order their
// snapshot came.  Sender: the lent region's block of force4 is contiguous, copy it.
__global__ void k_pack_force(DevParams P, const float4 *__restrict__ force4, int *__restrict__ msg, const FrameScalars *__restrict__ fs)
{
    const int n = fs->n_lent;
    float4 *dst = reinterpret_cast<float4 *>(msg + MSG_HEADER_WORDS);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { msg[0] = n; msg[1] = n; msg[2] = fs->error; }
    if (i < n) dst[i] = force4[P.reg_sorted[2] + i];
}

// Receiver (the owner): message cell j0 + b is own local cell lentout_c0 + b; its stored bodies
// sit at pack_off[j0 + b] - pack_off[j0] in the message.  One workgroup per lent-out cell.
__device__ __forceinline__ void unpack_force_block(const DevParams &P, int b, int j0, const int *__restrict__ msg, const int *__restrict__ pack_off,
                                                   const int *__restrict__ cell_start, float4 *__restrict__ force4, FrameScalars *fs)
{
    const int c = P.lentout_c0 + b;
    const int ncell = P.lentout_c1 - P.lentout_c0;
    const float4 *src = reinterpret_cast<const float4 *>(msg + MSG_HEADER_WORDS);
    if (b == 0 && threadIdx.x == 0) {
        if (msg[2]) atomicOr(&fs->error, msg[2]);
        if (msg[1] != pack_off[j0 + ncell] - pack_off[j0]) atomicOr(&fs->error, ERR_SLAB_MISMATCH);
    }
    if (msg[1] != pack_off[j0 + ncell] - pack_off[j0]) return;
    const int rel = pack_off[j0 + b] - pack_off[j0], n = pack_off[j0 + b + 1] - pack_off[j0 + b], dst = cell_start[c];
    for (int e = threadIdx.x; e < n; e += blockDim.x) force4[dst + e] = src[rel + e];
}

// Arrivals: every record a neighbour sent becomes a MoveRec whose state is staged already,
// plus the remove operation on this rank's queue, keyed as the sender keyed it.
// far_stride > 0: msg0 is the all-gathered far outbox, world messages far_stride ints apart; a record is taken by the
// rank that holds its queue and passed over by the others (and the own message holds nothing for oneself).
__global__ void k_inbox_merge(DevParams P, const int *__restrict__ msg0, const int *__restrict__ msg1, int blocks_each, int cap,
                              uint64_t *op_keys, int *op_args, int ops_cap,
                              MoveRec *moves, int moves_cap, float4 *stage, FrameScalars *fs, int far_stride)
{
    // (the message from the rank below and the one from the rank above in one launch)
    const int *msg = far_stride > 0 ? msg0 + (size_t)((int)blockIdx.x / blocks_each) * far_stride : (int)blockIdx.x < blocks_each ? msg0 : msg1;
    if (far_stride > 0) {
        if ((int)blockIdx.x / blocks_each == P.rank) return;
        // every rank's outbox must have arrived THIS step (a caller that does not know the far outbox would lose records
        // silently): the sender's mark is taken off once seen
        if ((int)blockIdx.x % blocks_each == 0 && threadIdx.x == 0) {
            if (msg[3] != FAR_MAGIC) atomicOr(&fs->error, ERR_SLAB_MISMATCH);
            const_cast<int *>(msg)[3] = 0;
        }
    }
    const int n = min(msg[0], cap);
    const XferRec *in = reinterpret_cast<const XferRec *>(msg + MSG_HEADER_WORDS);
    const int i = ((int)blockIdx.x % blocks_each) * blockDim.x + threadIdx.x;
    if (i == 0 && msg[2]) atomicOr(&fs->error, msg[2]);
    if (i >= n) return;
    const XferRec x = in[i];
    // the record names the queue it is for: it must be one of this rank's (anything else would be
    // replayed on a queue array this rank does not hold)
    if (!owns_record(P, (int)(x.key >> P.key_rec_shift))) { if (far_stride <= 0) atomicOr(&fs->error, ERR_SLAB_MISMATCH); return; }
    const unsigned long long old = atomicAdd((unsigned long long *)&fs->n_ops, (1ull << 32) | 1ull);
    const int k = (int)(old & 0xffffffffull), m = (int)(old >> 32);
    if (k >= ops_cap || m >= moves_cap) { atomicOr(&fs->error, ERR_OPS_OVERFLOW); return; }
    const int src = (int)((x.key >> 2) & ((1ull << (P.key_chunk_shift - 2)) - 1ull));     // the sender's slot (parent id for births)
    moves[m] = {src, -1, (x.kind & (0xff | MOVE_PARENT)) | MOVE_IN, x.new_cell};
    float4 *s = stage + (size_t)3 * m;
    s[0] = make_float4(x.pos[0], x.pos[1], x.pos[2], x.pos[3]);
    s[1] = make_float4(x.vel[0], x.vel[1], x.vel[2], x.vel[3]);
    s[2] = make_float4(x.acc[0], x.acc[1], x.acc[2], x.acc[3]);
    op_keys[k] = x.key; op_args[k] = m;
}

// ---- all-pairs forces across ranks (PSAMD_FLAG_ALL_PAIRS, world > 1) ----
// SURVEY 8(e)'s first row, literally: every rank contributes the snapshot of its own cells and an
// all-gather hands every rank all of them, once per step.  A rank's block: 16 header words ([0] own
// cells, [1] bodies, [2] the sender's error bits, [3] its first global cell), allg_cells raw cell
// counts, then the own block of snap_soa as it is -- x, y, z, w_eff planes of allg_cap floats, bodies
// cell-major in list order -- so the far walk reads the gathered buffer in place, in the same order
// a single GPU reads its own snapshot: same order, same bits.
__global__ void k_allg_pack(DevParams P, const int *__restrict__ cell_start, const float *__restrict__ snap_soa,
                            int *__restrict__ msg, FrameScalars *fs)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int nb = cell_start[P.n_own_cells];
    if (i == 0) {
        if (nb > P.allg_cap) atomicOr(&fs->error, ERR_HALO_OVERFLOW);
        msg[0] = P.n_own_cells; msg[1] = min(nb, P.allg_cap); msg[2] = fs->error; msg[3] = P.reg_first[0] * P.G * P.G;
    }
    if (i < P.n_own_cells) msg[MSG_HEADER_WORDS + i] = cell_start[i + 1] - cell_start[i];
    if (i < nb && i < P.allg_cap) {
        float *body = reinterpret_cast<float *>(msg + MSG_HEADER_WORDS + P.allg_cells);
        const size_t sc = (size_t)P.sorted_cap, cap = (size_t)P.allg_cap;
#pragma unroll
        for (int k = 0; k < 4; k++) body[k * cap + i] = snap_soa[k * sc + i];
    }
}

// one workgroup per gathered block: where every global cell's bodies start in the gathered buffer, and how many count
__global__ __launch_bounds__(1024) void k_allg_index(DevParams P, const int *__restrict__ all, int *__restrict__ gstart,
                                                      int *__restrict__ gn, FrameScalars *fs)
{
    __shared__ int wave_tot[16];
    __shared__ int carry;
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int *hdr = all + (size_t)r * P.allg_block;
    const int ncell = hdr[0], first = hdr[3];
    if (ncell < 0 || ncell > P.allg_cells || first < 0 || first + ncell > P.num_cells_global) {
        if (tid == 0) atomicOr(&fs->error, ERR_SLAB_MISMATCH);
        return;
    }
    if (tid == 0) { carry = 0; if (hdr[2]) atomicOr(&fs->error, hdr[2]); }
    __syncthreads();
    const int base = r * P.allg_block + MSG_HEADER_WORDS + P.allg_cells;      // index of the block's x[0] in the gathered buffer
    for (int b = 0; b < ncell; b += 1024) {
        const int j = b + tid;
        const int v = j < ncell ? max(hdr[MSG_HEADER_WORDS + j], 0) : 0;
        const int incl = wave_incl_scan(v);
        if (lane == 63) wave_tot[wv] = incl;
        __syncthreads();
        int o = carry;
        for (int k = 0; k < wv; k++) o += wave_tot[k];
        if (j < ncell) {
            const int at = o + incl - v;
            const bool fits = at + v <= P.allg_cap;
            gstart[first + j] = base + (fits ? at : 0);
            gn[first + j] = fits ? min(v, P.max_per_cell) : 0;
            if (!fits) atomicOr(&fs->error, ERR_SLAB_MISMATCH);
        }
        __syncthreads();
        if (tid == 1023) carry = o + incl;
        __syncthreads();
    }
}

hipError_t launch_allg_pack(hipStream_t st, const DevParams &P, const DeviceState &d, int *msg)
{
    const int n = std::max(P.n_own_cells, P.slots_total);
    if (n <= 0) return hipSuccess;
    k_allg_pack<<<(n + 255) / 256, 256, 0, st>>>(P, d.cell_start, d.snap_soa, msg, d.fs);
    return hipGetLastError();
}

hipError_t launch_allg_index(hipStream_t st, const DevParams &P, const DeviceState &d)
{
    k_allg_index<<<P.world, 1024, 0, st>>>(P, d.allg_in, d.gstart, d.gn, d.fs);
    return hipGetLastError();
}

// ------------------------------------------------------------------ self test
// Compare the hand-written sqrt / reciprocal with the compiler's correctly rounded forms
// on every float whose bit pattern lies in [lo_bits, hi_bits].  out[0..3] = mismatch
// counts of sqrt_rn_short, rcp_rn_newton, their composition (what the pair kernel uses)
// and of the rejected one-transcendental shortcut; out[4] = mismatches of inv_sqrt_guarded that it
// did not report, out[5] = inputs it reported; out[8..15] / out[16..23] = first
// offending inputs of sqrt / composition; out[24], out[25] = cursors.
__global__ void k_selftest_math(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out)
{
    const uint64_t span = (uint64_t)hi_bits - lo_bits + 1;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad[6] = {0, 0, 0, 0, 0, 0};
    for (; i < span; i += stride) {
        const float a = __uint_as_float(lo_bits + (uint32_t)i);
        const float s_ref = sqrtf(a), r_ref = 1.0f / a, c_ref = 1.0f / s_ref;
        if (__float_as_uint(sqrt_rn_short(a)) != __float_as_uint(s_ref)) {
            bad[0]++;
            const unsigned long long k = atomicAdd(&out[25], 1ull);
            if (k < 8) out[8 + k] = __float_as_uint(a);
        }
        if (__float_as_uint(rcp_rn_newton(a)) != __float_as_uint(r_ref)) bad[1]++;
        if (__float_as_uint(inv_sqrt_selected(a)) != __float_as_uint(c_ref)) {
            bad[2]++;
            const unsigned long long k = atomicAdd(&out[24], 1ull);
            if (k < 8) out[16 + k] = __float_as_uint(a);
        }
        if (__float_as_uint(inv_sqrt_one_transcendental(a)) != __float_as_uint(c_ref)) bad[3]++;
        bool tie = false;
        const float gq = inv_sqrt_guarded(a, tie);
        if (tie) bad[5]++;
        else if (__float_as_uint(gq) != __float_as_uint(c_ref)) bad[4]++;
    }
    for (int k = 0; k < 6; k++) if (bad[k]) atomicAdd(&out[k], bad[k]);
}

// out[0] += number of floats x with bits in [lo_bits, hi_bits] for which the fp32 add of
// eps2f differs from the reference's double add rounded to float
__global__ void k_validate_eps(uint32_t lo_bits, uint32_t hi_bits, double eps2, float eps2f, unsigned long long *out)
{
    const uint64_t span = (uint64_t)hi_bits - lo_bits + 1;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (; i < span; i += stride) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        if (__float_as_uint(x + eps2f) != __float_as_uint((float)((double)x + eps2))) bad++;
    }
    if (bad) atomicAdd(out, bad);
}

hipError_t launch_validate_eps(hipStream_t st, uint32_t lo_bits, uint32_t hi_bits, double eps2, float eps2f,
                               unsigned long long *out)
{
    k_validate_eps<<<2048, 256, 0, st>>>(lo_bits, hi_bits, eps2, eps2f, out);
    return hipGetLastError();
}

hipError_t launch_selftest_math(hipStream_t st, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out24)
{
    k_selftest_math<<<4096, 256, 0, st>>>(lo_bits, hi_bits, out24);
    return hipGetLastError();
}

// ------------------------------------------------------------------ launch wrappers
#define PS_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return e_; } while (0)

static inline int blocks_for(size_t n, int threads, int cap = 4096)
{
    size_t b = (n + threads - 1) / threads;
    if (b > (size_t)cap) b = cap;
    return b < 1 ? 1 : (int)b;
}

hipError_t launch_unpack_aos(hipStream_t st, const DevParams &P, const void *aos, int first, int count, float half_box,
                             const DeviceState &d)
{
    if (count <= 0) return hipSuccess;
    k_unpack_aos<<<(count + 255) / 256, 256, 0, st>>>(P, (const uint32_t *)aos, first, count, half_box,
                                                      d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.fs);
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_pack_aos(hipStream_t st, const DevParams &P, void *aos, int first, int count, const DeviceState &d)
{
    if (count <= 0) return hipSuccess;
    k_pack_aos<<<(count + 255) / 256, 256, 0, st>>>(P, (uint32_t *)aos, first, count, d.pos4, d.vel4,
                                                    d.acc4, d.cell, d.pflags, d.celltab);
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_place(hipStream_t st, const DevParams &P, int n, const int *ids, const float4 *p, const float4 *v, const float4 *a,
                        const int *cells, const DeviceState &d)
{
    if (n <= 0) return hipSuccess;
    k_place<<<(n + 255) / 256, 256, 0, st>>>(P, n, ids, p, v, a, cells, d.pos4, d.vel4, d.acc4, d.cell, d.pflags);
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

// init_iframe: zero the per-frame counts (cells, chunks, queue records: one array) and the
// per-frame scalars; the sticky error word survives
__global__ void k_frame_reset(int *frame, size_t n, FrameScalars *fs, StepState *st, int *status_out, int status_table)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) frame[i] = 0;
    if (i == 0) {
        const int err = fs->error;
        *fs = FrameScalars{};
        fs->error = err;
        if (st->pending) { st->step += 1; st->pending = 0; }      // the step whose scalars went out last is over
    }
    if (status_out) {
        if (i < (size_t)MSG_HEADER_WORDS) status_out[i] = 0;
        if (i < (size_t)status_table) status_out[STATUS_CHUNK_OFF + i] = 0;       // the (chunk, type) census
    }
}

hipError_t launch_frame_reset(hipStream_t st, const DeviceState &d, size_t frame_ints, int status_table)
{
    const size_t n = std::max(frame_ints, (size_t)status_table);
    k_frame_reset<<<(unsigned)((n + 1023) / 1024), 1024, 0, st>>>(d.cell_count, frame_ints, d.fs, d.st, d.status_out, d.status_out ? status_table : 0);
    return hipGetLastError();
}

// Run once the status records of all ranks are in, before k_apply.
// Workgroups [0, world), one per rank's record: adopt its error bits (status_error: the OR over ALL
// records, this rank's own included -- the same word on every rank, which is what makes a failure
// collective); the owner of queue record 0 queues the reported cell-overflow kills as the inserts
// build_grid would have made (ps.cpp:1523-1526): key = chunk field 0 | slot | insert, i.e. before
// every calc_forces operation and in slot order.
// Workgroups [world, world + num_chunks), one per chunk: the chunk lists' capacity rule
// (ps.cpp:1502-1508) across ranks.  The chunk's count is the sum of the ranks' parts (also what
// hostGridMax[0] is the maximum of); if it passed the capacity, this rank ranks the particles in its
// own segments of the chunk behind what the census says precedes them in slot order.
__global__ __launch_bounds__(1024) void k_status_merge(DevParams P, const int *__restrict__ status_all, uint64_t *op_keys, int *op_args,
                                                        int ops_cap, int *__restrict__ chunk_count, const int *__restrict__ cell_arr,
                                                        const CellInfo *__restrict__ celltab, const int2 *__restrict__ chunk_segs,
                                                        uint8_t *__restrict__ chunk_skip, FrameScalars *fs,
                                                        int force_j0, const int *__restrict__ force_msg, const int *__restrict__ pack_off,
                                                        const int *__restrict__ cell_start, float4 *__restrict__ force4)
{
    __shared__ int s_before[4];
    // workgroups past the status records and the chunks: the force records of the lent-out layers come home
    // (one workgroup per lent-out cell; same stage, so the same launch)
    if ((int)blockIdx.x >= P.world + P.num_chunks) {
        unpack_force_block(P, (int)blockIdx.x - P.world - P.num_chunks, force_j0, force_msg, pack_off, cell_start, force4, fs);
        return;
    }
    if ((int)blockIdx.x >= P.world) {
        const int ch = (int)blockIdx.x - P.world;
        int tot[4] = {0, 0, 0, 0}, below[4] = {0, 0, 0, 0};
        for (int r = 0; r < P.world; r++) {
            const int *t = status_all + (size_t)r * P.status_words + STATUS_CHUNK_OFF + 4 * ch;
#pragma unroll
            for (int k = 0; k < 4; k++) { const int v = max(t[k], 0); tot[k] += v; if (r < P.rank) below[k] += v; }
        }
        const int total = tot[0] + tot[1] + tot[2] + tot[3];
        if (threadIdx.x == 0) {
            chunk_count[ch] = total;                                   // k_apply tests the chunk's whole count
            atomicMax(&fs->gridmax[0], min(total, P.max_per_chunk));   // hostGridMax[0], ps.cpp:1507
            if (total > P.max_per_chunk) fs->chunk_over = 1;
            s_before[0] = below[0]; s_before[1] = tot[0] + below[1]; s_before[2] = tot[0] + tot[1] + below[2];
            s_before[3] = tot[0] + tot[1] + tot[2] + below[3];
        }
        if (total <= P.max_per_chunk) return;
        __syncthreads();
        chunk_cap_block(P, ch, chunk_count, cell_arr, celltab, chunk_segs, chunk_skip, s_before);
        return;
    }
    const int r = blockIdx.x;
    const int *st = status_all + (size_t)r * P.status_words;
    if (threadIdx.x == 0 && st[1]) { atomicOr(&fs->status_error, st[1]); if (r != P.rank) atomicOr(&fs->error, st[1]); }
    if (r == P.rank || !owns_record(P, 0)) return;
    const int n = min(st[0], STATUS_KILL_CAP);
    for (int e = threadIdx.x; e < n; e += 1024) {
        const int id = st[MSG_HEADER_WORDS + e];
        const int k = atomicAdd(&fs->n_ops, 1);
        if (k < ops_cap) { op_keys[k] = ((uint64_t)(uint32_t)id << 2) | 2ull; op_args[k] = id; }
        else atomicOr(&fs->error, ERR_OPS_OVERFLOW);
    }
}

// force_msg (may be null): the force records of the lent-out layers, unpacked by extra workgroups of the same launch
hipError_t launch_status_merge(hipStream_t st, const DevParams &P, const DeviceState &d, const int *status_all,
                               int force_j0, const int *force_msg, const int *pack_off)
{
    if (!status_all || P.world <= 1) return hipSuccess;
    const int ncell = force_msg ? std::max(0, P.lentout_c1 - P.lentout_c0) : 0;
    k_status_merge<<<P.world + P.num_chunks + ncell, 1024, 0, st>>>(P, status_all, d.op_keys, d.op_args, d.ops_cap, d.chunk_count, d.cell, d.celltab,
                                                                    d.chunk_segs, d.chunk_skip, d.fs, force_j0, force_msg, pack_off, d.cell_start, d.force4);
    return hipGetLastError();
}

hipError_t launch_fill_int(hipStream_t st, int *p, int v, size_t n)
{
    if (n == 0) return hipSuccess;
    k_fill_int<<<blocks_for(n, 256), 256, 0, st>>>(p, v, n);
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_init_tdata(hipStream_t st, const DevParams &P, const DeviceState &d)
{
    if (P.slots_total <= 0) return hipSuccess;
    k_init_tdata<<<(P.slots_total + 255) / 256, 256, 0, st>>>(P, d.tdata);
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

hipError_t launch_build_grid(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t *ev)
{
    const int nwg = std::max(1, (P.slots_total + SLOTS_PER_WG - 1) / SLOTS_PER_WG);
    if (ev) (void)hipEventRecord(ev[0], st);
    k_hist_lds<<<nwg, 1024, 0, st>>>(P, d.cell, d.cell_count, d.fs);
    PS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[1], st);
    k_scan<<<1, 1024, 0, st>>>(P, d.cell_count, d.cell_start, d.cursor, d.task_start, d.task_list, d.chunk_count, d.celltab, d.status_out, d.fs);
    PS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[2], st);
    k_scatter_lds<<<nwg + (P.world == 1 ? P.num_chunks : 0), 1024, 0, st>>>(P, nwg, d.cell, d.cursor, d.sorted_id, d.chunk_count, d.celltab,
                                                                             d.chunk_segs, d.chunk_skip, d.pos4, d.vel4, d.tdata);
    PS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[3], st);
    k_sort_cells<1024><<<P.n_own_cells, 256, 0, st>>>(P, d.cell_start, d.sorted_id, d.pos4, d.vel4, d.acc4, d.cell,
                                               d.pflags, d.snap_soa, d.snap_age, d.tdata, d.rank_of_slot, d.op_keys, d.op_args, d.ops_cap,
                                               P.two_pass ? d.halo_count : nullptr, d.halo_f, d.halo_id, d.snap_cid, d.status_out, d.fs, d.ctr);
    PS_LAUNCH_CHECK();
    k_sort_cells<SORT_MAX><<<std::min(P.n_own_cells, 512), 256, 0, st>>>(P, d.cell_start, d.sorted_id, d.pos4, d.vel4, d.acc4, d.cell,
                                               d.pflags, d.snap_soa, d.snap_age, d.tdata, d.rank_of_slot, d.op_keys, d.op_args, d.ops_cap,
                                               P.two_pass ? d.halo_count : nullptr, d.halo_f, d.halo_id, d.snap_cid, d.status_out, d.fs, d.ctr);
    PS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[4], st);
    return hipSuccess;
}

// ---- slab exchange ----
// the snapshots for the rank below (k = 0) and above (k = 1); ncell[k] == 0: no such message.  Also closes the status record.
hipError_t launch_pack_halos(hipStream_t st, const DevParams &P, const DeviceState &d, const int c0[2], const int ncell[2],
                             int *const msg[2], int *const pack_off[2])
{
    HaloOut2 H{};
    for (int k = 0; k < 2; k++)
        if (ncell[k] > 0) { H.h[H.n] = HaloOut{c0[k], ncell[k], msg[k], pack_off[k]}; H.n++; }
    if (H.n == 1) H.h[1] = HaloOut{0, 0, nullptr, nullptr};
    if (H.n > 0) {
        k_halo_prefix_out<<<H.n, 1024, 0, st>>>(P, H, d.cell_start, d.fs);
        PS_LAUNCH_CHECK();
    }
    const int blocks = std::max(1, H.h[0].ncell + (H.n > 1 ? H.h[1].ncell : 0));
    k_halo_bodies_out<<<blocks, 256, 0, st>>>(P, H, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_age, d.sorted_id, d.status_out, d.fs);
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

// The snapshots that arrived: from the rank below (its cells become region 1, the halo layer, then
// region 2, the lent layers) and from the rank above (region 3).  ncell == 0: no such message.
hipError_t launch_unpack_halos(hipStream_t st, const DevParams &P, const DeviceState &d, int ncell_below, const int *msg_below,
                               int *off_below, int ncell_above, const int *msg_above, int *off_above)
{
    const int GG = P.G * P.G;
    HaloIn2 H{};
    CellRanges3 R{};
    int nr = 0;
    if (ncell_below > 0) {
        const int split = P.reg_layers[1] * GG;
        H.h[H.n++] = HaloIn{P.reg_base[1], P.reg_base[2], ncell_below, split, P.reg_sorted[1], P.reg_sorted[2], P.reg_layers[2] > 0 ? 1 : 0, msg_below, off_below};
        if (split > 0) { R.lo[nr] = P.reg_base[1]; R.hi[nr] = P.reg_base[1] + split; nr++; }
        if (ncell_below > split) { R.lo[nr] = P.reg_base[2]; R.hi[nr] = P.reg_base[2] + ncell_below - split; nr++; }
    }
    if (ncell_above > 0) {
        H.h[H.n++] = HaloIn{P.reg_base[3], 0, ncell_above, ncell_above, P.reg_sorted[3], 0, 0, msg_above, off_above};
        R.lo[nr] = P.reg_base[3]; R.hi[nr] = P.reg_base[3] + ncell_above; nr++;
    }
    if (H.n == 0) return hipSuccess;
    if (H.n == 1) H.h[1] = HaloIn{0, 0, 0, 0, 0, 0, 0, nullptr, nullptr};
    k_halo_prefix_in<<<H.n, 1024, 0, st>>>(P, H, d.cell_start, d.task_list, d.fs);
    PS_LAUNCH_CHECK();
    const int cells = H.h[0].ncell + H.h[1].ncell;
    k_halo_bodies_in<<<cells, 256, 0, st>>>(P, H, d.cell_start, d.snap_soa, d.snap_age, d.sorted_id, d.snap_cid);
    PS_LAUNCH_CHECK();
    if (P.two_pass) {
        k_remote_halo_lists<<<cells, 256, 0, st>>>(P, R, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_cid, d.halo_count, d.halo_f, d.halo_id);
        PS_LAUNCH_CHECK();
    }
    return hipSuccess;
}

// a region that no message fills this frame (e.g. world == 1 never has any): nothing to do, its cells keep zero counts
hipError_t launch_pack_force(hipStream_t st, const DevParams &P, const DeviceState &d, int *msg, int cap_bodies)
{
    if (cap_bodies <= 0) return hipSuccess;
    k_pack_force<<<(cap_bodies + 255) / 256, 256, 0, st>>>(P, d.force4, msg, d.fs);
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

// the inboxes: from the ring neighbours, and (msgs[2], msgs[3]; null where no rank of this world can be flown over) from two ranks away
hipError_t launch_inbox_merge(hipStream_t st, const DevParams &P, const DeviceState &d, const int *const msgs[5])
{
    if (P.xfer_cap <= 0) return hipSuccess;
    const int nb = (P.xfer_cap + 255) / 256;
    k_inbox_merge<<<2 * nb, 256, 0, st>>>(P, msgs[0], msgs[1], nb, P.xfer_cap, d.op_keys, d.op_args, d.ops_cap, d.moves, d.moves_cap, d.stage, d.fs, 0);
    PS_LAUNCH_CHECK();
    if (P.xfer2_cap > 0 && msgs[2] && msgs[3]) {
        const int nb2 = (P.xfer2_cap + 255) / 256;
        k_inbox_merge<<<2 * nb2, 256, 0, st>>>(P, msgs[2], msgs[3], nb2, P.xfer2_cap, d.op_keys, d.op_args, d.ops_cap, d.moves, d.moves_cap, d.stage, d.fs, 0);
        PS_LAUNCH_CHECK();
    }
    if (P.far_cap > 0 && msgs[4]) {          // the all-gathered far outboxes of all ranks
        const int nbf = (P.far_cap + 255) / 256;
        const int stride = MSG_HEADER_WORDS + P.far_cap * (int)(sizeof(XferRec) / sizeof(int));
        k_inbox_merge<<<P.world * nbf, 256, 0, st>>>(P, msgs[4], nullptr, nbf, P.far_cap, d.op_keys, d.op_args, d.ops_cap, d.moves, d.moves_cap, d.stage, d.fs, stride);
        PS_LAUNCH_CHECK();
    }
    return hipSuccess;
}

// How one pass of the pair stage is launched, from the hint of its task count: everything that shapes the
// launches and is not read from device memory by the kernels themselves (what a captured graph is keyed by).
struct PairShape {
    bool two, merge, balanced, tile, packs_in_list;
    int nw;                  // wave slots of the balanced force pass
};

static PairShape pair_shape(const DevParams &P, bool lean, int64_t tasks_hint)
{
    PairShape s{};
    s.two = lean && P.two_pass;
    // leftover slices of several cells in one wave (k_pairs_merged).  A merged wave is long and
    // stalls on its tile loads; a small share (a slab with fewer than ~3 tasks per SIMD) has too
    // little other work to cover that and it becomes the critical path (measured on 1/4 and 1/8
    // shares of the N = 2^20 cloud).
    static const bool merge_off = std::getenv("PSAMD_NO_MERGE") != nullptr;
    static const bool balance_off = std::getenv("PSAMD_NO_BALANCE") != nullptr;
    static const int waves_env = std::getenv("PSAMD_WAVES") ? std::atoi(std::getenv("PSAMD_WAVES")) : 0;
    s.merge = s.two && !merge_off && (P.world == 1 || tasks_hint >= 3000) && !(P.flags & PSAMD_FLAG_ALL_PAIRS);   // (the merged kernel walks the stencil only)
    s.balanced = s.two && !balance_off && !(P.flags & PSAMD_FLAG_ALL_PAIRS);
    // Balanced pass: a fixed number of waves, all resident, each walking the same number of
    // bodies.  At least four per SIMD when there are that many tasks (fewer cannot cover their
    // scalar-load latency: 1024 / 2048 / 4096 / 6144 waves took 3.73 / 2.54 / 2.27 / 2.29 ms on
    // the N = 2^20 cloud), but not more waves than tasks (a task cut in three or more pieces is
    // a chain of waves that wait for each other).
    if (s.balanced) {
        static const int waves_per_simd = std::getenv("PSAMD_WAVES_PER_SIMD") ? std::atoi(std::getenv("PSAMD_WAVES_PER_SIMD")) : PSAMD_BALANCED_WAVES;      // (A/B runs)
        s.nw = 1024 * (int)std::min<int64_t>(waves_per_simd, std::max<int64_t>(1, tasks_hint / 1024));
        if (waves_env >= 32) s.nw = std::min(waves_env & ~31, MAX_PAIR_WAVES);
    }
    // few waves per SIMD (a slab of a multi-GPU run): the scalar-load walk cannot cover its own
    // load latency, bodies come through LDS tiles fetched a tile ahead instead -- and the partly
    // filled last slices are packed into tasks of the same pass
    static const int tile_env = std::getenv("PSAMD_TILE") ? std::atoi(std::getenv("PSAMD_TILE")) : -1;
    static const bool unified_packs = std::getenv("PSAMD_UNIFIED_PACKS") != nullptr;
    static const bool tile_packs = std::getenv("PSAMD_TILE_PACKS") != nullptr;
    s.tile = s.balanced && (tile_env >= 0 ? tile_env != 0 : s.nw <= 2048);
    // The packs of partly filled last slices as tasks of the balanced pass itself (tile walk).
    // Measured (pair stage, N = 2^20): one GPU, 8200 tasks: beside the pass in k_pairs_merged 2.31 ms,
    // in the list 2.40 (one kernel holding both walks needs 99 VGPRs: 4 waves per SIMD, not 6);
    // half the cloud (a slab of two): 1.67 vs 1.40 -- the separate kernel's 512 waves end long after
    // a pass that has only 4 waves per SIMD; an eighth (tile walk): no packs 0.58, packs 0.60 -- a
    // pack's four-group walk costs more than the two tasks it saves.  So: in the list for the slabs
    // that use the scalar walk, beside the pass on one GPU, none with the tile walk.
    s.packs_in_list = s.balanced && !merge_off && !(P.flags & PSAMD_FLAG_ALL_PAIRS) && (s.tile ? tile_packs : (s.merge && (unified_packs || P.world > 1)));
    if (s.packs_in_list) { s.merge = false; s.nw = std::min(s.nw, 4096); }      // (98 VGPRs with the tile walk in: 4 resident waves per SIMD)
    if (s.tile) s.merge = false;                  // no separate merged kernel beside a tile-walk pass
    return s;
}

uint64_t launch_pairs_shape(const DevParams &P, int64_t tasks_hint)
{
    const PairShape s = pair_shape(P, P.lean_math != 0, tasks_hint);
    return (uint64_t)(s.nw / 32) | (s.merge ? 1ull << 10 : 0) | (s.tile ? 1ull << 11 : 0) | (s.packs_in_list ? 1ull << 12 : 0) | (s.balanced ? 1ull << 13 : 0);
}

template <int MODE, int NQ>
static hipError_t launch_pairs_mode(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t ev_force, int64_t tasks_hint, int pass)
{
    const int ncomp = comp_count(P);
    if (ncomp <= 0) return hipSuccess;
    const int tasks = ncomp * P.slices;
    const PairShape shape = pair_shape(P, MODE != 0, tasks_hint);
    const bool two = shape.two, merge = shape.merge, balanced = shape.balanced, tile = shape.tile, packs_in_list = shape.packs_in_list;
    const int nw = shape.nw;
    if (two) {
        // collision flags and the per-cell lists of the particles that need a force, then the plan of the force pass
        if (P.max_per_cell + HALO_CAP / 2 <= 1024)
            k_collide_cell<1024><<<ncomp, 256, 0, st>>>(P, d.cell_start, d.snap_soa, d.snap_age, d.sorted_id, d.snap_cid, d.halo_count, d.halo_f,
                                                        d.halo_id, d.active_list, d.active_count, d.task_cost, d.force4);
        else
            k_collide_cell<2560><<<ncomp, 256, 0, st>>>(P, d.cell_start, d.snap_soa, d.snap_age, d.sorted_id, d.snap_cid, d.halo_count, d.halo_f,
                                                        d.halo_id, d.active_list, d.active_count, d.task_cost, d.force4);
        k_plan_force<<<8, 1024, 0, st>>>(P, balanced ? nw : 0, packs_in_list ? 2 : merge ? 1 : 0, d.cell_start, d.active_count, d.task_cost,
                                         d.task_list2, d.ctask_start, d.cost_start, d.merged_tasks, d.wave_pos, d.fs, d.trace);
        if (balanced) k_resolve_steps<<<(nw + 1 + 3) / 4, 256, 0, st>>>(P, nw, d.cell_start, d.task_list2, d.wave_pos, d.wave_unit);
    }
    if (ev_force) (void)hipEventRecord(ev_force, st);      // timing: the force pass proper starts here
    const int *task_list = two ? d.task_list2 : d.task_list;
    const int *active_list = two ? d.active_list : nullptr, *active_count = two ? d.active_count : nullptr;
    // the hand-off flags are indexed by task number, which starts at 0 in every pass of a frame:
    // each pass has its own block of them (both zeroed with the frame)
    int *task_ready = d.task_ready + (size_t)pass * P.n_local_cells * P.slices;
    if (balanced) {
        constexpr int M = MODE == 0 ? 1 : MODE;
        // the packs of partly filled slices (merge): the first nmb workgroups of the same launch
        const int nmb = merge ? (((ncomp + 3) / 4 + 7) & ~7) : 0;
#define PS_BALANCED(W) k_pairs_balanced<M, NQ, W><<<nmb + nw / 4, 256, 0, st>>>(P, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_soa, d.snap_age, d.sorted_id, task_list, \
                                                                     d.force4, d.fs, d.trace, active_list, active_count, d.wave_unit, task_ready, d.merged_tasks, nmb)
        if (tile) PS_BALANCED(1); else if (packs_in_list) PS_BALANCED(2); else PS_BALANCED(0);
#undef PS_BALANCED
    }
    else {
        FarCells far;
        // where the all-pairs walk finds the cells beyond the stencil: the own snapshot (one GPU: local cell == global cell,
        // lengths from consecutive starts) or the all-gathered snapshot of all ranks with its index by global cell
        const bool gathered = (P.flags & PSAMD_FLAG_ALL_PAIRS) && P.world > 1;
        const float *far_buf = gathered ? reinterpret_cast<const float *>(d.allg_in) : d.snap_soa;
        const int *far_start = gathered ? d.gstart : d.cell_start, *far_n = gathered ? d.gn : nullptr;
        far.plane = gathered ? (unsigned long long)P.allg_cap : (unsigned long long)P.sorted_cap;
        far.part_acc = d.part_acc; far.part_plane = (unsigned long long)d.part_tasks * 64;
        if (MODE != 0 && (P.flags & PSAMD_FLAG_ALL_PAIRS)) {
            // (two == true here: all-pairs contexts are created only with the two-pass pair stage)
            const int items = std::min(tasks, d.part_tasks) * ALLP_PARTS;
            k_pairs<MODE, NQ, MODE != 0><<<(items + 3) / 4, 256, 0, st>>>(P, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_soa, d.snap_age, d.sorted_id, task_list, d.force4,
                                                                        d.fs, d.trace, active_list, active_count, far, far_buf, far_start, far_n);
            k_allpairs_combine<<<(std::min(tasks, d.part_tasks) * 64 + 255) / 256, 256, 0, st>>>(P, d.cell_start, task_list, active_list, active_count, far, d.force4, d.fs);
        } else
            k_pairs<MODE, NQ, false><<<(tasks + 3) / 4, 256, 0, st>>>(P, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_soa, d.snap_age, d.sorted_id, task_list, d.force4,
                                                                      d.fs, d.trace, active_list, active_count, far, nullptr, nullptr, nullptr);
        // (unbalanced pass, A/B runs only: the packs as a kernel of their own behind it)
        if (merge) k_pairs_merged<MODE == 0 ? 1 : MODE, NQ><<<(ncomp + 3) / 4, 256, 0, st>>>(
                P, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.active_list, d.active_count, d.merged_tasks, d.force4, d.fs);
    }
    return hipGetLastError();
}

hipError_t launch_pairs(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t ev_force, int64_t tasks_hint, int pass)
{
    // fast math shares the lean modes' validity range (finite 1/sqrt(eps2^3))
    static const int fast_nq = std::getenv("PSAMD_FAST_NQ") ? std::atoi(std::getenv("PSAMD_FAST_NQ")) : 8;      // (A/B runs)
    if ((P.flags & PSAMD_FLAG_FAST_MATH) && P.lean_math)
        return fast_nq == 4 ? launch_pairs_mode<2, 4>(st, P, d, ev_force, tasks_hint, pass) : launch_pairs_mode<2, 8>(st, P, d, ev_force, tasks_hint, pass);
    // 8 pairs per slow-branch test: measured 3 % (full GPU) to 5 % (a 1/8 share) faster than 4
    if (P.lean_math) return launch_pairs_mode<1, 8>(st, P, d, ev_force, tasks_hint, pass);
    return launch_pairs_mode<0, 4>(st, P, d, ev_force, tasks_hint, pass);
}

hipError_t launch_apply(hipStream_t st, const DevParams &P, const SegLayout &S, const DeviceState &d)
{
    if (P.slots_total <= 0) return hipSuccess;
    // slots per thread: one (PSAMD_APPLY_ITEMS: the measurement quoted at the kernel)
#define PS_APPLY(I) k_apply<I><<<(P.slots_total + I * 1024 - 1) / (I * 1024), 1024, 0, st>>>(P, S, d.st, d.rank_of_slot, d.force4, d.pos4, \
        d.vel4, d.acc4, d.cell, d.pflags, d.celltab, d.op_keys, d.op_args, d.ops_cap, \
        d.moves, d.moves_cap, Outboxes{{d.xfer_out[0], d.xfer_out[1], d.xfer_out[2], d.xfer_out[3], d.xfer_out[4]}}, d.chunk_count, d.chunk_skip, d.fs, d.ctr)
    static const int items_env = std::getenv("PSAMD_APPLY_ITEMS") ? std::atoi(std::getenv("PSAMD_APPLY_ITEMS")) : 0;
    const int items = items_env ? items_env : 1;
    if (items >= 4) PS_APPLY(4); else if (items >= 2) PS_APPLY(2); else PS_APPLY(1);
#undef PS_APPLY
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

// slab mode, right after apply: the state of the departing particles goes into the outboxes
// (the rest of the staging waits for the queue replay) and the two messages get their headers
hipError_t launch_outbox_close(hipStream_t st, const DevParams &P, const DeviceState &d, int64_t live_bound, int *const msgs[5])
{
    const int64_t max_moves = std::min<int64_t>(d.moves_cap, 2 * live_bound);
    const int nb = std::max(1, (int)((max_moves + 255) / 256));
    k_moves_stage<<<nb, 256, 0, st>>>(P, d.moves, -2, d.fs, d.pos4, d.vel4, d.acc4, d.pflags, d.stage,
                                      Outboxes{{d.xfer_out[0], d.xfer_out[1], d.xfer_out[2], d.xfer_out[3], d.xfer_out[4]}}, OutboxMsgs{{msgs[0], msgs[1], msgs[2], msgs[3], msgs[4]}});
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

// Usual case, enqueued without waiting for the host: every queue's operations fit one
// workgroup's LDS.  Four launches: the census of the operations per queue record, their bucketing (each
// workgroup scanning the census for itself; after it the frame scalars are complete, longest bucket included -- the host reads them
// back at that point), replay the queues with the first relocation phase riding along, commit.
// `live_bound` >= live particles of the step (arrivals from the neighbour ranks included): at most 3
// queue operations and 2 move records each.
hipError_t launch_ops_bucket(hipStream_t st, const DevParams &P, const DeviceState &d, int nrec, int64_t live_bound)
{
    const int64_t max_ops = std::max<int64_t>(1, std::min<int64_t>(d.ops_cap, 3 * live_bound));
    const int nwg = (int)std::min<int64_t>((max_ops + SLOTS_PER_WG - 1) / SLOTS_PER_WG, 2048);    // (grid-stride beyond)
    k_ops_hist<<<std::min(nwg, 512), 1024, 0, st>>>(d.op_keys, d.fs, d.ops_cap, P.key_rec_shift, nrec, d.rec_count);
    PS_LAUNCH_CHECK();
    if (nrec <= LDS_CELLS)
        k_ops_scatter<true><<<nwg, 1024, 0, st>>>(d.op_keys, d.op_args, d.fs, d.fs_host, d.st, d.ops_cap, P.key_rec_shift, nrec, d.rec_count, d.rec_start,
                                                  d.rec_cursor, d.op_keys_sorted, d.op_args_sorted);
    else {
        k_ops_scan<<<1, 1024, 0, st>>>(nrec, d.rec_count, d.rec_start, d.fs, d.fs_host, d.st);
        PS_LAUNCH_CHECK();
        k_ops_scatter<false><<<nwg, 1024, 0, st>>>(d.op_keys, d.op_args, d.fs, nullptr, nullptr, d.ops_cap, P.key_rec_shift, nrec, d.rec_count, d.rec_start,
                                                   d.rec_cursor, d.op_keys_sorted, d.op_args_sorted);
    }
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

// part 0: the replay of the usual lists (with the first relocation phase), enqueued without waiting for the host;
// part 1, once the host has the step's scalars (they are out before part 0 starts running): the instance for
// long lists only if some queue got more than 2048 operations (`long_lists`), and the commit.  (The long-list
// instance used to be launched every step and leave at once: ~4.5 us on the timeline for nothing.)
hipError_t launch_lifecycle(hipStream_t st, const DevParams &P, const DeviceState &d, int nrec, int64_t live_bound, int part, bool long_lists)
{
    const int64_t max_moves = std::min<int64_t>(d.moves_cap, 2 * live_bound);
    const int nb = (int)((max_moves + REPLAY_THREADS - 1) / REPLAY_THREADS);
    if (part == 0) {
        k_replay_bucket<2048><<<nrec + nb, REPLAY_THREADS, 0, st>>>(P, nrec, d.rec_start, d.op_keys_sorted, d.op_args_sorted, d.qinfo, d.queue,
                                              d.moves, d.ctr, d.fs, d.trace, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage);
        PS_LAUNCH_CHECK();
        return hipSuccess;
    }
    if (long_lists) {
        k_replay_bucket<BUCKET_MAX><<<std::min(nrec, 256), REPLAY_THREADS, 0, st>>>(P, nrec, d.rec_start, d.op_keys_sorted, d.op_args_sorted, d.qinfo, d.queue,
                                              d.moves, d.ctr, d.fs, d.trace, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage);
        PS_LAUNCH_CHECK();
    }
    if (nb > 0) {
        k_moves_commit<<<(int)((max_moves + 255) / 256), 256, 0, st>>>(P, d.st, d.moves, -1, d.fs, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage);
        PS_LAUNCH_CHECK();
    }
    return hipSuccess;
}

// A queue with a very long list (e.g. record 0 during a collapse; the kernels above stood
// down): global sort + serial walk, sized by the counts the host has read back.
hipError_t launch_lifecycle_sorted(hipStream_t st, const DevParams &P, const DeviceState &d, int nrec,
                                   int n_ops, int n_moves)
{
    if (n_ops > 0) {
        hipError_t e = sort_ops(st, d, n_ops, P.key_bits);
        if (e != hipSuccess) return e;
        k_replay<<<nrec, RSORT_THREADS, 0, st>>>(P, n_ops, d.op_keys_sorted, d.op_args_sorted, d.op_args, d.qinfo, d.queue, d.moves, d.ctr);
        PS_LAUNCH_CHECK();
    }
    if (n_moves > 0) {
        const int nb = (n_moves + 255) / 256;
        k_moves_stage<<<nb, 256, 0, st>>>(P, d.moves, n_moves, d.fs, d.pos4, d.vel4, d.acc4, d.pflags, d.stage,
                                          Outboxes{{d.xfer_out[0], d.xfer_out[1], d.xfer_out[2], d.xfer_out[3], d.xfer_out[4]}}, OutboxMsgs{{nullptr, nullptr, nullptr, nullptr}});
        PS_LAUNCH_CHECK();
        k_moves_reset<<<nb, 256, 0, st>>>(P, d.moves, n_moves, d.fs, d.pos4, d.vel4, d.acc4, d.cell, d.pflags);
        PS_LAUNCH_CHECK();
        k_moves_commit<<<nb, 256, 0, st>>>(P, d.st, d.moves, n_moves, d.fs, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage);
        PS_LAUNCH_CHECK();
    }
    return hipSuccess;
}

}  // namespace psamd
