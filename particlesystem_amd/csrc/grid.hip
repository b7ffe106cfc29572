// grid.hip -- AoS <-> SoA, init_iframe and build_grid (ps.cpp:1574-1606, 1468-1537); see kernels_common.hpp for the map of the stage files
#include "kernels_common.hpp"

namespace psamd {

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
// Same two kernels with a workgroup-private histogram in LDS: a workgroup owns SLOTS_PER_WG
// consecutive slots, which by the container's construction belong to one or two segments,
// i.e. a handful of cells a few grid planes apart, so almost all atomics stay in LDS and only
// the touched bins go to memory.  The LDS histogram is a WINDOW of LDS_CELLS cells starting at
// the smallest cell the workgroup meets (grids of up to LDS_CELLS cells: the whole grid); the
// rare slot outside it (a workgroup straddling two distant segments) goes to memory directly.
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

constexpr int SCAN_LDS_CHUNKS = 4096;

__global__ __launch_bounds__(1024) void k_hist_lds(DevParams P, const int *__restrict__ cell, int *__restrict__ cell_count,
                                                    FrameScalars *fs, StepState *st)
{
    __shared__ int h[LDS_CELLS];
    __shared__ int s_min;
    const int tid = threadIdx.x, base = blockIdx.x * SLOTS_PER_WG, ncell = P.n_own_cells;
    // the build's first kernel: the step whose scalars went out last is over (nothing reads the step's number before k_apply)
    if (blockIdx.x == 0 && tid == 0 && st->pending) { st->step += 1; st->pending = 0; }
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
// Workgroups [0, nwg): the scatter.  Workgroups [nwg, nwg + num_chunks), one GPU only: the chunk
// lists' capacity rule for chunk blockIdx.x - nwg (chunk_cap_block; idle unless the chunk is over).
// ROWS: the scatter also writes the reference's T_DATA rows (id, x, y, z, w, age; ps.cpp:1495-1500) -- here a
// workgroup's live slots are consecutive, so are their 24-byte rows.  The rows are a MIRROR for callers that fetch the
// reference's buffer (psamd_download_tdata; psamd_set_tdata_mirror): nothing in the step reads them -- the snapshot the
// pair stage reads is gathered from the particle arrays by k_sort_cells, with or without the mirror -- and a host that
// never fetches T_DATA switches them off (24 B written and 32 B read per particle and step for nothing).
template <bool ROWS>
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
            if (ROWS) {
                const float4 p = pos4[si];
                const float age = vel4[si].w;
                uint2 *t = reinterpret_cast<uint2 *>(tdata + (size_t)6 * si);
                t[0] = make_uint2((uint32_t)id, __float_as_uint(p.x));
                t[1] = make_uint2(__float_as_uint(p.y), __float_as_uint(p.z));
                t[2] = make_uint2(__float_as_uint(p.w), __float_as_uint(age));
            }
        }
}

// One workgroup: exclusive prefix of the own cells' counts, the scatter cursors, the chunk
// totals and hostGridMax (ps.cpp:1504-1516: maxima are of stored entries, so capped).
// (Round 5 tried this as the histogram launch's last-arriving workgroup -- a ticket, no launch of its own: 28.5 us for the
// pair instead of 7.4 + 12.1.  The counts are written by other workgroups' device-scope atomics in the same launch, so
// the scan had to read them with agent-scope loads -- every one a trip past the XCD's L2 -- and a device-scope FENCE on
// this chip writes back and invalidates an L2: 733 workgroups fencing made the histogram 250 us.  A kernel boundary, on
// the other hand, costs this timeline nothing: rocprofv3 shows back-to-back kernels of one stream with no gap.)
__global__ __launch_bounds__(1024) void k_scan(DevParams P, const int *__restrict__ cell_count,
                                                int *__restrict__ cell_start, int *__restrict__ cursor,
                                                int *__restrict__ task_start, int *__restrict__ task_list,
                                                int *__restrict__ chunk_count,
                                                const CellInfo *__restrict__ celltab, int *__restrict__ status_out, FrameScalars *fs)
{
    __shared__ long long wave_tot[16];
    __shared__ int maxcell_s, maxraw_s;
    __shared__ int chunk_s[SCAN_LDS_CHUNKS];
    // Two prefix sums at once, packed in 64 bits: particles per cell (low word) and
    // 64-particle pair-kernel tasks per cell (high word; only the cells this rank computes).
    // Each thread owns a contiguous run of cells, so the whole scan needs one pass and two barriers.
    constexpr int LDS_CHUNKS = SCAN_LDS_CHUNKS;
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
// One workgroup per own cell.  The scatter left the cell's ids in arrival order; the
// reference's list is in slot order (build_grid walks slots 0..CONTAINER_SIZE-1), so
// rank each id among the cell's ids.  Then gather the snapshot the pair kernel reads
// (T_DATA_TYPE role, ps.cpp:1495-1500) in that order: x,y,z and the mass, the mass
// zeroed for "kids" because bodyBodyInteraction ignores them (app_common.cu:240-243;
// adding r*0 = +-0 leaves an fp32 sum that started at +0 bit-identical).
// Ids ranked at or past the list capacity are the ones the reference kills
// (ps.cpp:1517-1526): their sorted_id entry becomes -1 and the slot is reset.
// CAP: the ids an instance ranks in LDS.  CAP = 1024 (8 KB of LDS: the kernel is a chain of global round trips
// per cell, not a stream) is launched every step, one workgroup per cell; CAP = SORT_MAX (33 KB: four cells per CU)
// serves cells with more than 1024 ids and is launched only when the last frames the host has seen had such a cell
// (`take_big` false in the small instance).  When it is not launched the small instance takes a crowded cell itself,
// ranking through global memory (slow, rare: a cell that jumps past 1024 ids between two frames) -- so the host's
// hint, which is a step or two old, only ever costs time.  The same path serves cells beyond SORT_MAX ids in either
// instance (a dense clump in one cell of an 8-cell segment reaches 8 x 514 = 4112), whatever the list capacity.
template <int CAP>
__device__ __forceinline__ void sort_cell(const DevParams &P, const int c, const bool take_big, const int *__restrict__ cell_start,
                                                     int *__restrict__ sorted_id, int *__restrict__ scratch_ids,
                                                     float4 *pos4, float4 *vel4, float4 *acc4,
                                                     int *cell_arr, uint8_t *pflags,
                                                     float *__restrict__ snap_soa,
                                                     float *__restrict__ snap_age,
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
    constexpr int BITMAP_WORDS = 512;              // the ids' span one bitmap pass covers: 16384 slots
    __shared__ unsigned bitmap[BITMAP_WORDS];
    __shared__ int s_lo, s_hi, s_wt[4], s_total;
    const int tid = threadIdx.x;
    if (tid < 27) s_halo[tid] = 0;
    if (tid == 0) { s_lo = 0x7fffffff; s_hi = -1; s_total = 0; }
    const int start = cell_start[c];
    const int n = cell_start[c + 1] - start;
    if (n == 0) return;
    if (CAP == SMALL ? (n > SMALL && !take_big) : n <= SMALL) return;       // (the other instance's)
    const bool in_lds = n <= CAP;                   // else: ids read from, and the ordered list written to, global memory
    int ci1, ci2, ci3;
    cell_coords(P, c, ci1, ci2, ci3);
    if (in_lds) for (int e = tid; e < n; e += 256) ids[e] = sorted_id[start + e];
    __syncthreads();
    auto id_at = [&](int e) { return in_lds ? ids[e] : sorted_id[start + e]; };
    // The ids in ascending order.  A cell's particles live in the slots of one segment, so the ids span a few
    // thousand values: a bitmap of the span in LDS (one atomicOr per id), a prefix of the words' population
    // counts, and every thread writes out the ids of its two words.  (Before: every id counted the smaller ones
    // among all of them, n/4 16-byte broadcast reads per thread -- at 256 ids per cell the LDS pipe's
    // 14 us of the kernel.)  A span wider than the bitmap (a large chunk_dim: the interior segment of an 8^3-cell
    // chunk spans 111 000 slots) takes one pass per 16384 slots of it.
    int lo = 0x7fffffff, hi = -1;
    for (int e = tid; e < n; e += 256) { const int v = id_at(e); lo = min(lo, v); hi = max(hi, v); }
#pragma unroll
    for (int sft = 32; sft > 0; sft >>= 1) { lo = min(lo, __shfl_xor(lo, sft)); hi = max(hi, __shfl_xor(hi, sft)); }
    if ((tid & 63) == 0) { atomicMin(&s_lo, lo); atomicMax(&s_hi, hi); }
    __syncthreads();
    const int id_lo = s_lo, id_hi = s_hi;
    int *ordered_out = in_lds ? ordered : scratch_ids + start;
    for (int w0 = id_lo; w0 <= id_hi; w0 += BITMAP_WORDS * 32) {
        const int span = min(id_hi - w0 + 1, BITMAP_WORDS * 32), words = (span + 31) >> 5;
        const bool whole = w0 == id_lo && span == id_hi - id_lo + 1;        // one pass: every id is in the window
        for (int w = tid; w < words; w += 256) bitmap[w] = 0;
        __syncthreads();
        for (int e = tid; e < n; e += 256) { const int b = id_at(e) - w0; if (whole || (b >= 0 && b < span)) atomicOr(&bitmap[b >> 5], 1u << (b & 31)); }
        __syncthreads();
        constexpr int WPT = BITMAP_WORDS / 256;                 // words per thread
        unsigned w[WPT];
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < WPT; i++) { w[i] = (WPT * tid + i < words) ? bitmap[WPT * tid + i] : 0u; cnt += __popc(w[i]); }
        const int incl = wave_incl_scan(cnt);
        if ((tid & 63) == 63) s_wt[tid >> 6] = incl;
        __syncthreads();
        int pos = s_total + incl - cnt;
        for (int k = 0; k < (tid >> 6); k++) pos += s_wt[k];
#pragma unroll
        for (int i = 0; i < WPT; i++)
            for (unsigned m = w[i]; m; m &= m - 1) ordered_out[pos++] = w0 + (WPT * tid + i) * 32 + (__ffs(m) - 1);
        __syncthreads();
        if (tid == 0) s_total += s_wt[0] + s_wt[1] + s_wt[2] + s_wt[3];
        __syncthreads();
    }
    // (the small instance keeps what the halo lists need -- position, collision id, face bits -- in registers
    // instead of reading the rows back and redoing the three divisions)
    constexpr int KR = CAP == SMALL ? SMALL / 256 : 1;
    float hx[KR], hy[KR], hz[KR];
    int hid[KR], hm3[KR];
#pragma unroll
    for (int k = 0; k < KR; k++) hm3[k] = 0;
    bool wild_any = false;
    const bool regs = CAP == SMALL && in_lds;      // the register form of the halo pass
    auto row = [&](int e, int k) {
        const int id = ordered_out[e], si = slot_index(P, id);
        // the particle's snapshot (what the reference copies into its T_DATA row, ps.cpp:1495-1500): x, y, z, w and the age
        float4 p = pos4[si];
        const float age = vel4[si].w;
        // A kid is skipped by the reference's force loop and never collides (app_common.cu:240-243, 284-287);
        // here it stays in the lists with mass 0, so that r * 0 = +-0 leaves every sum as it was -- which needs r
        // to be a number.  A child born with the direction (0, 0, 0) has a velocity and, a step later, a position
        // that is not one (0/0, ps.cpp:1306-1333): in the snapshot a kid's position is the origin (its own state,
        // and its T_DATA row, keep what the reference has).
        if (age < P.kid_thr) p.x = p.y = p.z = 0.0f;
        if (e < P.max_per_cell) {
            sorted_id[start + e] = id;
            const float w_eff = (age < P.kid_thr) ? 0.0f : (P.force_sign < 0.f ? -p.w : p.w);
            {   // four separate arrays: what the pair kernel streams
                const size_t cap = (size_t)P.sorted_cap;
                snap_soa[start + e] = p.x; snap_soa[cap + start + e] = p.y;
                snap_soa[2 * cap + start + e] = p.z; snap_soa[3 * cap + start + e] = w_eff;
            }
            snap_age[start + e] = age;
            // collision id: the slot id, or -1 for a body that can never collide (kid, over age)
            const bool collides = !(age < P.kid_thr) && !(age > P.life_thr);
            snap_cid[start + e] = collides ? id : -1;
            // (register form: a candidate whose position is no number -- HALO_ALL -- is left to a pass of its own
            // below, so that the loop every body takes knows nothing of it: with the 26-neighbour case in here
            // the kernel took 10 us more, measured)
            const int m3 = !(halo_count && collides) ? 0 : regs ? halo_dirs_of_numbers(P, ci1, ci2, ci3, p.x, p.y, p.z)
                                                                : halo_dirs(P, ci1, ci2, ci3, p.x, p.y, p.z);
            if (regs && halo_count && collides && !finite3(p.x, p.y, p.z)) wild_any = true;
            if (m3) {
                if (regs) {
#pragma unroll
                    for (int m = 1; m < 8; m++) { const int dir = halo_dir_of_subset(m3, m); if (dir >= 0) atomicAdd(&s_halo[dir], 1); }
                    hx[k] = p.x; hy[k] = p.y; hz[k] = p.z; hid[k] = id; hm3[k] = m3;
                } else
                    for_each_halo_dir(m3, [&](int dir) { atomicAdd(&s_halo[dir], 1); });
            }
        } else {
            // ranked at or past the list capacity: the reference kills it (ps.cpp:1517-1526) -- its entry of the sorted
            // order becomes -1, the slot is reset
            sorted_id[start + e] = -1;
            cell_arr[si] = P.world > 1 ? -2 - cell_arr[si] : -1; pflags[si] = 0;       // (slab: the chunk-capacity walk still needs the cell, see chunk_cap_block)
            pos4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            vel4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            acc4[si] = make_float4(0.f, 0.f, 0.f, 0.f);
            atomicAdd(&(ctr + (blockIdx.x % COUNTER_COPIES))->cell_overflow_kills, 1ull);
            // freed with the already-reset segment (-1,-1): queue record 0 (ps.cpp:1523-1526).  On a
            // slab that does not hold that queue the slot id travels to its owner in the status message.
            if (owns_record(P, 0)) {
                const int k2 = atomicAdd(&fs->n_ops, 1);
                if (k2 < ops_cap) { op_keys[k2] = ((uint64_t)(uint32_t)id << 2) | 2ull; op_args[k2] = id; }
                else atomicOr(&fs->error, ERR_OPS_OVERFLOW);
            } else {
                const int k2 = atomicAdd(&status_out[0], 1);
                if (k2 < STATUS_KILL_CAP) status_out[MSG_HEADER_WORDS + k2] = id;
                else atomicOr(&fs->error, ERR_REMOTE_RECORD0);
            }
        }
    };
    if (regs) {
#pragma unroll
        for (int k = 0; k < KR; k++) { const int e = tid + 256 * k; if (e < n) row(e, k); }
    } else {
        for (int e = tid; e < n; e += 256) row(e, 0);
    }
    if (!halo_count) return;
    const bool wild_cell = __syncthreads_or(wild_any);   // the snapshot rows of this cell are in memory, the directions counted
    if (!regs) {
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
// collapsing cloud, 1024 particles per cell) strides over the cells with a few workgroups and is launched only when the
// host's last look at the frame scalars (max_cell_raw) showed a cell nearly that full.
template <int CAP>
__global__ __launch_bounds__(256) void k_sort_cells(DevParams P, int take_big, const int *__restrict__ cell_order, const int *__restrict__ cell_start, int *__restrict__ sorted_id,
                                                     int *__restrict__ scratch_ids,
                                                     float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                     float *__restrict__ snap_soa, float *__restrict__ snap_age,
                                                     uint64_t *op_keys, int *op_args, int ops_cap,
                                                     int *__restrict__ halo_count, float *__restrict__ halo_f,
                                                     int *__restrict__ halo_id, int *__restrict__ snap_cid,
                                                     int *__restrict__ status_out, FrameScalars *fs, DevCounters *ctr)
{
    if (CAP == 1024) {
        // (workgroups b and b + 8 share an XCD: each XCD takes one contiguous run of the segment-major order)
        sort_cell<CAP>(P, cell_order[xcd_contiguous((int)blockIdx.x, (int)gridDim.x)], take_big != 0, cell_start, sorted_id, scratch_ids, pos4, vel4, acc4, cell_arr, pflags, snap_soa, snap_age,
                       op_keys, op_args, ops_cap, halo_count, halo_f, halo_id, snap_cid, status_out, fs, ctr);
        return;
    }
    if (fs->max_cell_raw <= 1024) return;
    for (int c = blockIdx.x; c < P.n_own_cells; c += gridDim.x) {
        sort_cell<CAP>(P, c, true, cell_start, sorted_id, scratch_ids, pos4, vel4, acc4, cell_arr, pflags, snap_soa, snap_age,
                       op_keys, op_args, ops_cap, halo_count, halo_f, halo_id, snap_cid, status_out, fs, ctr);
        __syncthreads();
    }
}

// ------------------------------------------------------------------ launch wrappers
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
// per-frame scalars; the sticky error word survives.  A step's last kernel does this for the step that follows
// (lifecycle.hip, k_replay_commit); this launch is for a frame that has no finished step before it (a frame
// abandoned half way, state uploaded in between).
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

// diagnostics (psamd_download_force4): the force records in sorted order, from where they live (ForceBuf)
__global__ void k_force_gather(DevParams P, const int *__restrict__ cell_start, const int *__restrict__ sorted_id,
                               const float4 *__restrict__ force4, const float4 *__restrict__ acc4, const uint8_t *__restrict__ flag_slot,
                               float4 *__restrict__ out, int first, int count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const int gi = first + i;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gi < cell_start[P.n_own_cells]) {                      // an own cell's entry (the own block starts at 0)
        const int id = sorted_id[gi], si = id >= 0 ? slot_index(P, id) : -1;
        if (si >= 0) { const float4 a = acc4[si]; v = make_float4(a.x, a.y, a.z, __int_as_float((int)flag_slot[si])); }
    } else if (gi >= P.reg_sorted[1] && P.world > 1) v = force4[gi];
    out[i] = v;
}

hipError_t launch_force_gather(hipStream_t st, const DevParams &P, const DeviceState &d, void *out, int first, int count)
{
    if (count <= 0) return hipSuccess;
    k_force_gather<<<(count + 255) / 256, 256, 0, st>>>(P, d.cell_start, d.sorted_id, d.force4, d.acc4, d.flag_slot, (float4 *)out, first, count);
    return hipGetLastError();
}

hipError_t launch_build_grid(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t *ev, bool tdata_rows, bool big_cells)
{
    const int nwg = std::max(1, (P.slots_total + SLOTS_PER_WG - 1) / SLOTS_PER_WG);
    if (ev) (void)hipEventRecord(ev[0], st);
    k_hist_lds<<<nwg, 1024, 0, st>>>(P, d.cell, d.cell_count, d.fs, d.st);
    PS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[1], st);
    k_scan<<<1, 1024, 0, st>>>(P, d.cell_count, d.cell_start, d.cursor, d.task_start, d.task_list, d.chunk_count, d.celltab, d.status_out, d.fs);
    PS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[2], st);
    const int grid = nwg + (P.world == 1 ? P.num_chunks : 0);
    if (tdata_rows) k_scatter_lds<true><<<grid, 1024, 0, st>>>(P, nwg, d.cell, d.cursor, d.sorted_id, d.chunk_count, d.celltab, d.chunk_segs, d.chunk_skip, d.pos4, d.vel4, d.tdata);
    else k_scatter_lds<false><<<grid, 1024, 0, st>>>(P, nwg, d.cell, d.cursor, d.sorted_id, d.chunk_count, d.celltab, d.chunk_segs, d.chunk_skip, d.pos4, d.vel4, d.tdata);
    PS_LAUNCH_CHECK();
    if (ev) (void)hipEventRecord(ev[3], st);
    // (the ordered ids of a cell that is ranked through global memory go through active_list: written by k_collide_cell later in the frame)
    k_sort_cells<1024><<<P.n_own_cells, 256, 0, st>>>(P, big_cells ? 0 : 1, d.cell_order, d.cell_start, d.sorted_id, d.active_list, d.pos4, d.vel4, d.acc4, d.cell,
                                               d.pflags, d.snap_soa, d.snap_age, d.op_keys, d.op_args, d.ops_cap,
                                               P.two_pass ? d.halo_count : nullptr, d.halo_f, d.halo_id, d.snap_cid, d.status_out, d.fs, d.ctr);
    PS_LAUNCH_CHECK();
    if (big_cells) {
        k_sort_cells<SORT_MAX><<<std::min(P.n_own_cells, 512), 256, 0, st>>>(P, 1, d.cell_order, d.cell_start, d.sorted_id, d.active_list, d.pos4, d.vel4, d.acc4, d.cell,
                                               d.pflags, d.snap_soa, d.snap_age, d.op_keys, d.op_args, d.ops_cap,
                                               P.two_pass ? d.halo_count : nullptr, d.halo_f, d.halo_id, d.snap_cid, d.status_out, d.fs, d.ctr);
        PS_LAUNCH_CHECK();
    }
    if (ev) (void)hipEventRecord(ev[4], st);
    return hipSuccess;
}

}  // namespace psamd
