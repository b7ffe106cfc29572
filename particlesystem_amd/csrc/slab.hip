// slab.hip -- the messages of a multi-GPU step (what pmlib's subscriptions move for the reference, ps.cpp:380-487)
#include "kernels_common.hpp"

namespace psamd {

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
    // The message's room is POOLED over a cell layer: halo_cap_cell bodies per cell on average, any one cell up to
    // its list capacity -- the bodies are packed back to back by this prefix anyway, and the receiver's block of the
    // sorted arrays for a layer is G x G x halo_cap_cell entries.  (Until round 4 every cell was held to halo_cap_cell
    // on its own: one crowded cell in a half-empty layer was an error.)
    block_prefix_1024(ncell, [&](int j) {
        const int n = min(cell_start[c0 + j + 1] - cell_start[c0 + j], P.max_per_cell);
        counts[j] = n;
        return n; }, pack_off, wave_tot, &carry);
    const int GG = P.G * P.G;
    bool over = false;
    for (int l = threadIdx.x; l < ncell / GG; l += 1024) over |= pack_off[(l + 1) * GG] - pack_off[l * GG] > GG * P.halo_cap_cell;
    if (over) atomicOr(&fs->error, ERR_HALO_OVERFLOW);        // (a layer holds more than the message has room for: bodies past the room are not written)
    __syncthreads();
    if (threadIdx.x == 0) { msg[0] = ncell; msg[1] = pack_off[ncell]; msg[2] = fs->error; }
}

// one workgroup per cell of the messages; the first one also closes the rank's status record (everything
// the build stage can raise has been raised by now)
__global__ __launch_bounds__(256) void k_halo_bodies_out(DevParams P, HaloOut2 H, const int *__restrict__ cell_start,
                                                          const SnapSoa snap4,
                                                          const float *__restrict__ snap_age, const int *__restrict__ sorted_id,
                                                          int *__restrict__ status_out, const FrameScalars *__restrict__ fs, const StepState *__restrict__ st)
{
    if (blockIdx.x == 0 && threadIdx.x == 0 && status_out) { status_out[1] = fs->error; status_out[2] = fs->live; status_out[3] = st->last_departures; }
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
    for (int e = threadIdx.x; e < n && (size_t)(dst + e) < cap; e += 256) {
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
    block_prefix_1024(ncell, [&](int j) { return min(max(counts[j], 0), P.max_per_cell); }, unpack_off, wave_tot, &carry);
    {   // every layer must fit its block of the sorted arrays (the sender checked the same: its error bits are in the header)
        const int GG = P.G * P.G;
        bool over = false;
        for (int l = tid; l < ncell / GG; l += 1024) over |= unpack_off[(l + 1) * GG] - unpack_off[l * GG] > GG * P.halo_cap_cell;
        if (__syncthreads_or(over)) {
            if (tid == 0) atomicOr(&fs->error, ERR_HALO_OVERFLOW);
            for (int j = tid; j <= ncell; j += 1024) unpack_off[j] = 0;
            for (int j = tid; j <= split; j += 1024) cell_start[c0 + j] = s0;
            for (int j = tid; j <= ncell - split && split < ncell; j += 1024) cell_start[c1 + j] = s1;
            return;
        }
    }
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

// One workgroup per cell of the messages.  Two-pass pair stage (`lists`): the cell then lists its bodies in the
// halos of the cells around it, like k_sort_cells does for the own cells -- the workgroup has just written them
// (this was a launch of its own, k_remote_halo_lists: 15-20 us of a rank-step for a few microseconds of work).
__global__ __launch_bounds__(256) void k_halo_bodies_in(DevParams P, HaloIn2 H,
                                                         const int *__restrict__ cell_start,
                                                         float *snap_soa, float *__restrict__ snap_age,
                                                         int *__restrict__ sorted_id, int *snap_cid, int lists,
                                                         int *__restrict__ halo_count, float *__restrict__ halo_f,
                                                         int *__restrict__ halo_id)
{
    __shared__ int s_halo[27], s_halo_base[27];
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
    if (!lists) return;
    __syncthreads();                    // the cell's bodies, as written above, for every thread of the workgroup
    list_in_neighbour_halos(P, lc, dst, max(n, 0), SnapSoa{snap_soa, sc}, snap_cid, halo_count, halo_f, halo_id, s_halo, s_halo_base, false);
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
                                                   const int *__restrict__ cell_start, const ForceBuf force4, FrameScalars *fs)
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
    for (int e = threadIdx.x; e < n; e += blockDim.x) force4.put(P, c, dst + e, src[rel + e]);      // (an own cell: by slot)
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

// Run once the status records of all ranks are in.  Two launches:
//
// k_chunk_census, at the start of the PAIR stage (the records must have landed by then): one workgroup per chunk -- the
// chunk lists' capacity rule (ps.cpp:1502-1508) across ranks.  The chunk's count is the sum of the ranks' parts (also what
// hostGridMax[0] is the maximum of); if it passed the capacity, this rank ranks the particles in its own segments of the
// chunk behind what the census says precedes them in slot order.  Before the pair stage, because that stage writes a
// particle's new acceleration into the particle's own record (ForceBuf) and must leave the records of the particles the
// rule takes out of the step alone.
//
// k_status_merge, before k_apply.  Workgroups [0, world), one per rank's record: adopt its error bits (status_error: the
// OR over ALL records, this rank's own included -- the same word on every rank, which is what makes a failure
// collective); the owner of queue record 0 queues the reported cell-overflow kills as the inserts build_grid would have
// made (ps.cpp:1523-1526): key = chunk field 0 | slot | insert, i.e. before every calc_forces operation and in slot
// order; workgroup 0 also settles the transfer messages' next capacity.  Workgroups past them: the force records of the
// lent-out layers come home (one workgroup per lent-out cell).
__global__ __launch_bounds__(1024) void k_chunk_census(DevParams P, const int *__restrict__ status_all, int *__restrict__ chunk_count,
                                                        const int *__restrict__ cell_arr, const CellInfo *__restrict__ celltab,
                                                        const int2 *__restrict__ chunk_segs, uint8_t *__restrict__ chunk_skip, FrameScalars *fs)
{
    __shared__ int s_before[4];
    const int ch = (int)blockIdx.x;
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
}

__global__ __launch_bounds__(1024) void k_status_merge(DevParams P, const int *__restrict__ status_all, uint64_t *op_keys, int *op_args,
                                                        int ops_cap, FrameScalars *fs,
                                                        int force_j0, const int *__restrict__ force_msg, const int *__restrict__ pack_off,
                                                        const int *__restrict__ cell_start, const ForceBuf force4, StepState *stp)
{
    // workgroups past the status records: the force records of the lent-out layers come home
    // (one workgroup per lent-out cell; same stage, so the same launch)
    if ((int)blockIdx.x >= P.world) {
        unpack_force_block(P, (int)blockIdx.x - P.world, force_j0, force_msg, pack_off, cell_start, force4, fs);
        return;
    }
    const int r = blockIdx.x;
    const int *st = status_all + (size_t)r * P.status_words;
    if (threadIdx.x == 0 && st[1]) { atomicOr(&fs->status_error, st[1]); if (r != P.rank) atomicOr(&fs->error, st[1]); }
    if (r == 0 && threadIdx.x == 0) {
        // The transfer messages' capacity follows the traffic: every rank said how many records it sent in one direction in the
        // step before; all ranks hold all records and apply the same rule, so the number below is the same everywhere and every
        // host adopts it in the same step (two steps on: psamd_slab_build) -- both ends of every message change size together,
        // with no negotiation round.  The rule looks three steps ahead (a decision rests on the traffic of the step before and
        // takes effect two steps on): the busiest rank's count plus four times its rise since the step before, doubled, and
        // a little on top.  More than the messages hold now: they grow to it (up to the room of the buffers); less than half:
        // they shrink to it (a run's first step is a rise from nothing -- the forecast then is ten times the traffic, and a
        // steady run would carry that room for ever), never below what the context was created with.
        int peak = 0;
        for (int q = 0; q < P.world; q++) peak = max(peak, status_all[(size_t)q * P.status_words + 3]);
        const long long rise = max(0, peak - stp->peak_prev);
        stp->peak_prev = peak;
        const long long need = (2ll * ((long long)peak + 4ll * rise) + 64ll + 63ll) & ~63ll;
        long long next = P.xfer_cap;
        if (need > next) next = need; else if (2 * need <= next) next = need;
        fs->xfer_cap_next = (int)max((long long)P.xfer_cap0, min((long long)P.xfer_cap_max, next));
    }
    if (r == P.rank || !owns_record(P, 0)) return;
    const int n = min(st[0], STATUS_KILL_CAP);
    for (int e = threadIdx.x; e < n; e += 1024) {
        const int id = st[MSG_HEADER_WORDS + e];
        const int k = atomicAdd(&fs->n_ops, 1);
        if (k < ops_cap) { op_keys[k] = ((uint64_t)(uint32_t)id << 2) | 2ull; op_args[k] = id; }
        else atomicOr(&fs->error, ERR_OPS_OVERFLOW);
    }
}

// the chunk lists' capacity rule across ranks: before the pair stage (the status records of all ranks have landed)
hipError_t launch_chunk_census(hipStream_t st, const DevParams &P, const DeviceState &d, const int *status_all)
{
    if (!status_all || P.world <= 1) return hipSuccess;
    k_chunk_census<<<P.num_chunks, 1024, 0, st>>>(P, status_all, d.chunk_count, d.cell, d.celltab, d.chunk_segs, d.chunk_skip, d.fs);
    return hipGetLastError();
}

// force_msg (may be null): the force records of the lent-out layers, unpacked by extra workgroups of the same launch
hipError_t launch_status_merge(hipStream_t st, const DevParams &P, const DeviceState &d, const int *status_all,
                               int force_j0, const int *force_msg, const int *pack_off)
{
    if (!status_all || P.world <= 1) return hipSuccess;
    const int ncell = force_msg ? std::max(0, P.lentout_c1 - P.lentout_c0) : 0;
    k_status_merge<<<P.world + ncell, 1024, 0, st>>>(P, status_all, d.op_keys, d.op_args, d.ops_cap, d.fs, force_j0, force_msg, pack_off,
                                                     d.cell_start, force_buf(d), d.st);
    return hipGetLastError();
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
    k_halo_bodies_out<<<blocks, 256, 0, st>>>(P, H, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_age, d.sorted_id, d.status_out, d.fs, d.st);
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
    if (ncell_below > 0) {
        const int split = P.reg_layers[1] * GG;
        H.h[H.n++] = HaloIn{P.reg_base[1], P.reg_base[2], ncell_below, split, P.reg_sorted[1], P.reg_sorted[2], P.reg_layers[2] > 0 ? 1 : 0, msg_below, off_below};
    }
    if (ncell_above > 0) {
        H.h[H.n++] = HaloIn{P.reg_base[3], 0, ncell_above, ncell_above, P.reg_sorted[3], 0, 0, msg_above, off_above};
    }
    if (H.n == 0) return hipSuccess;
    if (H.n == 1) H.h[1] = HaloIn{0, 0, 0, 0, 0, 0, 0, nullptr, nullptr};
    k_halo_prefix_in<<<H.n, 1024, 0, st>>>(P, H, d.cell_start, d.task_list, d.fs);
    PS_LAUNCH_CHECK();
    const int cells = H.h[0].ncell + H.h[1].ncell;
    k_halo_bodies_in<<<cells, 256, 0, st>>>(P, H, d.cell_start, d.snap_soa, d.snap_age, d.sorted_id, d.snap_cid, P.two_pass ? 1 : 0,
                                            d.halo_count, d.halo_f, d.halo_id);
    PS_LAUNCH_CHECK();
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

}  // namespace psamd
