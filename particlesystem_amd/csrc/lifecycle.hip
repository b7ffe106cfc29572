// lifecycle.hip -- free-slot queues and relocation in the reference's serial order (ps.cpp:1335-1374, app_common.cu:305-376)
//
// Three launches behind k_apply, none of which waits for the host or is chosen by it:
//   k_ops_hist       census of the step's queue operations per queue record; extra workgroups of the same launch run the
//                    first relocation phase (copy_particle into the staging area, reset_particle on the vacated slot)
//   k_ops_scatter    the operations bucketed by record; the last workgroup hands the step's scalars to the host's pinned
//                    record (which the host reads a step LATE: nothing in this step depends on the host having seen them)
//   k_replay_commit  one workgroup per queue record replays its list on the circular FIFO -- whatever its length -- and
//                    drops every relocated particle / newborn child into the slot its remove operation was handed
//                    (the second relocation phase, fused: the operation's argument IS the move record); the same
//                    launch zeroes the per-frame counts for the NEXT step's init_iframe (ps.cpp:1574-1606).
#include "kernels_common.hpp"

namespace psamd {

constexpr int REPLAY_THREADS = 512;
constexpr int MSD_LEVELS = 4;        // partition levels of the long-list sort held in LDS (8 bits each below the keys' common prefix)

// ops per record (rec_count and rec_cursor are zeroed with the frame); n_ops is still on the device at
// this point.  (Counting where the operations are made, inside k_apply, was tried twice: a
// workgroup-wide LDS histogram cost that kernel 21 us -- two more barriers per 1024-thread workgroup --
// and per-wave aggregated global atomics 80 us: the 729 counters share 46 cache lines and same-line
// atomics are served one at a time.  This kernel takes 5 us.)
//
// Workgroups past the first `nhist`: relocation phase 1 for the step's move records (nothing in it depends on the
// queues, so it rides along instead of being a launch).
// One GPU: read the moving particle (copy_particle, ps.cpp:1363) or the parent of a child to be
// born into the staging area, and reset_particle the slot a relocation vacates (ps.cpp:1367).  A
// parent that also relocates this step has two records, written side by side by its k_apply
// thread (birth, then relocation): the relocation's thread stages for both and then resets, the
// birth's thread stands back -- so no record reads a slot another thread zeroes.
// Slab: everything local was staged when the outboxes were closed (k_moves_stage); only the reset is left.
__device__ __forceinline__ void moves_stage_reset(const DevParams &P, int m, int n, MoveRec *moves,
                                                  float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                  float4 *stage)
{
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

__global__ __launch_bounds__(1024) void k_ops_hist(DevParams P, int nhist, const uint64_t *__restrict__ keys, const FrameScalars *fs,
                                                    int ops_cap, int rec_shift, int nrec, int *rec_count,
                                                    MoveRec *moves, int moves_cap, float4 *pos4, float4 *vel4, float4 *acc4,
                                                    int *cell_arr, uint8_t *pflags, float4 *stage)
{
    __shared__ int h[LDS_CELLS];
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nhist) {
        const int nm = min(fs->n_moves, moves_cap), stride = ((int)gridDim.x - nhist) * 1024;
        for (int m = ((int)blockIdx.x - nhist) * 1024 + tid; m < nm; m += stride)
            moves_stage_reset(P, m, nm, moves, pos4, vel4, acc4, cell_arr, pflags, stage);
        return;
    }
    const int n = min(fs->n_ops, ops_cap);
    if ((long long)blockIdx.x * SLOTS_PER_WG >= n) return;
    const bool lds = nrec <= LDS_CELLS;
    if (lds) { for (int r = tid; r < nrec; r += 1024) h[r] = 0; __syncthreads(); }
    for (long long b0 = (long long)blockIdx.x * SLOTS_PER_WG; b0 < n; b0 += (long long)nhist * SLOTS_PER_WG)
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
// host's pinned copy, and the step's number behind it once the record is out.  The host keeps TWO records
// and the step's number picks one: the host reads a step's record while the next step is already running
// (it enqueues step k + 1 once it has seen the record of step k - 1), so the record of step k must not
// land on the one of step k - 1.  Called by all threads of one workgroup.
__device__ __forceinline__ void publish_scalars(const FrameScalars *fs, FrameScalars *fs_host, int longest, StepState *st)
{
    constexpr int WORDS = (int)(sizeof(FrameScalars) / sizeof(int)), SKIP = (int)(offsetof(FrameScalars, max_bucket) / sizeof(int)),
                  SEQ = (int)(offsetof(FrameScalars, seq) / sizeof(int));
    static_assert(sizeof(FrameScalars) % sizeof(int) == 0, "copied word by word");
    const int seq = st->seq + 1;                      // one more than the last one this context handed out (the host counts along)
    const int *src = reinterpret_cast<const int *>(fs);
    int *dst = reinterpret_cast<int *>(fs_host + (seq & 1));
    for (int i = threadIdx.x; i < WORDS; i += blockDim.x)
        if (i != SEQ) dst[i] = i == SKIP ? longest : src[i];              // (max_bucket is being written by this very workgroup)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        // the step this record closes is over as far as its number goes: the next frame's first kernel makes it step + 1
        st->seq = seq; st->pending = 1;
        __hip_atomic_store(&fs_host[seq & 1].seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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

// Bucket the operations by queue record.  SCAN: every workgroup first works out the buckets' starts
// for itself (an exclusive prefix of rec_count in LDS: a few hundred records) instead of waiting
// for a one-workgroup kernel to do it; workgroup 0 also leaves them in rec_start for the replay and
// publishes the longest bucket.  Grid-stride over the operations: whatever the step really produced is covered.
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
        start = s_start;
    }
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

// ------------------------------------------------------------------ replay + commit
// Relocation phase 2 for move record m, whose remove operation has just been handed slot `dst`: drop the
// particle (or the newborn child) there.  The slot belongs to the queue's own segment and was free until
// this operation: no other workgroup touches it.  (The first phase -- staging, and the reset of the vacated
// slots, one of which may be this very slot handed out again -- ran in the launch before.)
struct ParticleArrays { float4 *pos4, *vel4, *acc4; int *cell; uint8_t *pflags; };

__device__ __forceinline__ void commit_move(const DevParams &P, const StepState *__restrict__ stp, const MoveRec *__restrict__ moves, int m, int dst,
                                            const ParticleArrays &A, const float4 *__restrict__ stage)
{
    if (dst < 0) return;                                  // the queue was empty: the relocation is lost / the birth fails (ps.cpp:1340-1343)
    const MoveRec r = moves[m];
    const float4 *s = stage + (size_t)3 * m;
    const int di = slot_index(P, dst);
    if ((r.kind & 0xff) == 0) {
        A.pos4[di] = s[0]; A.vel4[di] = s[1]; A.acc4[di] = s[2];
        A.cell[di] = r.new_cell;
        A.pflags[di] = (r.kind & MOVE_PARENT) ? 1 : 0;
    } else {
        // create_particle_s (app.cu:189-208): child at the parent's position, opposite
        // velocity, age 0, fresh fertility age from the counter-based RNG
        const uint64_t h0 = splitmix64(P.seed ^ ((uint64_t)(uint32_t)stp->step << 32) ^ (uint64_t)(uint32_t)r.src);
        const uint64_t h3 = splitmix64(splitmix64(splitmix64(h0)));
        const double u = (double)(h3 >> 11) * (1.0 / 9007199254740992.0);
        const float fert = (float)((double)P.fert_lo + u * (double)(P.fert_hi - P.fert_lo));
        const float4 pp = s[0], pv = s[1];
        A.pos4[di] = make_float4(pp.x, pp.y, pp.z, P.w_default);
        A.vel4[di] = make_float4((float)(-1.0 * (double)pv.x), (float)(-1.0 * (double)pv.y),
                                 (float)(-1.0 * (double)pv.z), 0.0f);
        A.acc4[di] = make_float4(0.f, 0.f, 0.f, fert);
        A.cell[di] = r.new_cell;
        A.pflags[di] = 0;
    }
}

// q_insert / q_remove exactly as the reference's (app_common.cu:305-376), one at a time: what the lists are walked
// with when a queue is about to run empty or to fill up.  The queue's slots are read and written through `at`.
struct SerialTally { unsigned long long lost = 0, reloc = 0, births = 0, births_failed = 0; };
template <typename At>
__device__ __forceinline__ void serial_op(QueueInfo &q, int sub, int arg, At at, MoveRec *moves, SerialTally &t)
{
    if (sub == 2) {                                // q_insert(arg), app_common.cu:346-376
        if (q.count == q.seg_size) return;
        if (q.count == 0) { q.front = q.rloc; q.rear = q.rloc; }
        else if (q.rear == q.rloc + q.seg_size - 1) q.rear = q.rloc;
        else q.rear++;
        q.count++;
        at(q.rear) = arg;
    } else {                                       // q_remove, app_common.cu:305-339
        int item = -1;
        if (q.count > 0) {
            const int pos = q.front;
            if (q.count == 1) { q.front = -1; q.rear = -1; }
            else if (q.front == q.rloc + q.seg_size - 1) q.front = q.rloc;
            else q.front++;
            q.count--;
            item = at(pos); at(pos) = -1;
        }
        moves[arg].dst = item;                     // (committed by all threads once the walk is over)
        if (sub == 1) { if (item >= 0) t.reloc++; else t.lost++; }
        else { if (item >= 0) t.births++; else t.births_failed++; }
    }
}

// bitonic sort of np (a power of two >= 2) 8-byte keys in LDS, ARGS: with a 4-byte argument each moved along.
// For j <= 64 both elements of pair p lie in the 128-element chunk p >> 6, and all 64
// pairs of a chunk belong to one wave (p = t + m * NT, NT a multiple of 64): such
// stages need no workgroup barrier, only the wave's own order -- 56 of the 66 stages
// at 2048 operations, and the barriers were what a long list cost.
// (Ranking by counting -- every thread compares its keys with all of them, two per 16-byte broadcast read, no
// barriers -- was tried for the short lists: LDS-bandwidth-bound, 65 us against the network's 36 for the usual step.)
template <bool ARGS>
__device__ __forceinline__ void lds_bitonic(uint64_t *kbuf, int *abuf, int np)
{
    constexpr int NT = REPLAY_THREADS;
    const int tid = threadIdx.x;
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
                    if (ARGS) { const int x = abuf[e]; abuf[e] = abuf[partner]; abuf[partner] = x; }
                }
            }
            const int next_j = j > 1 ? (j >> 1) : k;              // the next k starts at j = k
            if (j > 64 || next_j > 64) __syncthreads();
            else PS_WAVE_SYNC();
        }
    __syncthreads();
}

// What a replay instance keeps in LDS.  keys + args while sorting; afterwards the same bytes hold the insert
// list (closed form) or the copy of the segment the serial walk works on.
template <int CAP>
struct ReplayLds {
    static constexpr int KEY_BYTES = (CAP + 64) * 8, SORT_BYTES = KEY_BYTES + CAP * 4;
    static constexpr int WINDOW_SLOTS = KEY_BYTES / 4;         // largest segment the serial walk copies into the key area
    static_assert(CAP * 4 <= KEY_BYTES, "abuf must survive the reuse of the key area");
    uint64_t *kbuf; int *abuf; int *window; unsigned char *sub; int *wave_tot; int *flag;
};

// A list of at most CAP operations: rank them by key in LDS, then replay.  When the queue provably neither runs
// empty nor fills up during the step (prefix sums of +1/-1 over the sorted operations), every operation's effect on
// the circular FIFO has a closed form -- the k-th remove takes logical element k, the
// k-th insert becomes logical element count0 + k -- and all of them are applied at once;
// otherwise one lane walks the list exactly as q_insert / q_remove do.
template <int CAP>
__device__ __forceinline__ void replay_short(const DevParams &P, const int rec, const int start, const int n,
                                             const uint64_t *__restrict__ keys, const int *__restrict__ args,
                                             QueueInfo *qinfo, int *queue, MoveRec *moves, DevCounters *ctr,
                                             const StepState *__restrict__ stp, const ParticleArrays &A, const float4 *__restrict__ stage,
                                             const ReplayLds<CAP> &L)
{
    constexpr int NT = REPLAY_THREADS;
    uint64_t *kbuf = L.kbuf;
    int *abuf = L.abuf, *window = L.window, *wave_tot = L.wave_tot;
    unsigned char *s_sub = L.sub;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    QueueInfo q = qinfo[rec];
    const bool in_lds = q.seg_size <= ReplayLds<CAP>::WINDOW_SLOTS;
    queue += slot_index(P, q.rloc) - q.rloc;           // owned segments only, back to back
    // Inside one bucket the record bits of the keys are all the same: what is sorted is (chunk, id, sub) with the
    // operation's place in the bucket packed in below it -- one 8-byte word per operation, its argument fetched
    // through that place once the order is known.  (With the arguments carried along as a second array every
    // exchange moved 24 bytes instead of 16; the sort is bound by LDS bandwidth, five workgroups to a CU.)
    constexpr int IDX_BITS = CAP == 2048 ? 11 : CAP == 4096 ? 12 : 13;
    static_assert((1 << IDX_BITS) >= CAP, "an operation's place in the bucket must fit");
    const bool packed_keys = P.key_rec_shift + IDX_BITS <= 64;              // (else, a geometry with > 2^51 (chunk, id) pairs: keys and arguments side by side)
    const uint64_t low_mask = P.key_rec_shift >= 64 ? ~0ull : ((1ull << P.key_rec_shift) - 1ull);
    for (int e = tid; e < n; e += NT) {
        const uint64_t k = keys[start + e];
        kbuf[e] = packed_keys ? (((k & low_mask) << IDX_BITS) | (uint64_t)e) : k;
        abuf[e] = args[start + e];
    }
    if (tid == 0) *L.flag = 0;
    int np = 2;
    while (np < n) np <<= 1;
    for (int e = n + tid; e < np; e += NT) { kbuf[e] = ~0ull; abuf[e] = -1; }
    __syncthreads();
    if (packed_keys) lds_bitonic<false>(kbuf, abuf, np); else lds_bitonic<true>(kbuf, abuf, np);
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
    int *ins_arg = (int *)kbuf;                        // keys no longer needed

    // prefix counts of inserts / removes before each of my (up to CAP / NT consecutive) operations
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
        if (bad) *L.flag = 1;
    }
    __syncthreads();
    unsigned long long lost = 0, reloc = 0, births = 0, births_failed = 0;
    if (*L.flag) {
        // rare (a queue about to run empty or fill up): one lane walks the list exactly as
        // q_insert / q_remove do, on a copy of the segment in LDS when it fits
        if (in_lds) { for (int e = tid; e < q.seg_size; e += NT) window[e] = queue[q.rloc + e]; __syncthreads(); }
        if (tid == 0) {
            SerialTally t;
            if (in_lds) for (int e = 0; e < n; e++) serial_op(q, s_sub[e], s_arg[e], [&](int pos) -> int & { return window[pos - q.rloc]; }, moves, t);
            else for (int e = 0; e < n; e++) serial_op(q, s_sub[e], s_arg[e], [&](int pos) -> int & { return queue[pos]; }, moves, t);
            lost = t.lost; reloc = t.reloc; births = t.births; births_failed = t.births_failed;
        }
        __syncthreads();                               // (the walk's moves[].dst, for every thread of the workgroup)
        if (in_lds) for (int e = tid; e < q.seg_size; e += NT) queue[q.rloc + e] = window[e];
        for (int e = tid; e < n; e += NT)
            if (s_sub[e] != 2) commit_move(P, stp, moves, s_arg[e], moves[s_arg[e]].dst, A, stage);
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
                commit_move(P, stp, moves, s_arg[e], item, A, stage);
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
}

// A list LONGER than the instance sorts in LDS (a collapsing cloud; 1024 particles per cell; queue record 0, where the
// reference frees overflow-killed slots, ps.cpp:1523-1526; or simply a step whose lists outgrew the host's hint):
// the same workgroup sorts it in global memory -- its bucket [start, start + n) of the sorted arrays, with the same
// range of the unsorted arrays (dead once bucketed) as the other side of a ping-pong -- by most-significant-digit
// radix partition, 8 bits a level below the keys' common prefix, down to ranges the LDS network takes; then streams
// the sorted list through the closed form in chunks of one operation per thread (three passes: count and check,
// removes, queue update), or walks it with one lane when the queue would run empty or fill up.  Keys are distinct
// (record | chunk | slot | sub-step), so the order is the reference's serial order whatever the partitions did.
// (Until round 5 this was the HOST's business: it read the step's scalars, saw a list beyond the replay's reach and
// launched a radix sort of all keys plus a streamed replay -- which kept the host on every step's critical path.)
template <int CAP>
__device__ void replay_long(const DevParams &P, const int rec, const int start, const int n,
                            uint64_t *keys, int *args, uint64_t *keys2, int *args2,
                            QueueInfo *qinfo, int *queue, MoveRec *moves, DevCounters *ctr,
                            const StepState *__restrict__ stp, const ParticleArrays &A, const float4 *__restrict__ stage,
                            const ReplayLds<CAP> &L, int (*s_cnt)[256], int *s_cur, int *s_misc)
{
    constexpr int NT = REPLAY_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t *K[2] = {keys + start, keys2 + start};
    int *Ar[2] = {args + start, args2 + start};

    // ---- the keys' common prefix below the record bits: partition from the first bit that differs
    uint64_t k_or = 0, k_and = ~0ull;
    for (int e = tid; e < n; e += NT) { const uint64_t k = K[0][e]; k_or |= k; k_and &= k; }
    for (int d = 32; d > 0; d >>= 1) { k_or |= __shfl_xor(k_or, d); k_and &= __shfl_xor(k_and, d); }
    uint64_t *red = reinterpret_cast<uint64_t *>(L.kbuf);
    if (lane == 0) { red[2 * wv] = k_or; red[2 * wv + 1] = k_and; }
    __syncthreads();
    for (int k = 0; k < NT / 64; k++) { k_or |= red[2 * k]; k_and &= red[2 * k + 1]; }
    __syncthreads();
    const uint64_t diff = (k_or ^ k_and) & (P.key_rec_shift >= 64 ? ~0ull : ((1ull << P.key_rec_shift) - 1ull));
    const int top = diff ? 64 - __builtin_clzll(diff) : 0;     // bits [0, top) distinguish the keys

    // tile: range [lo, lo + len), len <= CAP, living on side `side` -> sorted, on side 0
    auto sort_tile = [&](int lo, int len, int side) {
        int np = 2;
        while (np < len) np <<= 1;
        for (int e = tid; e < np; e += NT) {
            L.kbuf[e] = e < len ? K[side][lo + e] : ~0ull;
            L.abuf[e] = e < len ? Ar[side][lo + e] : -1;
        }
        __syncthreads();
        lds_bitonic<true>(L.kbuf, L.abuf, np);
        for (int e = tid; e < len; e += NT) { K[0][lo + e] = L.kbuf[e]; Ar[0][lo + e] = L.abuf[e]; }
        __syncthreads();
    };
    // Fallback for a range that is still longer than a tile when the levels (or the bits) have run out: a bitonic
    // network straight on global memory (slow; keys this skewed are not expected, a wrong order is not an option).
    // Ascending-only network, so that a length that is no power of two needs no padding: the first stage of every merge compares
    // e with its mirror image in the 2k-block (partner = block_end - offset), later stages are plain half-cleaners;
    // elements whose partner lies beyond `len` stay put (they are the largest of their block by induction).
    auto sort_global_merge = [&](int lo, int len, int side) {
        if (side != 0) { for (int e = tid; e < len; e += NT) { K[0][lo + e] = K[1][lo + e]; Ar[0][lo + e] = Ar[1][lo + e]; } __syncthreads(); }
        int np = 2;
        while (np < len) np <<= 1;
        auto cmpx = [&](int a_i, int b_i) {
            if (b_i >= len) return;
            const uint64_t a = K[0][lo + a_i], b = K[0][lo + b_i];
            if (a > b) {
                K[0][lo + a_i] = b; K[0][lo + b_i] = a;
                const int x = Ar[0][lo + a_i]; Ar[0][lo + a_i] = Ar[0][lo + b_i]; Ar[0][lo + b_i] = x;
            }
        };
        for (int k = 2; k <= np; k <<= 1) {
            for (int t = tid; t < (np >> 1); t += NT) {                  // mirror stage
                const int blk = t / (k >> 1), o = t % (k >> 1);
                cmpx(blk * k + o, blk * k + k - 1 - o);
            }
            __syncthreads();
            for (int j = k >> 2; j > 0; j >>= 1) {                       // half-cleaners
                for (int t = tid; t < (np >> 1); t += NT) {
                    const int e = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    cmpx(e, e | j);
                }
                __syncthreads();
            }
        }
    };

    // ---- iterative depth-first MSD partition.  Level l works on range [lo_l, lo_l + len_l) living on side l & 1;
    // s_cnt[l][d] = sizes of its 256 parts (on side (l + 1) & 1 once partitioned), s_cur[l] = next part to descend into,
    // s_misc[l] = running offset of that part.
    int *hist = L.window;                                   // 256 counters + 256 cursors (the key area is free between tiles)
    int level = -1;
    int lo = 0, len = n, shift = top;
    bool descend = true;                                     // a new range (lo, len, shift) to handle at level + 1
    for (;;) {
        if (descend) {
            const int side = (level + 1) & 1;
            if (len <= CAP) sort_tile(lo, len, side);
            else if (shift <= 0 || level + 1 >= MSD_LEVELS) sort_global_merge(lo, len, side);
            else {
                // partition [lo, lo + len) on the 8 bits below `shift`
                level++;
                const int b = min(8, shift), sh = shift - b;
                for (int d = tid; d < 512; d += NT) hist[d] = 0;
                __syncthreads();
                for (int e = tid; e < len; e += NT) atomicAdd(&hist[(int)((K[side][lo + e] >> sh) & ((1u << b) - 1u))], 1);
                __syncthreads();
                if (tid < 256) {
                    const int v = hist[tid];
                    const int incl = wave_incl_scan(v);
                    if (lane == 63) L.wave_tot[wv] = incl;
                    s_cnt[level][tid] = v;
                    hist[256 + tid] = incl - v;              // exclusive within the wave; the waves' offsets below
                }
                __syncthreads();
                if (tid < 256) { int o = 0; for (int k = 0; k < wv; k++) o += L.wave_tot[k]; hist[256 + tid] += o; }
                if (tid == 0) { s_cur[level] = 0; s_misc[level] = lo; s_misc[MSD_LEVELS + level] = sh; }
                __syncthreads();
                for (int e = tid; e < len; e += NT) {
                    const uint64_t k = K[side][lo + e];
                    const int a = Ar[side][lo + e];
                    const int p = atomicAdd(&hist[256 + (int)((k >> sh) & ((1u << b) - 1u))], 1);
                    K[side ^ 1][lo + p] = k; Ar[side ^ 1][lo + p] = a;
                }
                __syncthreads();
            }
            descend = false;
        }
        if (level < 0) break;
        // next part of the current level
        const int d = s_cur[level];
        if (d >= 256) { level--; continue; }
        const int plen = s_cnt[level][d], plo = s_misc[level];
        __syncthreads();
        if (tid == 0) { s_cur[level] = d + 1; s_misc[level] = plo + plen; }
        __syncthreads();
        if (plen == 0) continue;
        lo = plo; len = plen; shift = s_misc[MSD_LEVELS + level];
        descend = true;
    }
    __syncthreads();

    // ---- the sorted list, streamed: keys / args = side 0, scratch for the insert arguments = side 1's args
    const uint64_t *skeys = K[0];
    const int *sargs = Ar[0];
    int *scratch = Ar[1];
    QueueInfo q = qinfo[rec];
    queue += slot_index(P, q.rloc) - q.rloc;                // the queue array is stored like the slots: only the owned segments, back to back
    const int count0 = q.count, size = q.seg_size;
    int *s_carry = L.flag + 1, *s_bad = L.flag;
    // prefix of (inserts, removes) over one chunk of NT operations, carried from chunk to chunk
    auto chunk_scan = [&](int c0, int &sub, int &arg, int &ins_b, int &rem_b) {
        const int e = c0 + tid;
        sub = -1; arg = 0;
        if (e < n) { sub = (int)(skeys[e] & 3ull); arg = sargs[e]; }
        const int vi = sub == 2 ? 1 : 0, vr = (sub >= 0 && sub != 2) ? 1 : 0;
        const int ii = wave_incl_scan(vi), ir = wave_incl_scan(vr);
        if (lane == 63) { L.wave_tot[wv] = ii; L.wave_tot[NT / 64 + wv] = ir; }
        __syncthreads();
        int oi = s_carry[0], orr = s_carry[1];
        for (int k = 0; k < wv; k++) { oi += L.wave_tot[k]; orr += L.wave_tot[NT / 64 + k]; }
        ins_b = oi + ii - vi; rem_b = orr + ir - vr;
        __syncthreads();
        if (tid == NT - 1) { s_carry[0] = oi + ii; s_carry[1] = orr + ir; }
        __syncthreads();
    };
    if (tid == 0) { s_carry[0] = 0; s_carry[1] = 0; *s_bad = count0 <= 0 ? 1 : 0; }
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += NT) {
        int sub, arg, ins_b, rem_b;
        chunk_scan(c0, sub, arg, ins_b, rem_b);
        if (sub >= 0) {
            const int c = count0 + ins_b - rem_b;
            if (sub == 2) { if (!(c < size)) *s_bad = 1; scratch[ins_b] = arg; }
            else if (!(c >= 2)) *s_bad = 1;
        }
    }
    __syncthreads();
    const int I = s_carry[0], R = s_carry[1];
    unsigned long long lost = 0, reloc = 0, births = 0, births_failed = 0;
    if (!*s_bad) {
        int *seg = queue + q.rloc;
        const int F = q.front - q.rloc;                // offset of logical element 0
        __syncthreads();
        if (tid == 0) { s_carry[0] = 0; s_carry[1] = 0; }
        __syncthreads();
        for (int c0 = 0; c0 < n; c0 += NT) {
            int sub, arg, ins_b, rem_b;
            chunk_scan(c0, sub, arg, ins_b, rem_b);
            if (sub >= 0 && sub != 2) {
                const int item = (rem_b < count0) ? seg[(F + rem_b) % size] : scratch[rem_b - count0];
                moves[arg].dst = item;
                commit_move(P, stp, moves, arg, item, A, stage);
                if (sub == 1) reloc++; else births++;
            }
        }
        __syncthreads();
        for (int r = tid; r < R; r += NT) seg[(F + r) % size] = -1;             // every removed element
        __syncthreads();
        for (int k = tid; k < I; k += NT)                                        // inserts that stayed
            if (count0 + k >= R) seg[(F + count0 + k) % size] = scratch[k];
        if (tid == 0) {
            q.count = count0 + I - R;
            q.front = q.rloc + (F + R) % size;
            q.rear = q.rloc + (F + count0 + I - 1) % size;
            qinfo[rec] = q;
        }
    } else {
        const bool in_lds = q.seg_size <= ReplayLds<CAP>::WINDOW_SLOTS;
        int *window = L.window;
        if (in_lds) { for (int e = tid; e < q.seg_size; e += NT) window[e] = queue[q.rloc + e]; }
        __syncthreads();
        if (tid == 0) {
            SerialTally t;
            if (in_lds) for (int e = 0; e < n; e++) serial_op(q, (int)(skeys[e] & 3ull), sargs[e], [&](int pos) -> int & { return window[pos - q.rloc]; }, moves, t);
            else for (int e = 0; e < n; e++) serial_op(q, (int)(skeys[e] & 3ull), sargs[e], [&](int pos) -> int & { return queue[pos]; }, moves, t);
            lost = t.lost; reloc = t.reloc; births = t.births; births_failed = t.births_failed;
            qinfo[rec] = q;
        }
        __syncthreads();
        if (in_lds) for (int e = tid; e < q.seg_size; e += NT) queue[q.rloc + e] = window[e];
        for (int e = tid; e < n; e += NT)
            if ((skeys[e] & 3ull) != 2ull) commit_move(P, stp, moves, sargs[e], moves[sargs[e]].dst, A, stage);
    }
    DevCounters *mine = ctr + (blockIdx.x % COUNTER_COPIES);
    if (reloc) atomicAdd(&mine->relocations, reloc);
    if (lost) atomicAdd(&mine->relocations_lost, lost);
    if (births) atomicAdd(&mine->births, births);
    if (births_failed) atomicAdd(&mine->births_failed, births_failed);
}

// CAP: the longest list the instance sorts in LDS; one workgroup per queue record.  CAP = 2048: 27 KB of LDS, five
// workgroups per CU, every queue of the usual step at once; CAP = 4096 (54 KB, two per CU) and CAP = BUCKET_MAX
// (104 KB, one per CU: 66 us instead of 36 for the usual step) when the LAST lists the host has seen were that long
// (a cloud whose surface implodes keeps a queue or two of the end ranks at 2 400-2 700 operations, N = 2^22 on
// eight ranks).  A wrong hint only costs time: whatever is longer than the instance's CAP takes replay_long.
//
// The same launch is the NEXT step's init_iframe (ps.cpp:1574-1606): nothing of the per-frame counts is read any
// more when this kernel runs (the replay reads rec_start, not the counts), so every workgroup zeroes a share of them,
// and the first one the frame scalars (the sticky error word stays) and a slab's status record header and census.
template <int CAP>
__global__ __launch_bounds__(REPLAY_THREADS) void k_replay_commit(DevParams P, int nrec, const int *__restrict__ rec_start,
                                                        uint64_t *keys, int *args, uint64_t *keys2, int *args2,
                                                        QueueInfo *qinfo, int *queue, MoveRec *moves,
                                                        DevCounters *ctr, StepState *stp,
                                                        ParticleArrays A, const float4 *__restrict__ stage,
                                                        int *frame, unsigned frame_ints, FrameScalars *fs, int *status_out, int status_table)
{
    static_assert(CAP == 2048 || CAP == 4096 || CAP == BUCKET_MAX, "three instances");
    __shared__ __attribute__((aligned(16))) unsigned char raw[ReplayLds<CAP>::SORT_BYTES];
    __shared__ unsigned char s_sub[CAP];
    __shared__ int wave_tot[2 * (REPLAY_THREADS / 64)];
    __shared__ int s_flag[3];
    __shared__ int s_cnt[MSD_LEVELS][256];
    __shared__ int s_cur[MSD_LEVELS], s_misc[2 * MSD_LEVELS];
    const int tid = threadIdx.x, rec = blockIdx.x;
    {   // init_iframe for the next step
        const unsigned per = (frame_ints + gridDim.x - 1) / gridDim.x, a = min(frame_ints, blockIdx.x * per), b = min(frame_ints, a + per);
        for (unsigned i = a + tid; i < b; i += REPLAY_THREADS) frame[i] = 0;
        if (rec == 0) {
            if (tid == 0) {
                // (a slab reports in its NEXT status record how many transfer records it sent this step: what the messages' capacity follows)
                stp->last_departures = max(fs->n_out[0], fs->n_out[1]);
                const int err = fs->error; *fs = FrameScalars{}; fs->error = err;
            }
            if (status_out) {
                for (int i = tid; i < MSG_HEADER_WORDS; i += REPLAY_THREADS) status_out[i] = 0;
                for (int i = tid; i < status_table; i += REPLAY_THREADS) status_out[STATUS_CHUNK_OFF + i] = 0;       // the (chunk, type) census
            }
        }
    }
    if (rec >= nrec) return;
    const int start = rec_start[rec], n = rec_start[rec + 1] - start;
    if (n <= 0) return;
    ReplayLds<CAP> L{reinterpret_cast<uint64_t *>(raw), reinterpret_cast<int *>(raw + ReplayLds<CAP>::KEY_BYTES), reinterpret_cast<int *>(raw),
                     s_sub, wave_tot, s_flag};
    if (n <= CAP) replay_short<CAP>(P, rec, start, n, keys, args, qinfo, queue, moves, ctr, stp, A, stage, L);
    else replay_long<CAP>(P, rec, start, n, keys, args, keys2, args2, qinfo, queue, moves, ctr, stp, A, stage, L, s_cnt, s_cur, s_misc);
}

// Relocation phase 1a on a slab, right after apply: the state of the departing particles goes into the outboxes
// (a record that leaves for a neighbour rank, MOVE_OUT, gets its state written into the outbox entry
// k_apply reserved), every local record is staged (read-only on the particle arrays, so a
// parent that also relocates this step is seen intact by both of its records), and the messages get their headers.
// Grid-stride over the step's move records.
__global__ void k_moves_stage(DevParams P, MoveRec *moves, int moves_cap, const FrameScalars *__restrict__ fs,
                              const float4 *pos4, const float4 *vel4, const float4 *acc4,
                              const uint8_t *pflags, float4 *stage, Outboxes out, OutboxMsgs msgs)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g < 5 && msgs.m[g]) {          // closing the outboxes: the headers of the relocation messages
        int *h = msgs.m[g];
        h[0] = min(fs->n_out[g], g < 2 ? P.xfer_cap : g < 4 ? P.xfer2_cap : P.far_cap); h[1] = 0; h[2] = fs->error;
        if (g == 4) h[3] = FAR_MAGIC;      // (the receivers take it off again: a far outbox that was not all-gathered this step is noticed)
    }
    const int n = min(fs->n_moves, moves_cap);
    for (int m = g; m < n; m += gridDim.x * blockDim.x) {
        const MoveRec r = moves[m];
        if (r.kind & MOVE_IN) continue;
        const int si = slot_index(P, r.src);
        if (r.kind & MOVE_OUT) {
            if (r.dst < 0) continue;                          // the outbox was full (error already raised)
            XferRec *x = out.o[(r.kind & MOVE_FAR) ? 4 : ((r.kind & MOVE_UP) ? 1 : 0) + ((r.kind & MOVE_HOP2) ? 2 : 0)] + r.dst;
            const float4 p = pos4[si], v = vel4[si], a = acc4[si];
            x->pos[0] = p.x; x->pos[1] = p.y; x->pos[2] = p.z; x->pos[3] = p.w;
            x->vel[0] = v.x; x->vel[1] = v.y; x->vel[2] = v.z; x->vel[3] = v.w;
            x->acc[0] = a.x; x->acc[1] = a.y; x->acc[2] = a.z; x->acc[3] = a.w;
            if ((r.kind & 0xff) == 0 && pflags[si]) x->kind |= MOVE_PARENT;
            continue;
        }
        float4 *s = stage + (size_t)3 * m;
        s[0] = pos4[si]; s[1] = vel4[si]; s[2] = acc4[si];
        if (r.kind == 0 && pflags[si]) moves[m].kind = MOVE_PARENT;  // is_parent travels in bit 8
    }
}

// how many 1024-thread (or 256-thread) workgroups a grid-stride pass over at most `items` items gets: enough to fill
// the chip for the usual step, never sized from the host's idea of the population (a bound read a step late)
static int stride_blocks(int64_t items, int per_block, int cap)
{
    return (int)std::max<int64_t>(1, std::min<int64_t>((items + per_block - 1) / per_block, cap));
}

hipError_t launch_outbox_close(hipStream_t st, const DevParams &P, const DeviceState &d, int64_t live_hint, int *const msgs[5])
{
    const int64_t max_moves = std::min<int64_t>(d.moves_cap, 2 * std::max<int64_t>(live_hint, 1));
    k_moves_stage<<<stride_blocks(max_moves, 256, 2048), 256, 0, st>>>(P, d.moves, d.moves_cap, d.fs, d.pos4, d.vel4, d.acc4, d.pflags, d.stage,
                                      Outboxes{{d.xfer_out[0], d.xfer_out[1], d.xfer_out[2], d.xfer_out[3], d.xfer_out[4]}}, OutboxMsgs{{msgs[0], msgs[1], msgs[2], msgs[3], msgs[4]}});
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

// The step's tail, three launches (see the head of this file).  `live_hint`: about how many particles the step has
// (the host's figure, a step or two old): it sizes the grids of the grid-stride passes and nothing else.
// `cap0`: the replay instance, from the longest list of the last step the host has seen.
hipError_t launch_lifecycle(hipStream_t st, const DevParams &P, const DeviceState &d, int nrec, int64_t live_hint, int cap0,
                            size_t frame_ints, int status_table)
{
    const int64_t hint = std::max<int64_t>(live_hint, 1);
    const int64_t max_ops = std::max<int64_t>(1, std::min<int64_t>(d.ops_cap, 3 * hint));
    const int nwg = stride_blocks(max_ops, SLOTS_PER_WG, 2048);
    const int nhist = std::min(nwg, 512);
    const int nstage = stride_blocks(std::min<int64_t>(d.moves_cap, 2 * hint), 1024, 256);
    k_ops_hist<<<nhist + nstage, 1024, 0, st>>>(P, nhist, d.op_keys, d.fs, d.ops_cap, P.key_rec_shift, nrec, d.rec_count,
                                                d.moves, d.moves_cap, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage);
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
    const ParticleArrays A{d.pos4, d.vel4, d.acc4, d.cell, d.pflags};
    // (at least a few dozen workgroups even for a handful of queue records: they also zero the next frame's counts)
    const int grid = std::max(nrec, 64);
#define PS_REPLAY(CAP) k_replay_commit<CAP><<<grid, REPLAY_THREADS, 0, st>>>(P, nrec, d.rec_start, d.op_keys_sorted, d.op_args_sorted, d.op_keys, d.op_args, \
        d.qinfo, d.queue, d.moves, d.ctr, d.st, A, d.stage, d.cell_count, (unsigned)frame_ints, d.fs, d.status_out, d.status_out ? status_table : 0)
    if (cap0 > 4096) PS_REPLAY(BUCKET_MAX); else if (cap0 > 2048) PS_REPLAY(4096); else PS_REPLAY(2048);
#undef PS_REPLAY
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace psamd
