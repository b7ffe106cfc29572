// lifecycle.hip -- free-slot queues and relocation in the reference's serial order (ps.cpp:1335-1374, app_common.cu:305-376)
#include "kernels_common.hpp"

namespace psamd {

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

// CAP: the longest list this instance holds in LDS; it serves the queues with more than `lo` and at most CAP
// operations.  The instance launched every step, one workgroup per queue, has CAP = 2048 (27 KB of LDS: five
// workgroups per CU, every queue of the usual step at once) -- or CAP = 4096 (54 KB, two per CU) when the LAST step's
// longest list lay between the two (the host's hint: a cloud whose surface implodes keeps a queue or two of the end
// ranks at 2 400-2 700 operations, N = 2^22 on eight ranks, and the long-list instance behind the usual one cost those
// ranks 55 us a step).  CAP = BUCKET_MAX (104 KB, one workgroup per CU) takes what is longer than that -- launched by
// the HOST only in a step whose scalars show such a list (launch_lifecycle part 1: h_fs->max_bucket > the step's
// first cap; the host's copy of the scalars is the device's, so its test agrees with the kernel's own backstop).
// (One instance sized for the longest list ran one workgroup per CU for every queue: 66 us instead of 36.)
template <int CAP>
__device__ __forceinline__ void replay_record(const DevParams &P, const int rec, const int *__restrict__ rec_start,
                                                        const uint64_t *__restrict__ keys,
                                                        const int *__restrict__ args,
                                                        QueueInfo *qinfo, int *queue, MoveRec *moves,
                                                        DevCounters *ctr, const FrameScalars *__restrict__ fs,
                                                        unsigned long long *trace,
                                                        float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                        float4 *stage, const int lo)
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
    static_assert(CAP == 2048 || CAP == 4096 || CAP == BUCKET_MAX, "three instances: the usual lists (two sizes), long lists");
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
    if (n == 0 || n <= lo || n > CAP) return;           // (another instance's)
    QueueInfo q = qinfo[rec];
    const bool in_lds = q.seg_size <= WINDOW_SLOTS;
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

// The instance for the usual lists (PER_RECORD) runs one workgroup per queue record (and the first relocation phase
// in the workgroups past them); the one for long lists -- launched only in a step that has one, see above --
// strides over the records with a few workgroups.
template <int CAP, bool PER_RECORD>
__global__ __launch_bounds__(REPLAY_THREADS) void k_replay_bucket(DevParams P, int nrec, const int *__restrict__ rec_start,
                                                        const uint64_t *__restrict__ keys, const int *__restrict__ args,
                                                        QueueInfo *qinfo, int *queue, MoveRec *moves,
                                                        DevCounters *ctr, const FrameScalars *__restrict__ fs,
                                                        unsigned long long *trace,
                                                        float4 *pos4, float4 *vel4, float4 *acc4, int *cell_arr, uint8_t *pflags,
                                                        float4 *stage, int lo)
{
    if (PER_RECORD) {
        if ((int)blockIdx.x >= nrec) {
            moves_stage_reset(P, ((int)blockIdx.x - nrec) * REPLAY_THREADS + (int)threadIdx.x, moves, fs, pos4, vel4, acc4, cell_arr, pflags, stage);
            return;
        }
        replay_record<CAP>(P, (int)blockIdx.x, rec_start, keys, args, qinfo, queue, moves, ctr, fs, trace, pos4, vel4, acc4, cell_arr, pflags, stage, lo);
        return;
    }
    if (fs->max_bucket <= lo) return;
    for (int rec = blockIdx.x; rec < nrec; rec += gridDim.x) {
        replay_record<CAP>(P, rec, rec_start, keys, args, qinfo, queue, moves, ctr, fs, trace, pos4, vel4, acc4, cell_arr, pflags, stage, lo);
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

// part 0: the replay of the usual lists (with the first relocation phase), enqueued without waiting for the host --
// lists of up to `cap0` operations, 2048 or 4096 (the host's hint from the last step);
// part 1, once the host has the step's scalars (they are out before part 0 starts running): the instance for
// long lists only if some queue got more than cap0 operations (`long_lists`), and the commit.  (The long-list
// instance used to be launched every step and leave at once: ~4.5 us on the timeline for nothing.)
hipError_t launch_lifecycle(hipStream_t st, const DevParams &P, const DeviceState &d, int nrec, int64_t live_bound, int part, bool long_lists, int cap0)
{
    const int64_t max_moves = std::min<int64_t>(d.moves_cap, 2 * live_bound);
    const int nb = (int)((max_moves + REPLAY_THREADS - 1) / REPLAY_THREADS);
    if (part == 0) {
        if (cap0 > 2048)
            k_replay_bucket<4096, true><<<nrec + nb, REPLAY_THREADS, 0, st>>>(P, nrec, d.rec_start, d.op_keys_sorted, d.op_args_sorted, d.qinfo, d.queue,
                                                  d.moves, d.ctr, d.fs, d.trace, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage, 0);
        else
            k_replay_bucket<2048, true><<<nrec + nb, REPLAY_THREADS, 0, st>>>(P, nrec, d.rec_start, d.op_keys_sorted, d.op_args_sorted, d.qinfo, d.queue,
                                                  d.moves, d.ctr, d.fs, d.trace, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage, 0);
        PS_LAUNCH_CHECK();
        return hipSuccess;
    }
    if (long_lists) {
        k_replay_bucket<BUCKET_MAX, false><<<std::min(nrec, 256), REPLAY_THREADS, 0, st>>>(P, nrec, d.rec_start, d.op_keys_sorted, d.op_args_sorted, d.qinfo, d.queue,
                                              d.moves, d.ctr, d.fs, d.trace, d.pos4, d.vel4, d.acc4, d.cell, d.pflags, d.stage, cap0 > 2048 ? 4096 : 2048);
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
