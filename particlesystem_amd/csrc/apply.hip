// apply.hip -- calc_forces' tail per particle: kill / survive / integrate / wrap / explosion (ps.cpp:1210-1333)
#include "kernels_common.hpp"

namespace psamd {

// ------------------------------------------------------------------ apply
// Death, survival, integration, wrap and re-hash for every particle of the frame
// (ps.cpp:1182-1242, 1261-1302), one thread per owned SLOT so that the particle arrays stream
// through coalesced (live slots are dense at the head of every segment) -- the step's force too: the pair stage
// left it where the particle keeps its acceleration (acc4.xyz; the collision flag in a byte beside it: ForceBuf).  Lifecycle side
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
                                                const uint8_t *__restrict__ flag_slot,
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

    int flag = 0, new_cell = 0;
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
    // the particle's own state and its force record, one batch of loads (all by slot)
    float4 p = f, v = f;
    float fert = 0.f;
    uint8_t pf0 = 0;
    if (active) { p = pos4[si]; v = vel4[si]; f = acc4[si]; pf0 = pflags[si]; flag = (int)flag_slot[si]; fert = f.w; }
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
        uint8_t pf = pf0;
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
        if (P.drag > 0.f) acc4[si] = make_float4(axv, ayv, azv, fert);      // (else the record holds it already: the pair stage put it there)
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

hipError_t launch_apply(hipStream_t st, const DevParams &P, const SegLayout &S, const DeviceState &d)
{
    if (P.slots_total <= 0) return hipSuccess;
    // slots per thread: one (PSAMD_APPLY_ITEMS: the measurement quoted at the kernel)
#define PS_APPLY(I) k_apply<I><<<(P.slots_total + I * 1024 - 1) / (I * 1024), 1024, 0, st>>>(P, S, d.st, d.flag_slot, d.pos4, \
        d.vel4, d.acc4, d.cell, d.pflags, d.celltab, d.op_keys, d.op_args, d.ops_cap, \
        d.moves, d.moves_cap, Outboxes{{d.xfer_out[0], d.xfer_out[1], d.xfer_out[2], d.xfer_out[3], d.xfer_out[4]}}, d.chunk_count, d.chunk_skip, d.fs, d.ctr)
    static const int items_env = std::getenv("PSAMD_APPLY_ITEMS") ? std::atoi(std::getenv("PSAMD_APPLY_ITEMS")) : 0;
    const int items = items_env ? items_env : 1;
    if (items >= 4) PS_APPLY(4); else if (items >= 2) PS_APPLY(2); else PS_APPLY(1);
#undef PS_APPLY
    PS_LAUNCH_CHECK();
    return hipSuccess;
}

}  // namespace psamd
