// device_types.h -- plain structs shared by the host context and the HIP kernels.
#pragma once

#include <cstdint>

namespace psamd {

// Constants every kernel needs, passed by value as a kernel argument.
struct DevParams {
    int32_t G;              // grid_dim
    int32_t num_cells;
    int32_t num_chunks;
    int32_t container;      // slots
    int32_t max_per_cell;   // cell list capacity (MAX_PARTICLES_PER_CELL)
    int32_t max_per_chunk;
    int32_t slices;         // ceil(max_per_cell / 64): wave tasks per cell
    uint32_t flags;         // PSAMD_FLAG_*
    float t;                // (float)DT, ps.cpp:1272
    float kid_thr;          // (double)age <  KID_AGE       <=> age < kid_thr
    float life_thr;         // (double)age >  PARTICLE_LIFE <=> age > life_thr
    float coll_d2_gate;     // pairs with d2 <= gate get the exact collision test
    float dmax;             // (float)MAX_DX, ps.cpp:1278
    float vmax;             // (float)MAX_V,  ps.cpp:1293
    float w_default;        // (float)PARTICLE_WEIGHT_DEFAULT
    float fert_lo, fert_hi; // (float)MIN/MAX_FERTILITY_AGE as get_random_number_h receives them
    double cell_size;       // CELL_SIZE
    double eps2;            // EPS2 (double literal in the reference)
    double coll_radius;     // COLLISION_RADIUS (double literal)
    double kid_age;         // exact double thresholds for the rare slow path
    double life;
    double expl_speed;      // EXPLOSION_SPEED
    uint64_t seed;
    // layout of a queue-op sort key: | record | chunk+1 | slot id | sub-step (2 bits) |
    int32_t key_chunk_shift; // 2 + bits(container)
    int32_t key_rec_shift;   // key_chunk_shift + bits(num_chunks + 1)
    int32_t key_bits;        // key_rec_shift + bits(queue records)
    int32_t lean_math;       // 1: the lean exact sqrt/rcp are valid for this box (host-checked range)
    // (float)((double)d2 + eps2) == d2 + eps2f for every float d2 in [eps_f32_from, d2_max]
    // (verified exhaustively on the device at context creation); below it the add is done in double
    float eps2f;
    float eps_f32_from;
    float slow_below;        // max(eps_f32_from, just above coll_d2_gate): closer pairs take the slow branch
    float halo_reach;        // bodies this close to a cell face are collision candidates of the cell beyond it
    float coll_d2_max;       // (double)sqrtf(d2) > COLLISION_RADIUS  <=>  d2 > coll_d2_max (bisection at creation)
    int32_t two_pass;        // 1: collision flags first (k_collide_cell), forces only for the particles that move
    float pad_f;
    // ---- slab partition (partition.hpp); world == 1: one region, one slot range, the identity ----
    // Cells are addressed by LOCAL index: four regions of whole cell layers, each followed by one
    // empty "gap" cell so that cell_start stays a plain prefix array although every region has
    // its own fixed block of the sorted arrays.  0: layers whose particles live here, 1: halo
    // layer from the rank below, 2: layers lent by the rank below (computed here), 3: halo layer
    // from the rank above.
    int32_t reg_first[4];    // first global cell layer (i3) of the region
    int32_t reg_layers[4];   // number of layers (0: region absent)
    int32_t reg_base[4];     // local index of the region's first cell
    int32_t reg_sorted[4];   // first sorted index of the region's block
    int32_t n_local_cells;   // all regions and their gap cells
    int32_t n_own_cells;     // cells of region 0
    int32_t sorted_cap;      // capacity of the sorted-order arrays = plane stride of snap_soa
    int32_t own_comp0, own_comp1;   // own local cells this rank computes collisions / forces for (a contiguous run of region 0)
    // the cell ranges ONE pass of the pair stage works on, in work-list order (lent cells first:
    // their results travel back).  The whole stage: {lent, own computed, -}; cut in two to overlap
    // the halo exchange: {interior own cells} before the halo has arrived, then {lent, own cells
    // below the interior, own cells above it}.
    int32_t comp_lo[3], comp_hi[3];
    int32_t slot_lo[4], slot_n[4];   // owned slot range per segment type; storage index = position in their concatenation
    int32_t slots_total;
    int32_t rec_lo[4], rec_hi[4];    // owned QUEUE_INFO records per segment type
    int32_t rank, world;
    int32_t halo_cap_cell;   // bodies per cell, on average over a cell layer, a halo message has room for (pooled)
    int32_t xfer_cap;        // relocation records per direction and step a transfer message carries NOW (grows on demand, all ranks together)
    int32_t xfer_cap_max;    // ... and at most: the room of its buffers
    int32_t xfer_cap0;       // ... and at least: what the context was created with
    int32_t lentout_c0, lentout_c1;  // own local cells computed by the rank above (their force records come back)
    int32_t num_cells_global;
    float drag;              // linear drag coefficient (0: the reference's arithmetic)
    float force_sign;        // +1 gravity, -1 repulsion
    // all-pairs forces across ranks: every rank's own snapshot, all-gathered once per step, as blocks of
    // allg_block words: 16 header | allg_cells cell counts | 4 planes (x, y, z, w_eff) of allg_cap floats
    int32_t allg_cells, allg_cap, allg_block;
    int32_t status_words;    // words of one rank's status record
    // the ring neighbours' QUEUE_INFO records (0: the rank below, 1: above): which of them a departing record is for
    int32_t nbr_rec_lo[2][4], nbr_rec_hi[2][4];
    int32_t xfer2_cap;       // records per step and direction that may go TWO ranks away (0: no rank of this world can be flown over)
    int32_t far_cap;         // records per step that may go to a rank further away still, through the all-gathered far outbox (0: none)
};


// All-pairs force pass.  The cells beyond the stencil are found by GLOBAL cell in a snapshot buffer (the own
// one, or the all-gathered one of all ranks): x at buf[start], y, z, w_eff at multiples of `plane` behind.
// The stencil's chain comes from the ordinary (cutoff) force pass; the rest of a particle's sum is split into
// ALLP_PARTS partial sums, each by a wave of its own (or a lone rank of eight would have half a wave per SIMD
// walking the whole cloud): part p covers its share of the 64-cell blocks of the global cell order; partial sums
// land in part_acc[p * part_plane + r], r = the particle's place among those that need a force, and
// k_allpairs_combine adds them to the stencil's chain in part order (pairs.hip, k_allp_far).
struct FarCells {
    unsigned long long plane = 0;
    float4 *part_acc = nullptr;
    unsigned long long part_plane = 0;
};
constexpr int ALLP_PARTS = 16;

// Which cells / slots / records a rank holds.  All device code goes through these.
#if defined(__HIPCC__)
#define PS_HD __host__ __device__ __forceinline__
#else
#define PS_HD inline
#endif

// local index of global cell (i3, i1, i2), or -1 if this rank does not hold its layer
PS_HD int local_cell(const DevParams &P, int i3, int i1, int i2)
{
    if (i1 < 0 || i1 >= P.G || i2 < 0 || i2 >= P.G) return -1;
    for (int r = 0; r < 4; r++) {
        const int l = i3 - P.reg_first[r];
        if (l >= 0 && l < P.reg_layers[r]) return P.reg_base[r] + (l * P.G + i1) * P.G + i2;
    }
    return -1;
}

PS_HD int local_of_global(const DevParams &P, int gc)
{
    const int GG = P.G * P.G, i3 = gc / GG, rem = gc - i3 * GG;
    for (int r = 0; r < 4; r++) {
        const int l = i3 - P.reg_first[r];
        if (l >= 0 && l < P.reg_layers[r]) return P.reg_base[r] + l * GG + rem;
    }
    return -1;
}

// (i1, i2, i3) of a local cell (not a gap cell)
PS_HD void cell_coords(const DevParams &P, int lc, int &i1, int &i2, int &i3)
{
    int r = 0;
    for (int k = 1; k < 4; k++) if (P.reg_layers[k] > 0 && lc >= P.reg_base[k]) r = k;
    const int rel = lc - P.reg_base[r], GG = P.G * P.G, l = rel / GG, rem = rel - l * GG;
    i3 = P.reg_first[r] + l; i1 = rem / P.G; i2 = rem - i1 * P.G;
}

PS_HD int global_of_local(const DevParams &P, int lc)
{
    int i1, i2, i3;
    cell_coords(P, lc, i1, i2, i3);
    return (i3 * P.G + i1) * P.G + i2;
}

// j-th cell of the pass's ranges
PS_HD int comp_cell(const DevParams &P, int j)
{
    const int n0 = P.comp_hi[0] - P.comp_lo[0], n1 = P.comp_hi[1] - P.comp_lo[1];
    return j < n0 ? P.comp_lo[0] + j : j < n0 + n1 ? P.comp_lo[1] + (j - n0) : P.comp_lo[2] + (j - n0 - n1);
}
PS_HD int comp_count(const DevParams &P)
{
    return (P.comp_hi[0] - P.comp_lo[0]) + (P.comp_hi[1] - P.comp_lo[1]) + (P.comp_hi[2] - P.comp_lo[2]);
}

// storage index of an owned slot (position in the concatenation of the four owned ranges), -1 if not owned
PS_HD int slot_index(const DevParams &P, int slot)
{
    int off = 0;
    for (int t = 0; t < 4; t++) {
        const int d = slot - P.slot_lo[t];
        if (d >= 0 && d < P.slot_n[t]) return off + d;
        off += P.slot_n[t];
    }
    return -1;
}

// inverse: the slot stored at index idx (idx < slots_total)
PS_HD int slot_of_index(const DevParams &P, int idx)
{
    for (int t = 0; t < 3; t++) {
        if (idx < P.slot_n[t]) return P.slot_lo[t] + idx;
        idx -= P.slot_n[t];
    }
    return P.slot_lo[3] + idx;
}

PS_HD bool nbr_owns_record(const DevParams &P, int which, int rec)      // which: 0 the rank below, 1 the rank above
{
    for (int t = 0; t < 4; t++) if (rec >= P.nbr_rec_lo[which][t] && rec < P.nbr_rec_hi[which][t]) return true;
    return false;
}

PS_HD bool owns_record(const DevParams &P, int rec)
{
    for (int t = 0; t < 4; t++) if (rec >= P.rec_lo[t] && rec < P.rec_hi[t]) return true;
    return false;
}

// Per-frame scalars living in device memory (zeroed by init_iframe).
struct FrameScalars {
    int32_t gridmax[2];     // hostGridMax: biggest chunk, biggest cell (ps.cpp:76)
    int32_t live;           // particles with a valid cell at build_grid
    int32_t error;          // sticky bit mask, see ERR_* below
    int32_t n_ops;          // queue operations emitted by apply   \ allocated together as one
    int32_t n_moves;        // relocation / birth records emitted  / 64-bit word (ops low)
    int32_t max_bucket;     // most queue operations any one segment received this step
    int32_t n_tasks;        // non-empty (cell, slice) tasks of the pair kernel this frame
    int32_t n_tasks2;       // two-pass mode: (cell, 64-slice) tasks over the particles that need a force
    int32_t n_merged;       // ... and merged tasks (up to four cells' partly filled last slices in one wave)
    int32_t n_out[5];       // slab mode: relocation / birth records leaving for the rank below [0] / above [1], two ranks below [2] / above [3], any other rank [4] (the all-gathered far outbox)
    int32_t n_lent;         // slab mode: bodies in the lent-in region this frame
    int32_t chunk_over;     // a chunk's count passed MAX_PARTICLES_PER_CHUNK this frame: the tail of its list is skipped (k_chunk_cap)
    int32_t status_error;   // slab mode: OR of the error bits in this step's all-gathered status records (every rank sees the same word)
    int32_t seq;            // host copy only: the number of the step whose scalars these are, written last (the host polls it)
    int32_t max_cell_raw;   // most ids any own cell received this frame, uncapped (gridmax[1] is capped at the list capacity)
    int32_t xfer_cap_next;  // slab mode: the transfer messages' capacity every rank adopts two steps on (k_status_merge: the same number on every rank)
    int32_t pad_fs;
    long long cost_total;   // two-pass mode: sum over the force pass's tasks of the bodies each walks (its stencil's population)
};

// What a step needs to know about its own number, kept on the device so that no kernel argument changes from
// one step to the next (a captured hipGraph replays the arguments it was captured with): `step` keys the
// explosion RNG (k_apply, k_replay_commit's commit_move), `seq` counts the scalar records handed to the host.  The workgroup
// that publishes a step's scalars raises `pending`; the next frame's first kernel (k_hist_lds) -- nothing reads `step`
// while it runs -- turns that into step + 1.  snapshot_restore rewinds `step` (k_restore).
struct StepState {
    int32_t step, pending, seq;
    int32_t last_departures;   // slab mode: most records this rank sent in one direction in the step before (goes out with the next status record)
    int32_t peak_prev, pad_st; // slab mode: the busiest rank's count in the status records of the step before (the same number on every rank)
    // The balanced force pass paces its waves against the clock (pairs.hip, WavePace): per pass of a frame (0 / 1), when
    // the pass's planning ended (100 MHz real-time counter), when its last wave ended, and how long the last such pass
    // took -- what this one expects to take.
    unsigned long long pairs_t0[2], pairs_end[2];
    int32_t pairs_ticks[2];
};

// Cumulative event counters, mirrors psamd_counters.  Kept in COUNTER_COPIES copies on
// separate 128-byte lines (workgroup b adds to copy b % COUNTER_COPIES; the host sums
// them): same-line atomics are served one at a time by the memory side.
struct alignas(128) DevCounters {
    unsigned long long deaths_age, deaths_collision, survives, integrated;
    unsigned long long relocations, relocations_lost, births, births_failed, cell_overflow_kills;
};
constexpr int COUNTER_COPIES = 64;

enum : int32_t {
    ERR_CELL_TOO_BIG = 1,   // (not raised since round 5: k_sort_cells ranks a cell of any size, through global memory beyond its LDS room; the bit and its message stay for ABI stability)
    ERR_BAD_ID = 2,         // uploaded P_DATA_TYPE with id != slot
    ERR_OPS_OVERFLOW = 4,   // lifecycle op buffer too small
    ERR_BAD_POS = 16,       // uploaded live particle outside the box (or cell out of range)
    ERR_FOREIGN_CELL = 32,  // slab mode: a particle stored here sits in a layer this rank holds no state for
    ERR_HALO_OVERFLOW = 64, // slab mode: a halo / force / relocation message had no room for what it must carry
    ERR_SLAB_MISMATCH = 128,// slab mode: a message disagrees with the receiver's own counts
    ERR_REMOTE_RECORD0 = 256,// slab mode: a cell-overflow kill on a rank that does not own queue record 0
    ERR_CHUNK_CAP = 512,    // a chunk list passed MAX_PARTICLES_PER_CHUNK (the reference would skip its tail)
    ERR_HANDOFF_TIMEOUT = 1024, // force pass: a wave never saw the partial sums of the task it continues (should be impossible)
};

// A free-slot-queue operation produced by calc_forces is a (key, arg) pair kept in
// two parallel arrays: key = record | chunk+1 | id | sub orders the operations of
// one queue exactly as the reference's serial loops meet them (chunk by chunk, slot
// by slot; within a particle: birth remove (0), relocation remove (1), insert (2));
// arg = slot id to free (insert) or index of the MoveRec waiting for a slot (remove).
// A particle that needs a new slot (segment change) or a child to be born.
struct MoveRec {
    int32_t src;            // slot of the particle (parent for births)
    int32_t dst;            // filled in by the queue replay: new slot or -1 (MOVE_OUT: index in the outbox)
    int32_t kind;           // low byte: 0 relocation, 1 birth; flags below
    int32_t new_cell;
};
constexpr int MOVE_PARENT = 0x100;   // the relocating particle has is_parent set
constexpr int MOVE_IN = 0x200;       // arrived from a neighbour rank: state already in the staging area, no local source
constexpr int MOVE_OUT = 0x400;      // leaves for a neighbour rank (whose queue hands out the slot); | MOVE_UP: the rank above
constexpr int MOVE_UP = 0x800;
constexpr int MOVE_HOP2 = 0x1000;    // ... two ranks away (a two-layer jump over a rank whose state is a single layer)
constexpr int MOVE_FAR = 0x2000;     // ... any rank further away: the record goes into the far outbox, which every rank receives

// One particle on its way to a segment another rank owns (relocation or birth): the queue
// operation's key in the reference's serial order, and the state to place.  64 bytes.
struct XferRec {
    uint64_t key;           // record | chunk + 1 | source slot | sub-step, as in the op lists
    int32_t new_cell;       // global cell
    int32_t kind;           // 0 relocation (| MOVE_PARENT), 1 birth
    float pos[4], vel[4], acc[4];   // relocation: the particle; birth: the parent's position and velocity
};
// Status record of a slab, all-gathered once per step (it must have landed before slab_apply): 16 header
// words ([0] cell-overflow kills, [1] sticky error bits at the end of the build stage, [2] live, [3] the most transfer records it
// sent in one direction in the step before: what the transfer messages' capacity is adapted to), the
// killed slot ids, then -- per chunk and segment type -- how many of the chunk's particles live in this
// rank's segments of that type (the chunk lists' capacity rule ranks a chunk's particles in slot
// order, and a chunk's 27 segments are spread over up to three ranks).
constexpr int STATUS_KILL_CAP = 4080;   // cell-overflow kills one rank can report per step
constexpr int STATUS_CHUNK_OFF = 16 + STATUS_KILL_CAP;     // first word of the (chunk, type) table; the record is STATUS_CHUNK_OFF + 4 * num_chunks words
constexpr int MSG_HEADER_WORDS = 16;   // every message starts with 16 ints: [0] count, [1] bodies, [2] error bits

constexpr int SORT_MAX = 4096;   // ids one cell may hold for the in-LDS ranking
constexpr int REPLAY_CHUNK = 2048;   // queue ops staged through LDS at a time
constexpr int QUEUE_WINDOW = 6144;   // largest segment (slots) whose queue is replayed in LDS
constexpr int BUCKET_MAX = 8192;     // ops per segment the one-workgroup fast replay sorts in LDS (104 of the CU's 160 KB)
constexpr int MAX_PAIR_WAVES = 16384;    // wave slots of the balanced force pass at most (seven resident per SIMD: 7168; more: the later ones start as the first ones end)
constexpr int STENCIL = 27;          // cells a particle's force walk visits, in the reference's order (app.cu:370-409)
constexpr int HALO_CAP = 768;        // collision candidates one cell can list from its neighbours (else: full stencil)

}  // namespace psamd
