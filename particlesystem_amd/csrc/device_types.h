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
    int32_t two_pass;        // 1: collision flags first (k_collide), forces only for the particles that move
    float pad_f;
};

// Per-frame scalars living in device memory (zeroed by init_iframe).
struct FrameScalars {
    int32_t gridmax[2];     // hostGridMax: biggest chunk, biggest cell (ps.cpp:76)
    int32_t live;           // particles with a valid cell at build_grid
    int32_t error;          // sticky bit mask, see ERR_* below
    int32_t n_ops;          // queue operations emitted by apply   \ allocated together as one
    int32_t n_moves;        // relocation / birth records emitted  / 64-bit word (ops low)
    int32_t max_bucket;     // most queue operations any one segment received this step
    int32_t n_tasks;        // non-empty (cell, slice) tasks of the pair kernel this frame
    int32_t shard_task_lo;  // first pair-kernel task (cell * slices) of this rank's share
    int32_t shard_task_n;   // number of tasks from there that can hold a particle of the share
    int32_t n_tasks2;       // two-pass mode: (cell, 64-slice) tasks over the particles that need a force
    int32_t n_merged;       // ... and merged tasks (up to four cells' partly filled last slices in one wave)
    int32_t shard_cell_lo, shard_cell_hi;   // cells that hold this rank's share of the sorted particles
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
    ERR_CELL_TOO_BIG = 1,   // a cell holds more ids than the sort kernel can rank
    ERR_BAD_ID = 2,         // uploaded P_DATA_TYPE with id != slot
    ERR_OPS_OVERFLOW = 4,   // lifecycle op buffer too small
    ERR_SHARD_BOUND = 8,    // more sorted particles than world * share
    ERR_BAD_POS = 16,       // uploaded live particle outside the box (or cell out of range)
};

// A free-slot-queue operation produced by calc_forces is a (key, arg) pair kept in
// two parallel arrays: key = record | chunk+1 | id | sub orders the operations of
// one queue exactly as the reference's serial loops meet them (chunk by chunk, slot
// by slot; within a particle: birth remove (0), relocation remove (1), insert (2));
// arg = slot id to free (insert) or index of the MoveRec waiting for a slot (remove).
// A particle that needs a new slot (segment change) or a child to be born.
struct MoveRec {
    int32_t src;            // slot of the particle (parent for births)
    int32_t dst;            // filled in by the queue replay: new slot or -1
    int32_t kind;           // 0 relocation, 1 birth
    int32_t new_cell;
};

constexpr int SORT_MAX = 4096;   // ids one cell may hold for the in-LDS ranking
constexpr int REPLAY_CHUNK = 2048;   // queue ops staged through LDS at a time
constexpr int QUEUE_WINDOW = 6144;   // largest segment (slots) whose queue is replayed in LDS
constexpr int BUCKET_MAX = 2048;     // ops per segment the one-workgroup fast replay sorts in LDS
constexpr int HALO_CAP = 768;        // collision candidates one cell can list from its neighbours (else: full stencil)

}  // namespace psamd
