/*
 * psamd.h -- C ABI of the MI355X-native particle-system step ("psamd").
 *
 * Drop-in boundary for the per-step hot path of abraj/particleSystem: the three
 * stage bodies the reference registers with pmlib (task 3 init_iframe, task 8
 * build_grid, task 6 calc_forces; DoParallelProcess loop, particleSystem.cpp
 * 1843-1928) plus the one-off setup stages that create their inputs.  Plain
 * pointers and sizes only; every function returns a psamd_status and never
 * exits or throws across the boundary (the reference printf+exit(1)s instead,
 * particleSystem.cpp:937-938, app.cu:429-431).
 *
 * Citations: "ps.cpp" = source/code/src/particleSystem.cpp, "psCUDA.cu" =
 * source/code/src/particleSystemCUDA.cu, the rest under source/code/inc/.
 *
 * Ownership: the library owns all device memory and its HIP streams.  Host
 * buffers passed in or out belong to the caller and are only touched during
 * the call.  One context is used by one host thread at a time (the reference's
 * driver thread blocks in pmWaitForTaskCompletion the same way, ps.cpp:1716).
 */
#ifndef PSAMD_H
#define PSAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSAMD_ABI_VERSION 6

#define PSAMD_MAX_RANKS 64

typedef enum psamd_status {
    PSAMD_OK = 0,
    PSAMD_ERR_INVALID_ARG   = 1,  /* null pointer, out-of-range index, bad config          */
    PSAMD_ERR_NO_DEVICE     = 2,  /* no usable HIP device: the product has no CPU fallback */
    PSAMD_ERR_HIP           = 3,  /* a HIP runtime call failed (see psamd_last_error)      */
    PSAMD_ERR_OUT_OF_MEMORY = 4,
    PSAMD_ERR_OUTSIDE_BOX   = 5,  /* fill: "Particle location OUTSIDE box", ps.cpp:954-957 */
    PSAMD_ERR_QUEUE_EMPTY   = 6,  /* fill: "Overflow (Reserved space full)", ps.cpp:936-939*/
    PSAMD_ERR_CELL_OVERFLOW = 7,  /* a cell holds more than MAX_PARTICLES_PER_CELL and the
                                     overflow policy is PSAMD_OVERFLOW_ERROR               */
    PSAMD_ERR_STATE         = 8,  /* stage called out of order (e.g. calc before build)    */
    PSAMD_ERR_UNSUPPORTED   = 9
} psamd_status;

/* config.flags */
#define PSAMD_FLAG_EXPLOSIONS   0x1u  /* births enabled (ps.cpp:1306-1333) with the counter-based RNG below */
#define PSAMD_FLAG_FAST_MATH    0x2u  /* FMA/rsq pair arithmetic: NOT bit-identical to the reference, see DESIGN.md */
#define PSAMD_FLAG_ALL_PAIRS    0x4u  /* force walk over EVERY cell, not only the 27-cell stencil (the reference has only the
                                         cutoff, app.cu:352-452): the stencil first, in the reference's order, then the other
                                         cells in index order -- so a cloud that fits a 2x2x2 block of cells gets the cutoff
                                         result bit for bit.  Collisions stay short-range.  With world > 1 every rank
                                         contributes the snapshot of its own cells to an all-gather once per step
                                         (allg_out -> allg_in below) and walks the gathered buffer in the same order. */
#define PSAMD_FLAG_EULER        0x8u  /* position update x += v*dt (explicit Euler) instead of the reference's
                                         x += v*dt + 0.5*a*dt*dt (ps.cpp:1274-1276); the velocity update is the same */

/* Runtime form of the reference's compile-time configuration, common.h:12-70.
 * psamd_default_config() fills in the shipped values. */
typedef struct psamd_config {
    int32_t  max_particles_num;  /* MAX_PARTICLES_NUM        common.h:12 */
    int32_t  x_factor;           /* X_FACTOR                 common.h:13 */
    int32_t  chunk_factor;       /* CHUNK_FACTOR             common.h:29 */
    int32_t  chunk_dim;          /* CHUNK_DIM                common.h:30 */
    double   cell_size;          /* CELL_SIZE                common.h:52 */
    double   eps2;               /* EPS2                     common.h:53 */
    double   collision_radius;   /* COLLISION_RADIUS         common.h:54 */
    double   particle_weight;    /* PARTICLE_WEIGHT_DEFAULT  common.h:55 */
    double   dt;                 /* DT                       common.h:69 */
    double   max_v;              /* MAX_V                    common.h:66 */
    double   explosion_speed;    /* EXPLOSION_SPEED          common.h:67 */
    double   life_steps;         /* the 300 in PARTICLE_LIFE common.h:58 */
    int32_t  device;             /* HIP device ordinal                    */
    uint32_t flags;              /* PSAMD_FLAG_*                          */
    uint64_t seed;               /* explosion RNG seed (RAND_SEED, common.h:56) */
    /* Multi-GPU: world > 1 makes this context ONE SLAB of the system (see "slab partition"
     * below): it holds only the segments of its cell layers -- their slots, particles and
     * free-slot queues -- and steps through the psamd_slab_* stage calls.  cuts[0..world]
     * (cell-layer boundaries along i3, cuts[0] = 0, cuts[world] = grid_dim, at least two
     * layers per rank) overrides the balanced partition when cuts[world] != 0. */
    int32_t  rank;
    int32_t  world;
    int32_t  halo_cap_cell;      /* bodies per cell, ON AVERAGE OVER A CELL LAYER, a halo message has room for (any one cell up to its list
                                    capacity: the room is pooled); 0 = MAX_PARTICLES_PER_CELL (never overflows) */
    int32_t  xfer_cap;           /* records a transfer message carries per step and direction (particles changing owner) TO BEGIN WITH;
                                    0 = a quarter of what a cell layer can hold.  The ranks raise it together when the traffic asks for
                                    it and lower it again, never below this number, when the traffic has gone (see xfer_cap_max) */
    int32_t  cuts[PSAMD_MAX_RANKS + 1];
    /* Not in the reference (BASELINE.json asks for them; nothing there can pin them): */
    double   drag;               /* linear drag k >= 0: the acceleration that is integrated and stored is a - k*v; 0 = the
                                    reference's arithmetic, untouched                                          */
    double   force_sign;         /* +1 gravity (reference), -1 repulsion: multiplies every mass in the force term; 0 reads as +1 */
    int32_t  xfer_cap_max;       /* how far the transfer messages may grow: their BUFFERS have this room from the start, the bytes that
                                    travel are xfer_cap's and follow the traffic.  Every rank reports in its status record how many records it
                                    sent in the step before; all ranks see all records and apply the same rule -- twice (the busiest rank's
                                    count + four times its rise since the step before), from the step after next; down to that number again
                                    when it is no more than half of what the messages hold -- so both
                                    ends of every message change size in the same step with no negotiation round (psamd_slab_buffers_get().xfer_bytes is the size to post,
                                    read it after psamd_slab_apply).  0 = what two cell layers and their children can hold (a step's
                                    worst case: the reference ships whole segments, ps.cpp:431-487); < xfer_cap: no growth. */
    int32_t  reserved0;
} psamd_config;

/* Sizes DoInit derives (ps.cpp:2204-2222), in elements. */
typedef struct psamd_sizes {
    int32_t grid_dim, num_cells, num_chunks, cells_per_chunk;
    int32_t max_per_cell, max_per_chunk;
    int32_t container_size;      /* nParticles = nTdata = nQueue        */
    int32_t queue_info_size;     /* nQueueInfo                           */
    int64_t n_chunkgrid;         /* NUM_CHUNKS*(1+MAX_PARTICLES_PER_CHUNK) */
    int64_t n_cellgrid;          /* NUM_CELLS*(1+MAX_PARTICLES_PER_CELL)   */
    int32_t n_pkgdistrib;        /* NUM_CHUNKS*27 PAIRs                  */
    int32_t seg_count[4], seg_size_t[4], seg_size[4]; /* types 1,2,4,8  */
} psamd_sizes;

/* Event counts of the last psamd_calc_forces / psamd_step call(s), cumulative. */
typedef struct psamd_counters {
    int64_t deaths_age, deaths_collision, survives, integrated;
    int64_t relocations, relocations_lost, births, births_failed, cell_overflow_kills;
    int64_t steps;
    int64_t particles_processed; /* sum over steps of the live particles at build_grid */
    int64_t max_ops_one_queue;   /* most free-slot-queue operations one segment got in one step */
} psamd_counters;

/* Raw device pointers of the SoA state, for plumbing (collectives, interop).
 * Valid until psamd_destroy.  Layouts are described in DESIGN.md section 3. */
typedef struct psamd_device_view {
    /* slot arrays hold the OWNED slots only, back to back (world == 1: the whole container) */
    void    *pos4;        /* float4[container]  x,y,z,w                         */
    void    *vel4;        /* float4[container]  vx,vy,vz,age                    */
    void    *acc4;        /* float4[container]  ax,ay,az,fertility_age          */
    void    *cell;        /* int[container]     cell index, -1 = free slot      */
    void    *pflags;      /* uint8[container]   bit0 = is_parent                */
    void    *sorted_id;   /* int[container]     slot ids, cell-major, id-ascending in a cell */
    void    *snap_soa;    /* float[4][sorted_cap] snapshot in sorted order: x, y, z, w_eff planes */
    void    *force4;      /* float4[sorted_cap] sorted order: force records of the lent region, hand-off scratch (the own cells'
                             records live by slot since ABI 6: read them with psamd_download_force4) */
    void    *cell_start;  /* int[num_cells+1]   exclusive prefix of cell counts */
    int64_t  container_size;
    int32_t  num_cells;
    int32_t  live;        /* live particles at the last build_grid             */
    int64_t  sorted_cap;  /* plane stride of snap_soa, in floats                */
    void    *stream;      /* hipStream_t the stages are enqueued on            */
} psamd_device_view;

typedef struct psamd_ctx psamd_ctx;

/* ---- lifetime ------------------------------------------------------------ */
int         psamd_abi_version(void);
const char *psamd_status_string(int status);
int         psamd_default_config(psamd_config *cfg);
/* DoInit + init_particles + q_start_fast + pkg_distrib (ps.cpp:2200-2235,
 * 722-753, 814-871, 893-911): allocates the container, marks every slot free. */
int         psamd_create(const psamd_config *cfg, psamd_ctx **out);
int         psamd_destroy(psamd_ctx *ctx);
const char *psamd_last_error(const psamd_ctx *ctx);
int         psamd_get_sizes(const psamd_ctx *ctx, psamd_sizes *out);
int         psamd_get_config(const psamd_ctx *ctx, psamd_config *out);

/* ---- host-only geometry (no device needed) ------------------------------------------ */
/* What DoInit and the one-off setup stages derive from a configuration, computed on the
 * host without touching a GPU: sizes (ps.cpp:2204-2222), the cell -> (chunk, seg_type,
 * seg_tid) table (get_cell_info), the chunk package table (set_pkg_segments) and the
 * initial free-slot queues (q_start_fast).  Any output pointer may be NULL. */
int psamd_describe(const psamd_config *cfg, psamd_sizes *sizes, int32_t *cell_table3,
                   int32_t *pkgdistrib_pairs, void *queue_info24, int32_t *queue);

/* ---- setup stage: fill_particles, task 5 (ps.cpp:915-1048) --------------- */
/* Places n particles in order; each takes the next free slot of its segment
 * (q_remove) and is initialised as create_particle_s does (app.cu:189-208).
 * w / age / fert_age may be NULL => particle_weight / 0 / 0.  ids_out (may be
 * NULL) receives the slot ids.  On error nothing after the failing particle is
 * placed and *n_done (may be NULL) says how many were. */
int psamd_fill_particles(psamd_ctx *ctx, int64_t n, const float *xyz, const float *vxyz,
                         const float *w, const float *age, const float *fert_age,
                         int32_t *ids_out, int64_t *n_done);
/* The reference's own initial distribution (ps.cpp:974-1028) with a fixed seed in
 * place of std::random_device: n points uniform in the box, written to xyz_out. */
int psamd_uniform_cloud(const psamd_ctx *ctx, int64_t n, uint32_t seed, float *xyz_out);

/* ---- the reference's buffers, in the reference's own layouts -------------- */
/* P_DATA_TYPE[count] (72-byte records, common.h:94-120) for slots first..first+count-1 */
int psamd_upload_particles(psamd_ctx *ctx, const void *p72, int64_t first, int64_t count);
int psamd_download_particles(psamd_ctx *ctx, void *p72, int64_t first, int64_t count);
/* T_DATA_TYPE[count] (24-byte records, common.h:122-132): the build_grid snapshot.  Inside the library the rows are a
 * MIRROR kept for this call: nothing in the step reads them (the pair stage reads its own sorted snapshot, gathered
 * from the particle arrays).  A host that never fetches T_DATA switches the mirror off -- psamd_set_tdata_mirror(ctx, 0):
 * build_grid then leaves the rows alone (56 bytes of traffic per particle and step less) and this call returns
 * PSAMD_ERR_STATE; switching it on again makes the rows exact from the next build_grid on for the slots alive then
 * (rows of slots that were alive only while it was off keep their older contents).  Default: on. */
int psamd_download_tdata(psamd_ctx *ctx, void *t24, int64_t first, int64_t count);
int psamd_set_tdata_mirror(psamd_ctx *ctx, int enabled);
/* QUEUE_INFO[queue_info_size] + int[container_size] (common.h:134-139, ps.cpp:72-73) */
int psamd_upload_queues(psamd_ctx *ctx, const void *queue_info24, const int32_t *queue);
int psamd_download_queues(psamd_ctx *ctx, void *queue_info24, int32_t *queue);
/* int[n_cellgrid] / int[n_chunkgrid], element 0 of each row = count (ps.cpp:1502-1516) */
int psamd_download_cellgrid(psamd_ctx *ctx, int32_t *out);
int psamd_download_chunkgrid(psamd_ctx *ctx, int32_t *out);
/* int[num_cells]: per cell, the particles the last pair pass computed a force for (those the
 * reference's force loop runs for, ps.cpp:1242-1263: no collision this step, not a kid).
 * Valid after psamd_calc_forces_pairs of the same frame. */
int psamd_download_force_counts(psamd_ctx *ctx, int32_t *out);
/* PAIR[num_chunks*27] (app_common.cu:150-232) and the cell -> (chunk, seg_type,
 * seg_tid) table (get_cell_info, app_common.cu:50-148), 3 ints per cell */
int psamd_get_pkgdistrib(const psamd_ctx *ctx, int32_t *pairs_out);
int psamd_get_cell_table(const psamd_ctx *ctx, int32_t *out3_per_cell);
/* hostGridMax: [0] biggest chunk, [1] biggest cell (ps.cpp:76, read at ps.cpp:1900) */
int psamd_get_gridmax(psamd_ctx *ctx, int32_t out2[2]);

/* ---- the three per-step stages ------------------------------------------ */
int psamd_init_iframe(psamd_ctx *ctx);  /* task 3, ps.cpp:1574-1606 / psCUDA.cu:104-150 */
int psamd_build_grid(psamd_ctx *ctx);   /* task 8, ps.cpp:1468-1537 / psCUDA.cu:442-499 */
int psamd_calc_forces(psamd_ctx *ctx);  /* task 6, ps.cpp:1120-1383 / psCUDA.cu:152-423 */
/* calc_forces in its two halves: _pairs computes every particle's collision flag and acceleration
 * (ps.cpp:1182-1263) and leaves the acceleration where the particle keeps it -- T_DATA's ax, ay, az, as
 * the reference's thread does at ps.cpp:1300-1302 -- and the flag beside it (psamd_download_force4 reads
 * both back in the cell-sorted order); _apply does everything after the two neighbour loops (kill /
 * survive / integrate / explosion / relocation, ps.cpp:1210-1374).  psamd_calc_forces == _pairs then
 * _apply; between the two a download of the particles shows the new accelerations beside the old
 * positions and velocities. */
int psamd_calc_forces_pairs(psamd_ctx *ctx);
int psamd_calc_forces_apply(psamd_ctx *ctx);
/* nsteps x {init_iframe, build_grid, calc_forces}, enqueued on the context's stream.  NOTHING in a step waits
 * for the host: the step's one read-back (live count, sticky error bits, list sizes -- what the reference's driver
 * fetches as hostGridMax, ps.cpp:1878-1900) lands in a pinned host record that the library reads ONE STEP LATE.
 * A stage call therefore returns the verdict of the steps BEFORE the one it has just enqueued (run-ahead 1, the
 * default: the host stays a step ahead of the GPU and is never on the step's critical path); psamd_synchronize
 * waits for everything enqueued and returns whatever verdict is outstanding.  psamd_set_run_ahead(ctx, 0): every
 * call that ends a step (psamd_step, psamd_calc_forces[_apply], psamd_slab_finish) waits for that step's own
 * record before it returns, as the reference's driver waits for its task (ps.cpp:1716).  Calls that hand buffers or
 * counters to the caller synchronise by themselves.  If a step's record does not arrive within 10 s (environment
 * PSAMD_WAIT_LIMIT_S) while its stream stays busy, the call returns PSAMD_ERR_STATE and the context refuses all
 * further work: a wedged GPU is reported, not waited for. */
int psamd_step(psamd_ctx *ctx, int32_t nsteps);
int psamd_synchronize(psamd_ctx *ctx);
int psamd_set_run_ahead(psamd_ctx *ctx, int steps);   /* 0 or 1 */

/* float4 (ax, ay, az, flag-as-int-bits) entries [first, first+count) of the sorted-order
 * force array (diagnostics). */
int psamd_download_force4(psamd_ctx *ctx, void *out_float4, int64_t first, int64_t count);

/* ---- checkpoint / resume --------------------------------------------------------- */
/* The reference keeps its whole state in the nine buffers; the
 * device-side image of them (particles + free-slot queues) can be saved once and
 * restored any number of times without leaving HBM. */
int psamd_snapshot_save(psamd_ctx *ctx);
int psamd_snapshot_restore(psamd_ctx *ctx);

/* ---- slab partition (multi-GPU) ---------------------------------------------------- */
/* The reference distributes by segment: a chunk subtask subscribes to its interior segment
 * and the 26 face / edge / corner segments around it (ps.cpp:380-487, set_pkg_segments
 * app_common.cu:150-232), each one contiguous slot range with its own free-slot queue.
 * Here the same segments are dealt to the GPUs of a node in slabs of cell layers along i3
 * (the slowest cell index, so a slab is one contiguous run of the cell-major order):
 *   - STATE layers: the segments whose particles, slots and queues live on the rank.  The
 *     reference numbers segments plane by plane, so a rank owns one slot range and one run of
 *     QUEUE_INFO records per segment type.  Every queue has exactly one owner, which replays
 *     its operations in the reference's serial order.
 *   - COMPUTE layers: the cells whose collision flags and forces the rank evaluates; cut for
 *     balance, anywhere.  Where a cut falls inside a segment group, the upper layers are
 *     computed by the rank above ("lent"): their snapshot travels up with the halo layer and
 *     their (ax, ay, az, flag) records come back before the owner integrates.
 * Per step a rank exchanges, with its two neighbours only: the snapshot (x, y, z, mass, age,
 * id) of its boundary layers; the force records of lent layers; and the particles whose new
 * segment belongs to the neighbour (periodic box: the ring closes), keyed so that the
 * neighbour's queue hands out their slots in the reference's order.  Nothing is replicated
 * but the O(cells) tables.  Message sizes are known to both ends without a word between them (fixed by the plan;
 * the transfer messages grow by a rule every rank evaluates on the same all-gathered numbers); the transport (RCCL
 * send/recv on device memory, or anything else) is the caller's: see particlesystem_amd/slab.py. */
typedef struct psamd_slab_plan {
    int32_t world, rank, grid_dim;
    int32_t cut_lo, cut_hi;          /* compute layers [lo, hi)                                  */
    int32_t state_lo, state_hi;      /* layers whose particles live here                         */
    int32_t below_lo, below_hi;      /* layers received from rank-1: halo layer, then lent layers */
    int32_t above_lo, above_hi;      /* halo layer received from rank+1 (group-aligned cut only)  */
    int32_t lentin_lo, lentin_hi;    /* the part of `below` this rank computes for rank-1         */
    int32_t lentout_lo, lentout_hi;  /* own layers computed by rank+1                             */
    int32_t send_up_lo, send_up_hi;      /* own layers whose snapshot goes to rank+1              */
    int32_t send_down_lo, send_down_hi;  /* own layers whose snapshot goes to rank-1              */
    int32_t slot_lo[4], slot_hi[4];  /* owned slot range per segment type (1, 2, 4, 8)            */
    int32_t rec_lo[4], rec_hi[4];    /* owned QUEUE_INFO records per segment type                 */
    int32_t up_rank, down_rank;      /* ring neighbours for particles that change owner; -1: none */
} psamd_slab_plan;
/* host only, no device needed: the plan of cfg->rank in a world of cfg->world ranks */
int psamd_slab_plan_describe(const psamd_config *cfg, psamd_slab_plan *out);
int psamd_get_slab_plan(const psamd_ctx *ctx, psamd_slab_plan *out);

/* Message buffers (device memory owned by the context; bytes = 0: this rank has no such
 * message).  Index 0 = the neighbour below (rank-1), 1 = the neighbour above (rank+1). */
typedef struct psamd_slab_buffers {
    void   *halo_out[2], *halo_in[2];      /* layer snapshots                                    */
    int64_t halo_out_bytes[2], halo_in_bytes[2];
    void   *force_out, *force_in;          /* force records of lent layers: out to rank-1, in from rank+1 */
    int64_t force_out_bytes, force_in_bytes;
    void   *xfer_out[2], *xfer_in[2];      /* particles changing owner (ring: down_rank / up_rank) */
    int64_t xfer_bytes;                    /* all four the same size: what to post THIS step -- it may change from step to step, on
                                              every rank in the same step (config.xfer_cap_max); read it after psamd_slab_apply */
    void   *status_out, *status_in;        /* ALL-GATHERED once per step: status_in = world records of status_bytes each, by rank */
    int64_t status_bytes;
    void   *allg_out, *allg_in;            /* PSAMD_FLAG_ALL_PAIRS only, ALL-GATHERED once per step between slab_build and slab_pairs:
                                              the snapshot (x, y, z, w_eff) of every rank's own cells; allg_in = world blocks of allg_bytes */
    int64_t allg_bytes;
    void   *xfer2_out[2], *xfer2_in[2];    /* same exchange as xfer_*, but between ranks TWO apart on the ring: out[0] -> rank-2's in[1],
                                              out[1] -> rank+2's in[0].  Only in worlds (>= 4 ranks) where some rank's whole state is one
                                              cell layer, which a particle crossing two layers in a step can fly over; else 0 bytes */
    int64_t xfer2_bytes;
    void   *far_out, *far_in;              /* ALL-GATHERED in the transfer phase (with xfer_*): records for a rank further away than the
                                              neighbour messages reach; far_in = world blocks of far_bytes, by rank.  A particle whose
                                              position stopped being a number is filed under one fixed cell wherever it was (the
                                              reference's conversion): only births make such particles, so the buffers exist in worlds
                                              of >= 4 ranks with PSAMD_FLAG_EXPLOSIONS; else 0 bytes */
    int64_t far_bytes;
    int64_t xfer_bytes_max;                /* the room of the xfer_* buffers: xfer_bytes never grows beyond it (config.xfer_cap_max) */
} psamd_slab_buffers;
int psamd_slab_buffers_get(psamd_ctx *ctx, psamd_slab_buffers *out);

/* One step = build, [all-gather status_out into every rank's status_in -- it must have landed before the
 * FIRST pair-stage call, pairs_interior where that is used: the stage writes the particles' new accelerations
 * into their records and has to know whom the chunk lists' capacity rule takes out of the step;
 * exchange halo_out -> neighbours' halo_in; with PSAMD_FLAG_ALL_PAIRS the all-gather of allg_out into
 * allg_in, which must have landed], pairs, [force_out -> rank-1's force_in], apply, [xfer_out -> neighbours' xfer_in; where they exist xfer2_* likewise and the all-gather of far_out into far_in], finish.  The status record carries a rank's
 * sticky error bits, the slots the cell-overflow rule killed, which the reference frees into queue
 * record 0 wherever they were (ps.cpp:1523-1526), and the rank's part of every chunk's particle count
 * per segment type, from which all ranks reproduce the chunk lists' capacity rule (ps.cpp:1502-1508)
 * and hostGridMax[0].  A slab fails COLLECTIVELY: slab_finish returns an error only for error bits
 * that were in a step's status records, which all ranks see alike -- with run-ahead 1 (the default) from the
 * slab_finish of the step AFTER, on every rank alike; an error raised after a rank's
 * record was closed goes out with the next step's record and stops every rank there (or is reported by
 * psamd_synchronize).  All asynchronous on the context's stream.  With world == 1
 * the four calls are psamd_step(1) cut in four and no message exists. */
int psamd_slab_build(psamd_ctx *ctx);   /* init_iframe + build_grid of the own layers; packs halo_out   */
int psamd_slab_pairs_interior(psamd_ctx *ctx);  /* optional, while the halo travels: the pair stage of the cells whose
                                                    stencil lies in the rank's own layers (needs the status records only) */
int psamd_slab_pairs(psamd_ctx *ctx);   /* unpacks halo_in; collision flags + forces (of the remaining cells); packs force_out */
int psamd_slab_apply(psamd_ctx *ctx);   /* unpacks force_in; integrate ... (calc_forces' tail); closes xfer_out */
int psamd_slab_finish(psamd_ctx *ctx);  /* merges xfer_in; queue replay and relocation                  */
/* Transport through host memory (tests, two processes sharing one GPU): copy message buffer
 * `which` to / from the host.  which: 0/1 halo_out[0/1], 2/3 halo_in[0/1], 4 force_out,
 * 5 force_in, 6/7 xfer_out[0/1], 8/9 xfer_in[0/1], 10 status_out, 11 status_in, 12 allg_out, 13 allg_in,
 * 14/15 xfer2_out[0/1], 16/17 xfer2_in[0/1], 18 far_out, 19 far_in. */
int psamd_slab_msg_download(psamd_ctx *ctx, int which, void *host, int64_t bytes);
int psamd_slab_msg_upload(psamd_ctx *ctx, int which, const void *host, int64_t bytes);

/* Enqueue all further work on the caller's HIP stream (e.g. the one RCCL orders
 * against) instead of the context's own.  NULL restores the context's stream. */
int psamd_set_stream(psamd_ctx *ctx, void *hip_stream);
/* the HIP stream the context enqueues on now (its own unless psamd_set_stream gave it another) */
int psamd_get_stream(psamd_ctx *ctx, void **hip_stream_out);

/* One submission per stage sequence: with graphs on, the kernels a stage call enqueues (psamd_slab_build / _pairs /
 * _apply / _finish up to its read-back; psamd_step: init_iframe .. the queue replay) are captured into a hipGraph the
 * first time a launch shape is met and replayed afterwards -- a rank's step is then four or five submissions instead of
 * two dozen launches.  The results are the same kernels' (tests compare every byte with graphs on); steps that carry
 * timing events run eagerly.  Nothing a graph replays depends on the step: sizes, the step's number and the sequence
 * number of the scalar record live in device memory.  psamd_get_graph_stats: replays and captures so far; returns
 * PSAMD_ERR_UNSUPPORTED (and says why) if the runtime refused a capture and the context fell back to plain launches. */
int psamd_set_graphs(psamd_ctx *ctx, int enabled);
int psamd_get_graph_stats(psamd_ctx *ctx, int64_t *launches, int64_t *captures);
/* How the calling thread waits for a step's scalars when it has to (the one read-back of a step, ps.cpp:1878-1900;
 * with run-ahead the record is there long before it is asked for): 0 spins on the
 * pinned record (default of a single context: lowest latency), 1 spins for a few microseconds and then sleeps in
 * 5-us naps (default of a slab: a node's eight ranks do not pin eight cores).  While it naps the library lowers the
 * calling thread's timer slack (prctl PR_SET_TIMERSLACK) to 1 us and restores the old value before the call returns. */
int psamd_set_wait_policy(psamd_ctx *ctx, int policy);

/* ---- introspection -------------------------------------------------------- */
int psamd_get_counters(psamd_ctx *ctx, psamd_counters *out);
int psamd_live_count(psamd_ctx *ctx, int64_t *out);
int psamd_device_view_get(psamd_ctx *ctx, psamd_device_view *out);
/* Diagnostic builds (-DPSAMD_WAVE_TRACE) record per pair-kernel wave: start, end
 * (100 MHz real-time counter) and hardware id; 3 words per wave slot.  Zeros otherwise. */
int psamd_debug_wave_trace(psamd_ctx *ctx, uint64_t *out, int64_t n_words);

/* Exhaustive check of the hand-written correctly rounded fp32 sqrt / reciprocal used by
 * the pair kernel against the compiler's forms, over every float with bit pattern in
 * [lo_bits, hi_bits].  out24[0..2] = mismatches of the sqrt, the reciprocal and their
 * composition RN(1/RN(sqrt x)) as used; [3] = mismatches of a rejected shortcut (for the
 * record); [8..15], [16..23] = first offending inputs of the sqrt and the composition. */
int psamd_selftest_math(psamd_ctx *ctx, uint32_t lo_bits, uint32_t hi_bits, uint64_t out24[24]);
/* Device time per kernel group, accumulated over the steps since psamd_set_timing, in
 * microseconds, measured with HIP events on the context's stream: hist, scan, scatter,
 * sort, pairs (the force pass), apply, lifecycle, frame reset, collide (collision flags and the
 * lists of the particles that need a force: the two-pass prologue of the pair stage).  level 0: off; 1: collide, pairs, apply and lifecycle
 * only (four events per step); 2: every stage (an event between two kernels costs a few
 * microseconds of idle GPU, so this is for diagnosis).  Never makes a step wait: a step's events are read two timed
 * steps later, or by psamd_get_timing.
 * psamd_set_timing_period(ctx, n): record the events on every n-th step only (n >= 1; default 1) --
 * the accumulated times and `launches` then count those steps; what a long timed run uses so that
 * the events' idle gaps (four to six per step at level 1) do not weigh on the steps in between. */
#define PSAMD_NUM_TIMERS 9
int psamd_set_timing(psamd_ctx *ctx, int level);
int psamd_set_timing_period(psamd_ctx *ctx, int every);
int psamd_get_timing(psamd_ctx *ctx, double us_out[PSAMD_NUM_TIMERS], int64_t *launches);
/* the same intervals as a distribution over the timed steps: median and maximum per timer (a mean hides a stall) */
int psamd_get_timing_stats(psamd_ctx *ctx, double median_us[PSAMD_NUM_TIMERS], double max_us[PSAMD_NUM_TIMERS], int64_t *samples);

#ifdef __cplusplus
}
#endif
#endif /* PSAMD_H */
