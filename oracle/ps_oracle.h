/*
 * ps_oracle.h -- CPU ORACLE for the abraj/particleSystem per-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference's
 * `_host` code path (reset frame -> build grid -> calc forces) used as the
 * checker in tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 * Nothing under particlesystem_amd/ (the product) may include, link or call it.
 *
 * Parity status: PINNED.  The per-particle arithmetic and index math restated
 * here is checked bit-for-bit against the reference's own L4 helper sources
 * (app_common.cu, app.cu, common.h), compiled unmodified from /root/reference
 * into oracle/_ref/libref_l4.so by oracle/Makefile, and against the committed
 * golden vectors in tests/golden/ that were generated from that library
 * (tests/golden/make_golden.py).  The stage bodies themselves live in
 * particleSystem.cpp, which cannot be built here (it needs the absent pmlib
 * runtime, commonAPI.h); they are restated from the source text and pinned by
 * the known answers SURVEY.md section 8(c) records from the reference.
 *
 * All "ps.cpp" citations are /root/reference/source/code/src/particleSystem.cpp,
 * the others are under /root/reference/source/code/inc/.
 */
#ifndef PS_ORACLE_H
#define PS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Runtime form of the reference's compile-time configuration (common.h:12-70). */
typedef struct pso_config {
    int    max_particles_num;  /* MAX_PARTICLES_NUM  common.h:12 (1024*1024) */
    int    x_factor;           /* X_FACTOR           common.h:13 (2)         */
    int    chunk_factor;       /* CHUNK_FACTOR       common.h:29 (4)         */
    int    chunk_dim;          /* CHUNK_DIM          common.h:30 (4)         */
    double cell_size;          /* CELL_SIZE          common.h:52 (5.0)       */
    double eps2;               /* EPS2               common.h:53 (0.2)       */
    double collision_radius;   /* COLLISION_RADIUS   common.h:54 (0.4)       */
    double particle_weight;    /* PARTICLE_WEIGHT_DEFAULT common.h:55 (60.0) */
    double dt;                 /* DT                 common.h:69 (0.05)      */
    double max_v;              /* MAX_V              common.h:66 (10.0)      */
    double explosion_speed;    /* EXPLOSION_SPEED    common.h:67 (3.0)       */
    double life_steps;         /* the 300 of PARTICLE_LIFE = (300*DT), common.h:58 */
} pso_config;

/* Everything common.h derives from the above (common.h:20-50, 58-65). */
typedef struct pso_derived {
    int grid_dim, num_cells, num_chunks, cells_per_chunk;
    int max_per_cell, max_per_chunk, max_neib_particles;
    int seg_cells[4], seg_count[4], seg_size_t[4], seg_size[4]; /* index 0..3 = type 1,2,4,8 */
    int container_size, queue_info_size;
    double particle_life, kid_age, min_fertility_age, max_fertility_age;
    double min_adult_age, max_adult_age, max_dx;
} pso_derived;

/* Field-for-field image of P_DATA_TYPE (common.h:94-120): 72 bytes. */
typedef struct pso_particle {
    int id, cell, chunk, seg_type, seg_tid;
    unsigned char seg_fault, is_parent; /* C++ bool, 1 byte each; 2 pad bytes follow */
    float w, age, fertility_age;
    float x, y, z;
    float vx, vy, vz;
    float ax, ay, az;
} pso_particle;

/* T_DATA_TYPE (common.h:122-132): 24 bytes. */
typedef struct pso_tdata { int id; float x, y, z, w, age; } pso_tdata;

/* QUEUE_INFO (common.h:134-139): 24 bytes. */
typedef struct pso_queue_info { int front, rear, count, lock, rloc, seg_size; } pso_queue_info;

typedef struct pso_pair { int c, p; } pso_pair; /* PAIR common.h:141-145 */

/* Explosion randomness (ps.cpp:29-56 uses std::random_device, so the reference
 * is non-deterministic there).  The oracle asks the caller for the three
 * integers in [-50,49] and the uniform u in [0,1) instead. */
typedef void (*pso_rng_fn)(void *user, int parent_id, int step, int ints_out[3], double *u_out);

typedef struct pso_counters {
    long long deaths_age, deaths_collision, survives, integrated;
    long long relocations, relocations_lost, births, births_failed, cell_overflow_kills;
    long long explosions_skipped; /* explosion due but no rng callback installed */
} pso_counters;

typedef struct pso_system pso_system;

void pso_default_config(pso_config *cfg);
int  pso_derive(const pso_config *cfg, pso_derived *d);

/* create = DoInit allocation + init_particles + q_start_fast + pkg_distrib
 * (ps.cpp:2200-2235, 722-753, 814-871, 893-911). Returns NULL on bad config. */
pso_system *pso_create(const pso_config *cfg);
void        pso_destroy(pso_system *s);

const pso_config  *pso_get_config(const pso_system *s);
const pso_derived *pso_get_derived(const pso_system *s);

/* fill_particle (ps.cpp:915-960) with explicit age / fertility age instead of
 * random_device draws.  Returns the slot id, -1 queue empty, -2 outside box. */
int pso_fill_particle(pso_system *s, float x, float y, float z,
                      float w, float age, float fert_age);

/* n calls of pso_fill_particle in array order; returns how many were placed */
int pso_fill_particles(pso_system *s, int n, const float *xyz, const float *w,
                       const float *age, const float *fert_age, int *ids_out);

/* The three per-step stages, ps.cpp:1574-1606, 1468-1537, 1120-1383 (driven as
 * DoParallelProcess does, ps.cpp:1843-1928: chunks 0..NUM_CHUNKS-1 in order). */
void pso_init_iframe(pso_system *s);
void pso_build_grid(pso_system *s);
void pso_calc_forces(pso_system *s);
/* calc_forces for one chunk only (one pmlib subtask); elems = GridMax[0]. */
void pso_calc_forces_chunk(pso_system *s, int chunk, int subtask_elems);
void pso_step(pso_system *s, int nsteps);

/* calc_forces cut where ranks exchange results (test support for the sharded path):
 * pso_calc_pairs evaluates, WITHOUT touching any state, the collision flag and the
 * acceleration of the particles with sorted index in [lo, hi) -- sorted = cell-major,
 * slot-ascending inside a cell, i.e. the concatenated cell lists -- into
 * force4[4*k] = ax, ay, az, flag (as float bits of the int).  pso_apply_forces then
 * performs everything calc_forces does after its two neighbour loops, in the
 * reference's serial order, taking flag and acceleration from force4.
 * pairs(0, n) followed by apply is exactly pso_calc_forces. */
int  pso_sorted_count(const pso_system *s);
void pso_calc_pairs(pso_system *s, int lo, int hi, float *force4);
/* same results, the range split over `nthreads` host threads; 0 on success */
int pso_calc_pairs_threads(pso_system *s, int lo, int hi, float *force4, int nthreads);
void pso_apply_forces(pso_system *s, const float *force4);

/* calc_forces' tail with the queue operations deferred (test support for the slab-partitioned
 * path): pso_apply_collect = pso_apply_forces minus every q_insert / q_remove and every
 * placement, which come out as operations keyed by their place in the reference's serial
 * order; pso_replay_ops executes operations (own and received) queue by queue in key order.
 * collect + replay of the same list == pso_apply_forces. */
typedef struct pso_op {
    unsigned long long key;   /* (chunk + 1) << 34 | source slot << 2 | sub-step                  */
    int rec;                  /* QUEUE_INFO record the operation acts on                          */
    int kind;                 /* 0 insert `slot`; 1 remove for the relocation of `body`; 2 remove for the birth of `body` */
    int slot;                 /* insert: the slot freed; remove: the source slot (parent)          */
    int dst;                  /* filled in by the replay: the slot handed out, or -1               */
    int old_cell;             /* cell of the source particle before the step                       */
    int pad;
    pso_particle body;
} pso_op;
int  pso_apply_collect(pso_system *s, const float *force4, pso_op *ops, int cap); /* ops written, < 0: cap too small */
void pso_replay_ops(pso_system *s, pso_op *ops, int n);
void pso_advance_step(pso_system *s);   /* what pso_step does after calc_forces */

void pso_set_rng(pso_system *s, pso_rng_fn fn, void *user);
void pso_set_explosions(pso_system *s, int enabled);

/* borrowed views of the nine reference buffers (ps.cpp:70-78) */
pso_particle   *pso_particles(pso_system *s);
pso_tdata      *pso_tdata_buf(pso_system *s);
int            *pso_queue(pso_system *s);
pso_queue_info *pso_queue_info_buf(pso_system *s);
int            *pso_chunkgrid(pso_system *s);
int            *pso_cellgrid(pso_system *s);
int            *pso_gridmax(pso_system *s);
pso_pair       *pso_pkgdistrib(pso_system *s);
const pso_counters *pso_get_counters(const pso_system *s);
int             pso_step_index(const pso_system *s);
int             pso_live_count(const pso_system *s);

/* ---- L4 helpers, exposed so tests can pin them one by one against _ref ---- */
void pso_get_cell_info(const pso_derived *d, const pso_config *c, int cell, int out3[3]);
int  pso_get_cont_rloc(const pso_derived *d, int seg_type, int seg_tid);
int  pso_get_info_rloc(const pso_derived *d, int seg_type, int seg_tid);
void pso_get_id_info(const pso_derived *d, int id, int out2[2]);
void pso_set_pkg_segments(const pso_config *c, int chunk, pso_pair *seg_list27);
int  pso_fill_cells(const pso_derived *d, int cell, int out27[27]);
int  pso_set_pos_t(const pso_config *c, const pso_derived *d, pso_particle *p, float rx, float ry, float rz);
void pso_set_pos_i(const pso_config *c, const pso_derived *d, pso_particle *p, float rx, float ry, float rz);
void pso_set_pos_x(const pso_config *c, const pso_derived *d, pso_particle *p, float rx, float ry, float rz);
void pso_body_body_interaction(const pso_config *c, const pso_derived *d,
                               const pso_particle *bi, const pso_tdata *bj, float ai[3]);
int  pso_body_body_collision(const pso_config *c, const pso_derived *d,
                             const pso_particle *bi, const pso_tdata *bj);
void pso_integrate(const pso_config *c, const pso_derived *d, pso_particle *p);
void pso_reset_particle(pso_particle *p);
void pso_survive_particle(pso_particle *p);
int  pso_q_remove(pso_queue_info *qi, int *queue, const pso_derived *d, int seg_type, int seg_tid);
void pso_q_insert(pso_queue_info *qi, int *queue, const pso_derived *d, int seg_type, int seg_tid, int x);

#ifdef __cplusplus
}
#endif
#endif
