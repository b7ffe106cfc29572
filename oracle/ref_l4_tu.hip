/*
 * ref_l4_tu.hip -- translation unit that compiles the REFERENCE's own L4 helper
 * sources, unmodified and from where they lie under /root/reference, into
 * oracle/_ref/libref_l4.so (recipe: oracle/Makefile, target `ref`).
 *
 * TEST INFRASTRUCTURE ONLY: used to pin oracle/ps_oracle.c and to generate the
 * golden vectors under tests/golden/.  Never shipped, never measured.  The GPU box
 * has no /root/reference, so nothing is built from this file there, and no `-m gpu`
 * test, smoke() or bench.py loads the library (tests/test_oracle_vs_ref.py, which
 * does, runs without a GPU and skips itself where the library is absent).
 *
 * What is built: common.h, app_common.cu, app.cu (the per-particle arithmetic,
 * cell/segment index math, free-slot queues).  hipcc's host-only pass supplies
 * __device__/__host__, float3 and atomicCAS natively; the two device-only RNG
 * helpers in app.cu name cuRAND's state type, which the recipe maps onto the
 * hipRAND type that ships in this image (-DcurandState=... on the command
 * line).  No header or library is written to stand in for a missing one.
 *
 * What is NOT built: particleSystem.cpp (stage bodies + driver) and
 * particleSystemCUDA.cu need the pmlib/Unicorn runtime headers, which are not in
 * the reference tree or the image => unbuildable here.
 *
 * The only declarations added here are a prototype the reference keeps in
 * particleSystem.h (not includable: it does `using namespace pm`) and the
 * extern "C" accessors below, which are this repo's own code.
 */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <hip/hip_runtime.h>
#include <hiprand/hiprand_kernel.h>

#include "common.h"
int get_natural_pos(int particle_id, int subtask_id); /* particleSystem.h:13 */
#include "app_common.cu"
#include "app.cu"

extern "C" {

/* sizes/offsets of the reference structs, so the oracle's images can be checked */
void ref_struct_layout(int out[24])
{
    int n = 0;
    out[n++] = (int)sizeof(P_DATA_TYPE);
    out[n++] = (int)offsetof(P_DATA_TYPE, id);
    out[n++] = (int)offsetof(P_DATA_TYPE, cell);
    out[n++] = (int)offsetof(P_DATA_TYPE, chunk);
    out[n++] = (int)offsetof(P_DATA_TYPE, seg_type);
    out[n++] = (int)offsetof(P_DATA_TYPE, seg_tid);
    out[n++] = (int)offsetof(P_DATA_TYPE, seg_fault);
    out[n++] = (int)offsetof(P_DATA_TYPE, is_parent);
    out[n++] = (int)offsetof(P_DATA_TYPE, w);
    out[n++] = (int)offsetof(P_DATA_TYPE, age);
    out[n++] = (int)offsetof(P_DATA_TYPE, fertility_age);
    out[n++] = (int)offsetof(P_DATA_TYPE, x);
    out[n++] = (int)offsetof(P_DATA_TYPE, vx);
    out[n++] = (int)offsetof(P_DATA_TYPE, ax);
    out[n++] = (int)sizeof(T_DATA_TYPE);
    out[n++] = (int)sizeof(QUEUE_INFO);
    out[n++] = (int)sizeof(PAIR);
    while (n < 24) out[n++] = 0;
}

/* the integer macros of common.h:12-50 at their shipped values */
void ref_int_constants(int out[32])
{
    int n = 0;
    out[n++] = MAX_PARTICLES_NUM; out[n++] = X_FACTOR; out[n++] = CHUNK_FACTOR; out[n++] = CHUNK_DIM;
    out[n++] = GRID_DIM; out[n++] = NUM_CELLS; out[n++] = NUM_CHUNKS; out[n++] = NUM_CELLS_PER_CHUNK;
    out[n++] = MAX_PARTICLES_PER_CELL; out[n++] = MAX_PARTICLES_PER_CHUNK; out[n++] = MAX_NEIB_PARTICLES;
    out[n++] = SEG1_CELLS; out[n++] = SEG2_CELLS; out[n++] = SEG4_CELLS; out[n++] = SEG8_CELLS;
    out[n++] = SEG1_COUNT; out[n++] = SEG2_COUNT; out[n++] = SEG4_COUNT; out[n++] = SEG8_COUNT;
    out[n++] = SEG1_SIZE_T; out[n++] = SEG2_SIZE_T; out[n++] = SEG4_SIZE_T; out[n++] = SEG8_SIZE_T;
    out[n++] = SEG1_SIZE; out[n++] = SEG2_SIZE; out[n++] = SEG4_SIZE; out[n++] = SEG8_SIZE;
    out[n++] = CONTAINER_SIZE; out[n++] = QUEUE_INFO_SIZE;
    while (n < 32) out[n++] = 0;
}

/* the floating macros of common.h:52-69 */
void ref_real_constants(double out[16])
{
    int n = 0;
    out[n++] = CELL_SIZE; out[n++] = EPS2; out[n++] = COLLISION_RADIUS; out[n++] = PARTICLE_WEIGHT_DEFAULT;
    out[n++] = DT; out[n++] = PARTICLE_LIFE; out[n++] = KID_AGE;
    out[n++] = MIN_FERTILITY_AGE; out[n++] = MAX_FERTILITY_AGE; out[n++] = MIN_ADULT_AGE; out[n++] = MAX_ADULT_AGE;
    out[n++] = MAX_DX; out[n++] = MAX_V; out[n++] = EXPLOSION_SPEED;
    while (n < 16) out[n++] = 0.0;
}

void ref_get_cell_info(int cell, int out3[3])
{
    INT3 r = get_cell_info(cell);
    out3[0] = r.a; out3[1] = r.b; out3[2] = r.c;
}
int ref_get_cont_rloc(int seg_type, int seg_tid) { return get_cont_rloc(seg_type, seg_tid); }
int ref_get_info_rloc(int seg_type, int seg_tid) { return get_info_rloc(seg_type, seg_tid); }
void ref_get_id_info(int id, int out2[2])
{
    PAIR p = get_id_info(id);
    out2[0] = p.c; out2[1] = p.p;
}
void ref_set_pkg_segments(int chunk, int out54[54])
{
    PAIR L[27];
    set_pkg_segments(chunk, L);
    for (int i = 0; i < 27; i++) { out54[2 * i] = L[i].c; out54[2 * i + 1] = L[i].p; }
}
int ref_fill_cells(int cell, int out27[27])
{
    NEIB_CELLS nc;
    nc.size = 0;
    nc.data[nc.size++] = cell;
    fill_cells(nc);
    for (int i = 0; i < nc.size; i++) out27[i] = nc.data[i];
    return nc.size;
}
/* gather the neighbour id list of `cell` from a reference-layout cell grid */
int ref_fill_particles(int cell, int *cellGrid, int *out, int cap)
{
    NEIB_CELLS nc;
    nc.size = 0;
    nc.data[nc.size++] = cell;
    fill_cells(nc);
    static NEIB_PARTICLES np; /* 55 KB */
    np.size = 0;
    fill_particles(np, nc, cellGrid);
    int n = np.size < cap ? np.size : cap;
    memcpy(out, np.data, sizeof(int) * (size_t)n);
    return np.size;
}

/* P_DATA_TYPE images are passed as raw 72-byte records (caller zeroes padding) */
void ref_set_pos_x(void *p72, float x, float y, float z)
{
    FLOAT3 r = {x, y, z};
    set_pos_x(*(P_DATA_TYPE *)p72, r);
}
void ref_set_pos_i(void *p72, float x, float y, float z)
{
    FLOAT3 r = {x, y, z};
    set_pos_i(*(P_DATA_TYPE *)p72, r);
}
void ref_create_particle_s(void *arr72, int t, float w, float age, float fert_age,
                           float x, float y, float z, float vx, float vy, float vz)
{
    create_particle_s((P_DATA_TYPE *)arr72, t, w, age, fert_age, x, y, z, vx, vy, vz);
}
void ref_reset_particle(void *p72)   { reset_particle(*(P_DATA_TYPE *)p72); }
void ref_survive_particle(void *p72) { survive_particle(*(P_DATA_TYPE *)p72); }
void ref_copy_particle(void *dst72, const void *src72)
{
    copy_particle(*(P_DATA_TYPE *)dst72, *(const P_DATA_TYPE *)src72);
}

/* batched pair kernels: n independent (bi, bj, ai) triples */
void ref_body_body_interaction(int n, const void *bi72, const void *bj24, float *ai3)
{
    const P_DATA_TYPE *bi = (const P_DATA_TYPE *)bi72;
    const T_DATA_TYPE *bj = (const T_DATA_TYPE *)bj24;
    for (int k = 0; k < n; k++) {
        FLOAT3 a = {ai3[3 * k], ai3[3 * k + 1], ai3[3 * k + 2]};
        a = bodyBodyInteraction(bi[k], bj[k], a);
        ai3[3 * k] = a.x; ai3[3 * k + 1] = a.y; ai3[3 * k + 2] = a.z;
    }
}
/* serial accumulation of m snapshot bodies onto ONE particle, in array order */
void ref_accumulate(const void *bi72, int m, const void *bj24, float *ai3)
{
    const P_DATA_TYPE *bi = (const P_DATA_TYPE *)bi72;
    const T_DATA_TYPE *bj = (const T_DATA_TYPE *)bj24;
    FLOAT3 a = {ai3[0], ai3[1], ai3[2]};
    for (int k = 0; k < m; k++)
        if (bi->id != bj[k].id) a = bodyBodyInteraction(*bi, bj[k], a);
    ai3[0] = a.x; ai3[1] = a.y; ai3[2] = a.z;
}
void ref_body_body_collision(int n, const void *bi72, const void *bj24, int *flags)
{
    const P_DATA_TYPE *bi = (const P_DATA_TYPE *)bi72;
    const T_DATA_TYPE *bj = (const T_DATA_TYPE *)bj24;
    for (int k = 0; k < n; k++) flags[k] = bodyBodyCollision(bi[k], bj[k]);
}

int  ref_q_remove(void *qinfo, int *queue, int seg_type, int seg_tid)
{
    return q_remove((QUEUE_INFO *)qinfo, queue, seg_type, seg_tid);
}
void ref_q_insert(void *qinfo, int *queue, int seg_type, int seg_tid, int x)
{
    q_insert((QUEUE_INFO *)qinfo, queue, seg_type, seg_tid, x);
}

} /* extern "C" */
