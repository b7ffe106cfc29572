/*
 * ps_oracle.c -- CPU ORACLE (test infrastructure, never shipped, never linked
 * into the product).  See ps_oracle.h for the parity status and the rules.
 *
 * Plain C99 restatement of the reference's `_host` stage path with the
 * reference's precision choices kept site by site (which operations are
 * evaluated in double and where the result is rounded back to float).
 * Build with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 *
 * "ps.cpp" = /root/reference/source/code/src/particleSystem.cpp; the other
 * files cited are in /root/reference/source/code/inc/.
 */
#include "ps_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef char pso_static_assert_particle[(sizeof(pso_particle) == 72) ? 1 : -1];
typedef char pso_static_assert_tdata[(sizeof(pso_tdata) == 24) ? 1 : -1];
typedef char pso_static_assert_qinfo[(sizeof(pso_queue_info) == 24) ? 1 : -1];

struct pso_system {
    pso_config   cfg;
    pso_derived  d;
    /* the nine buffers of ps.cpp:70-78 (rand states are not needed on host) */
    pso_particle   *particles;
    pso_tdata      *tdata;
    int            *queue;
    pso_queue_info *queue_info;
    int            *chunkgrid;
    int            *cellgrid;
    int             gridmax[2];
    pso_pair       *pkgdistrib;
    /* scratch for the neighbour gather (NEIB_PARTICLES, common.h:183-187) */
    int            *neib;
    pso_counters    ctr;
    pso_rng_fn      rng;
    void           *rng_user;
    int             explosions;
    int             step;
};

/* ------------------------------------------------------------------ config */

void pso_default_config(pso_config *c)
{
    c->max_particles_num = 1024 * 1024; /* common.h:12 */
    c->x_factor          = 2;           /* common.h:13 */
    c->chunk_factor      = 4;           /* common.h:29 */
    c->chunk_dim         = 4;           /* common.h:30 */
    c->cell_size         = 5.0;         /* common.h:52 */
    c->eps2              = 0.2;         /* common.h:53 */
    c->collision_radius  = 0.4;         /* common.h:54 */
    c->particle_weight   = 60.0;        /* common.h:55 */
    c->dt                = 0.05;        /* common.h:69 */
    c->max_v             = 10.0;        /* common.h:66 */
    c->explosion_speed   = 3.0;         /* common.h:67 */
    c->life_steps        = 300.0;       /* common.h:58 */
}

/* common.h:20-50 and 58-65, evaluated at run time */
int pso_derive(const pso_config *c, pso_derived *d)
{
    int F = c->chunk_factor, D = c->chunk_dim, k;
    if (F < 1 || D < 3 || c->max_particles_num < 1 || c->x_factor < 1) return -1;
    memset(d, 0, sizeof *d);
    d->grid_dim        = F * D;
    d->num_cells       = d->grid_dim * d->grid_dim * d->grid_dim;
    d->num_chunks      = F * F * F;
    d->cells_per_chunk = D * D * D;
    d->max_per_cell    = (c->max_particles_num / d->num_cells + 1) * c->x_factor;
    d->max_per_chunk   = d->max_per_cell * d->cells_per_chunk;
    d->max_neib_particles = d->max_per_cell * 27;

    d->seg_cells[0] = (D - 2) * (D - 2) * (D - 2);
    d->seg_cells[1] = 2 * (D - 2) * (D - 2);
    d->seg_cells[2] = 4 * (D - 2);
    d->seg_cells[3] = 8;
    d->seg_count[0] = F * F * F;
    d->seg_count[1] = 3 * F * F * (F + 1);
    d->seg_count[2] = 3 * F * (F + 1) * (F + 1);
    d->seg_count[3] = (F + 1) * (F + 1) * (F + 1);
    d->container_size = 0;
    d->queue_info_size = 0;
    for (k = 0; k < 4; k++) {
        d->seg_size_t[k] = d->seg_cells[k] * d->max_per_cell;
        d->seg_size[k]   = d->seg_count[k] * d->seg_size_t[k];
        d->container_size  += d->seg_size[k];
        d->queue_info_size += d->seg_count[k];
    }
    d->particle_life     = c->life_steps * c->dt;     /* (300*DT)          */
    d->kid_age           = d->particle_life / 10.0;   /* common.h:59       */
    d->min_fertility_age = d->particle_life / 6.0;    /* common.h:60       */
    d->max_fertility_age = d->particle_life * 2.0;    /* common.h:61       */
    d->min_adult_age     = d->particle_life / 7.0;    /* common.h:62       */
    d->max_adult_age     = d->particle_life / 2.0;    /* common.h:63       */
    d->max_dx            = c->cell_size;              /* common.h:65       */
    return 0;
}

/* ----------------------------------------------------- segment index math */

static int seg_slot(int seg_type) /* 1,2,4,8 -> 0..3, anything else -> -1 */
{
    switch (seg_type) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; }
    return -1;
}

/* app_common.cu:6-26: start slot of segment (type, tid) in the container.
 * An unknown type yields 0, exactly as the reference's fall-through switches. */
int pso_get_cont_rloc(const pso_derived *d, int seg_type, int seg_tid)
{
    int k = seg_slot(seg_type), pos = 0, j;
    if (k < 0) return 0;
    for (j = 0; j < k; j++) pos += d->seg_size[j];
    return pos + seg_tid * d->seg_size_t[k];
}

/* app_common.cu:28-48: index of the segment's QUEUE_INFO record */
int pso_get_info_rloc(const pso_derived *d, int seg_type, int seg_tid)
{
    int k = seg_slot(seg_type), pos = 0, j;
    if (k < 0) return 0;
    for (j = 0; j < k; j++) pos += d->seg_count[j];
    return pos + seg_tid;
}

/* app.cu:24-65 and the identical ladders at ps.cpp:1213-1228, 1338-1353 */
void pso_get_id_info(const pso_derived *d, int id, int out2[2])
{
    int k, base = 0;
    out2[0] = -1; out2[1] = -1;
    if (id < 0 || id >= d->container_size) return;
    for (k = 0; k < 4; k++) {
        if (id < base + d->seg_size[k]) {
            out2[0] = 1 << k;
            out2[1] = (id - base) / d->seg_size_t[k];
            return;
        }
        base += d->seg_size[k];
    }
}

/* app_common.cu:50-148: cell -> (chunk, seg_type, seg_tid) */
void pso_get_cell_info(const pso_derived *d, const pso_config *c, int cell, int out3[3])
{
    int G = d->grid_dim, F = c->chunk_factor, D = c->chunk_dim;
    int i3 = cell / (G * G), rem = cell % (G * G);
    int i1 = rem / G, i2 = rem % G;
    int idx[3], q[3], kind[3], t[3], a;
    int FF = F * F, B = 2 * F * (F + 1), E = (F + 1) * (F + 1);
    int seg_type, seg_tid = -1;

    idx[0] = i1; idx[1] = i2; idx[2] = i3;
    for (a = 0; a < 3; a++) {
        int r = idx[a] % D;
        q[a] = (int)floor((idx[a] * 1.0) / D);
        if (r == 0)          { kind[a] = 2; t[a] = q[a]; }
        else if (r == D - 1) { kind[a] = 2; t[a] = q[a] + 1; }
        else                 { kind[a] = 1; t[a] = q[a]; }
    }
    seg_type = kind[0] * kind[1] * kind[2];
    if (seg_type == 1) {
        seg_tid = t[2] * FF + t[0] * F + t[1];
    } else if (seg_type == 2) {
        if (kind[2] == 2)      seg_tid = t[2] * (FF + B) + t[0] * F + t[1];
        else if (kind[1] == 2) seg_tid = (t[2] + 1) * FF + t[2] * B + (t[0] + 1) * F + t[0] * (F + 1) + t[1];
        else if (kind[0] == 2) seg_tid = (t[2] + 1) * FF + t[2] * B + t[0] * (F + F + 1) + t[1];
    } else if (seg_type == 4) {
        if (kind[2] == 1)      seg_tid = (t[2] + 1) * B + t[2] * E + t[0] * (F + 1) + t[1];
        else if (kind[1] == 1) seg_tid = t[2] * B + t[2] * E + t[0] * (2 * F + 1) + t[1];
        else if (kind[0] == 1) seg_tid = t[2] * B + t[2] * E + t[0] * (2 * F + 1) + F + t[1];
    } else if (seg_type == 8) {
        seg_tid = t[2] * E + t[0] * (F + 1) + t[1];
    }
    out3[0] = q[2] * FF + q[0] * F + q[1];
    out3[1] = seg_type;
    out3[2] = seg_tid;
}

/* app_common.cu:150-232: the 27 (type, tid) segments covering a chunk + halo */
void pso_set_pkg_segments(const pso_config *c, int chunk, pso_pair *L)
{
    int F = c->chunk_factor, k;
    int i3 = chunk / (F * F), rem = chunk % (F * F), i1 = rem / F, i2 = rem % F;
    int r = (F + 1) * (F + 1), s = (F + 1) * F, t = F * F;
    int t8[8], t4[12], t2[6];

    t8[0] = i3 * r + i1 * (F + 1) + i2;
    t8[1] = t8[0] + 1;
    t8[2] = t8[0] + F + 1;
    t8[3] = t8[2] + 1;
    for (k = 0; k < 4; k++) t8[4 + k] = t8[k] + r;

    t4[0] = i3 * (2 * s + r) + i1 * (2 * F + 1) + i2;
    t4[1] = t4[0] + F;
    t4[2] = t4[1] + 1;
    t4[3] = t4[2] + F;
    t4[4] = i3 * (2 * s + r) + 2 * s + i1 * (F + 1) + i2;
    t4[5] = t4[4] + 1;
    t4[6] = t4[5] + F;
    t4[7] = t4[6] + 1;
    for (k = 0; k < 4; k++) t4[8 + k] = t4[k] + (2 * s + r);

    t2[0] = i3 * (t + 2 * s) + i1 * F + i2;
    t2[1] = i3 * (t + 2 * s) + t + i1 * (2 * F + 1) + i2;
    t2[2] = t2[1] + F;
    t2[3] = t2[2] + 1;
    t2[4] = t2[3] + F;
    t2[5] = t2[0] + (t + 2 * s);

    L[0].c = 1; L[0].p = chunk;
    for (k = 0; k < 6; k++)  { L[1 + k].c = 2;  L[1 + k].p = t2[k]; }
    for (k = 0; k < 12; k++) { L[7 + k].c = 4;  L[7 + k].p = t4[k]; }
    for (k = 0; k < 8; k++)  { L[19 + k].c = 8; L[19 + k].p = t8[k]; }
}

/* -------------------------------------------------------- neighbour cells */

/* app.cu:370-409: candidate order is the cell itself, then these 26 deltas,
 * written as (d_i2, d_i1, d_i3) with cell delta = d_i2 + d_i1*G + d_i3*G*G. */
static const signed char k_stencil[26][3] = {
    {-1, 0, 0}, {+1, 0, 0},
    {-1, -1, 0}, {0, -1, 0}, {+1, -1, 0},
    {-1, +1, 0}, {0, +1, 0}, {+1, +1, 0},
    {-1, -1, -1}, {0, -1, -1}, {+1, -1, -1},
    {-1, 0, -1}, {0, 0, -1}, {+1, 0, -1},
    {-1, +1, -1}, {0, +1, -1}, {+1, +1, -1},
    {-1, -1, +1}, {0, -1, +1}, {+1, -1, +1},
    {-1, 0, +1}, {0, 0, +1}, {+1, 0, +1},
    {-1, +1, +1}, {0, +1, +1}, {+1, +1, +1},
};

static void cell_index(int G, int cell, int p[3]) /* app.cu:335-350 -> (i1,i2,i3) */
{
    int n = cell;
    p[2] = n / (G * G); n -= p[2] * G * G;
    p[0] = n / G;       n -= p[0] * G;
    p[1] = n;
}

/* app.cu:352-409.  A candidate is kept when its linear index is inside the grid
 * and its (i1,i2,i3) lies within squared index distance 3 of the centre cell:
 * this is what makes the stencil non-periodic. */
int pso_fill_cells(const pso_derived *d, int cell, int out27[27])
{
    int G = d->grid_dim, n = 0, k, p0[3];
    out27[n++] = cell;
    cell_index(G, cell, p0);
    for (k = 0; k < 26; k++) {
        int cand = cell + k_stencil[k][0] + k_stencil[k][1] * G + k_stencil[k][2] * G * G;
        if (cand >= 0 && cand < G * G * G) {
            int p[3], r0, r1, r2;
            cell_index(G, cand, p);
            r0 = p0[0] - p[0]; r1 = p0[1] - p[1]; r2 = p0[2] - p[2];
            if (r0 * r0 + r1 * r1 + r2 * r2 <= 3) out27[n++] = cand;
        }
    }
    return n;
}

/* ------------------------------------------------------ particle helpers */

void pso_reset_particle(pso_particle *p) /* app.cu:239-264 (id is kept) */
{
    p->cell = -1; p->chunk = -1; p->seg_type = -1; p->seg_tid = -1;
    p->seg_fault = 0; p->is_parent = 0;
    p->w = 0.0f; p->age = 0.0f; p->fertility_age = 0.0f;
    p->x = p->y = p->z = 0.0f;
    p->vx = p->vy = p->vz = 0.0f;
    p->ax = p->ay = p->az = 0.0f;
}

void pso_survive_particle(pso_particle *p) /* app.cu:271-283 */
{
    p->age = 0.0f; p->is_parent = 0;
    p->vx = p->vy = p->vz = 0.0f;
    p->ax = p->ay = p->az = 0.0f;
}

/* app.cu:117-158.  Cell indices come from a double floor; positions outside the
 * box are wrapped one grid length at a time, the shift being added in double
 * and rounded back to float. Axis mapping: i1 <- -y, i2 <- +x, i3 <- -z. */
int pso_set_pos_t(const pso_config *c, const pso_derived *d, pso_particle *p,
                  float rx, float ry, float rz)
{
    int G = d->grid_dim, guard = 0;
    double cs = c->cell_size;
    float tx = rx, ty = ry, tz = rz;
    int i1 = (int)(floor((-1.0 * ty) / cs) + (G / 2));
    int i2 = (int)(floor(( 1.0 * tx) / cs) + (G / 2));
    int i3 = (int)(floor((-1.0 * tz) / cs) + (G / 2));

    while (!((i1 >= 0 && i1 < G) && (i2 >= 0 && i2 < G) && (i3 >= 0 && i3 < G))) {
        if (!(i1 >= 0 && i1 < G)) { int o = i1; i1 = (i1 + G) % G; ty = (float)(ty + (-1.0 * (i1 - o) * cs)); }
        if (!(i2 >= 0 && i2 < G)) { int o = i2; i2 = (i2 + G) % G; tx = (float)(tx + ((i2 - o) * cs)); }
        if (!(i3 >= 0 && i3 < G)) { int o = i3; i3 = (i3 + G) % G; tz = (float)(tz + (-1.0 * (i3 - o) * cs)); }
        if (++guard > (1 << 20)) break; /* the reference would spin; unreachable for finite input */
    }
    p->x = tx; p->y = ty; p->z = tz;
    p->cell = i3 * G * G + i1 * G + i2;
    return p->cell;
}

void pso_set_pos_i(const pso_config *c, const pso_derived *d, pso_particle *p,
                   float rx, float ry, float rz) /* app.cu:160-169 */
{
    int info[3];
    int cell = pso_set_pos_t(c, d, p, rx, ry, rz);
    pso_get_cell_info(d, c, cell, info);
    p->chunk = info[0]; p->seg_type = info[1]; p->seg_tid = info[2];
}

void pso_set_pos_x(const pso_config *c, const pso_derived *d, pso_particle *p,
                   float rx, float ry, float rz) /* app.cu:171-187 */
{
    int info[3];
    int cell = pso_set_pos_t(c, d, p, rx, ry, rz);
    pso_get_cell_info(d, c, cell, info);
    p->chunk = info[0];
    if (!(p->seg_type == info[1] && p->seg_tid == info[2])) {
        if (!(p->seg_type == -1 && p->seg_tid == -1)) p->seg_fault = 1;
        p->seg_type = info[1];
        p->seg_tid  = info[2];
    }
}

/* app.cu:189-208 */
static void create_particle(const pso_config *c, const pso_derived *d, pso_particle *p,
                            float w, float age, float fert_age, float x, float y, float z,
                            float vx, float vy, float vz)
{
    pso_set_pos_i(c, d, p, x, y, z);
    p->w = w; p->age = age; p->fertility_age = fert_age; p->is_parent = 0;
    p->vx = vx; p->vy = vy; p->vz = vz;
    p->ax = 0.0f; p->ay = 0.0f; p->az = 0.0f;
}

/* ---------------------------------------------------------------- queues */

/* app_common.cu:305-339 (host overload; natural position == global index) */
int pso_q_remove(pso_queue_info *qi, int *queue, const pso_derived *d, int seg_type, int seg_tid)
{
    pso_queue_info *q = &qi[pso_get_info_rloc(d, seg_type, seg_tid)];
    int pos, item;
    if (q->count <= 0) return -1;
    pos = q->front;
    if (q->count == 1) { q->front = -1; q->rear = -1; }
    else if (q->front == q->rloc + q->seg_size - 1) q->front = q->rloc;
    else q->front++;
    q->count--;
    item = queue[pos];
    queue[pos] = -1;
    return item;
}

/* app_common.cu:346-376 */
void pso_q_insert(pso_queue_info *qi, int *queue, const pso_derived *d, int seg_type, int seg_tid, int x)
{
    pso_queue_info *q = &qi[pso_get_info_rloc(d, seg_type, seg_tid)];
    if (q->count == q->seg_size) return;
    if (q->count == 0) { q->front = q->rloc; q->rear = q->rloc; }
    else if (q->rear == q->rloc + q->seg_size - 1) q->rear = q->rloc;
    else q->rear++;
    q->count++;
    queue[q->rear] = x;
}

/* --------------------------------------------------------- pair kernels */

/* app_common.cu:236-267.  Softened gravity of snapshot body bj on bi.
 * distSq is fp32; EPS2 is a double literal, so the add is done in double and
 * rounded to float; the cube, sqrtf and the reciprocal are fp32. */
void pso_body_body_interaction(const pso_config *c, const pso_derived *d,
                               const pso_particle *bi, const pso_tdata *bj, float ai[3])
{
    float rx, ry, rz, distSq, distSqr, distSixth, invDistCube, s;
    if ((double)bi->age < d->kid_age || (double)bj->age < d->kid_age) return;
    rx = bj->x - bi->x; ry = bj->y - bi->y; rz = bj->z - bi->z;
    distSq  = rx * rx + ry * ry + rz * rz;
    distSqr = (float)((double)distSq + c->eps2);
    distSixth = distSqr * distSqr * distSqr;
    invDistCube = 1.0f / sqrtf(distSixth);
    s = bj->w * invDistCube;
    ai[0] += rx * s; ai[1] += ry * s; ai[2] += rz * s;
}

/* app_common.cu:269-301.  0 none, 1 survive (bi has the higher id), 2 kill. */
int pso_body_body_collision(const pso_config *c, const pso_derived *d,
                            const pso_particle *bi, const pso_tdata *bj)
{
    float rx = bj->x - bi->x, ry = bj->y - bi->y, rz = bj->z - bi->z;
    float dist = sqrtf(rx * rx + ry * ry + rz * rz);
    if ((double)dist > c->collision_radius || (double)bi->age < d->kid_age || (double)bj->age < d->kid_age) return 0;
    if ((double)bi->age > d->particle_life || (double)bj->age > d->particle_life) return 0;
    if (bi->id > bj->id) return 1;
    if (bi->id < bj->id) return 2;
    return 0;
}

static float clamp_mag(float v, float lim) /* ps.cpp:1279-1281, 1294-1296 */
{
    if (fabsf(v) > lim) v = lim * (v / fabsf(v));
    return v;
}

/* ps.cpp:1268-1302.  dx = v*t (fp32) + 0.5*a*t*t (double), rounded once to
 * float; v += a*t in fp32; both clamped per axis; age += t. */
void pso_integrate(const pso_config *c, const pso_derived *d, pso_particle *p)
{
    float t = (float)c->dt;
    float dx = (float)(p->vx * t + 0.5 * p->ax * t * t);
    float dy = (float)(p->vy * t + 0.5 * p->ay * t * t);
    float dz = (float)(p->vz * t + 0.5 * p->az * t * t);
    float dmaxr = (float)d->max_dx, maxv = (float)c->max_v;
    float rx, ry, rz, vx, vy, vz;
    dx = clamp_mag(dx, dmaxr); dy = clamp_mag(dy, dmaxr); dz = clamp_mag(dz, dmaxr);
    rx = p->x + dx; ry = p->y + dy; rz = p->z + dz;
    pso_set_pos_x(c, d, p, rx, ry, rz);
    vx = p->vx + p->ax * t; vy = p->vy + p->ay * t; vz = p->vz + p->az * t;
    p->vx = clamp_mag(vx, maxv); p->vy = clamp_mag(vy, maxv); p->vz = clamp_mag(vz, maxv);
    p->age += t;
}

/* ------------------------------------------------------------ lifecycle */

pso_system *pso_create(const pso_config *cfg)
{
    pso_system *s = (pso_system *)calloc(1, sizeof *s);
    int i, k, tid;
    if (!s) return NULL;
    s->cfg = *cfg;
    if (pso_derive(cfg, &s->d) != 0) { free(s); return NULL; }
    s->particles  = (pso_particle *)calloc((size_t)s->d.container_size, sizeof(pso_particle));
    s->tdata      = (pso_tdata *)calloc((size_t)s->d.container_size, sizeof(pso_tdata));
    s->queue      = (int *)calloc((size_t)s->d.container_size, sizeof(int));
    s->queue_info = (pso_queue_info *)calloc((size_t)s->d.queue_info_size, sizeof(pso_queue_info));
    s->chunkgrid  = (int *)calloc((size_t)s->d.num_chunks * (1 + (size_t)s->d.max_per_chunk), sizeof(int));
    s->cellgrid   = (int *)calloc((size_t)s->d.num_cells * (1 + (size_t)s->d.max_per_cell), sizeof(int));
    s->pkgdistrib = (pso_pair *)calloc((size_t)s->d.num_chunks * 27, sizeof(pso_pair));
    s->neib       = (int *)calloc((size_t)s->d.max_neib_particles, sizeof(int));
    if (!s->particles || !s->tdata || !s->queue || !s->queue_info || !s->chunkgrid ||
        !s->cellgrid || !s->pkgdistrib || !s->neib) { pso_destroy(s); return NULL; }
    s->explosions = 1;

    /* init_particles_host, ps.cpp:722-753 */
    for (i = 0; i < s->d.container_size; i++) {
        s->particles[i].id = i;
        pso_reset_particle(&s->particles[i]);
        s->tdata[i].id = i;
    }
    /* q_start_fast_host, ps.cpp:814-871: every slot is free, queue k covers
     * the slots of segment k in order */
    for (i = 0; i < s->d.container_size; i++) s->queue[i] = i;
    tid = 0;
    for (k = 0; k < 4; k++) {
        int j;
        for (j = 0; j < s->d.seg_count[k]; j++, tid++) {
            pso_queue_info *q = &s->queue_info[tid];
            int rloc = pso_get_cont_rloc(&s->d, 1 << k, j);
            q->front = rloc; q->rear = rloc + s->d.seg_size_t[k] - 1;
            q->count = s->d.seg_size_t[k]; q->lock = 0;
            q->rloc = rloc; q->seg_size = s->d.seg_size_t[k];
        }
    }
    /* pkg_distrib_host, ps.cpp:893-911 */
    for (i = 0; i < s->d.num_chunks; i++) pso_set_pkg_segments(&s->cfg, i, s->pkgdistrib + i * 27);
    return s;
}

void pso_destroy(pso_system *s)
{
    if (!s) return;
    free(s->particles); free(s->tdata); free(s->queue); free(s->queue_info);
    free(s->chunkgrid); free(s->cellgrid); free(s->pkgdistrib); free(s->neib);
    free(s);
}

const pso_config  *pso_get_config(const pso_system *s)  { return &s->cfg; }
const pso_derived *pso_get_derived(const pso_system *s) { return &s->d; }
pso_particle   *pso_particles(pso_system *s)      { return s->particles; }
pso_tdata      *pso_tdata_buf(pso_system *s)      { return s->tdata; }
int            *pso_queue(pso_system *s)          { return s->queue; }
pso_queue_info *pso_queue_info_buf(pso_system *s) { return s->queue_info; }
int            *pso_chunkgrid(pso_system *s)      { return s->chunkgrid; }
int            *pso_cellgrid(pso_system *s)       { return s->cellgrid; }
int            *pso_gridmax(pso_system *s)        { return s->gridmax; }
pso_pair       *pso_pkgdistrib(pso_system *s)     { return s->pkgdistrib; }
const pso_counters *pso_get_counters(const pso_system *s) { return &s->ctr; }
int pso_step_index(const pso_system *s) { return s->step; }
void pso_set_rng(pso_system *s, pso_rng_fn fn, void *user) { s->rng = fn; s->rng_user = user; }
void pso_set_explosions(pso_system *s, int enabled) { s->explosions = enabled; }

int pso_live_count(const pso_system *s)
{
    int i, n = 0;
    for (i = 0; i < s->d.container_size; i++)
        if (s->particles[i].cell >= 0 && s->particles[i].cell < s->d.num_cells) n++;
    return n;
}

/* ps.cpp:915-960 */
int pso_fill_particle(pso_system *s, float x, float y, float z, float w, float age, float fert_age)
{
    int G = s->d.grid_dim, info[3], nid;
    double cs = s->cfg.cell_size;
    int i1 = (int)(floor((-1.0 * y) / cs) + (G / 2));
    int i2 = (int)(floor(( 1.0 * x) / cs) + (G / 2));
    int i3 = (int)(floor((-1.0 * z) / cs) + (G / 2));
    if (!((i1 >= 0 && i1 < G) && (i2 >= 0 && i2 < G) && (i3 >= 0 && i3 < G))) return -2;
    pso_get_cell_info(&s->d, &s->cfg, i3 * G * G + i1 * G + i2, info);
    nid = pso_q_remove(s->queue_info, s->queue, &s->d, info[1], info[2]);
    if (nid < 0) return -1;
    create_particle(&s->cfg, &s->d, &s->particles[nid], w, age, fert_age, x, y, z, 0.0f, 0.0f, 0.0f);
    return nid;
}

int pso_fill_particles(pso_system *s, int n, const float *xyz, const float *w,
                       const float *age, const float *fert_age, int *ids_out)
{
    int k;
    for (k = 0; k < n; k++) {
        int id = pso_fill_particle(s, xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2],
                                   w ? w[k] : (float)s->cfg.particle_weight,
                                   age ? age[k] : 0.0f, fert_age ? fert_age[k] : 0.0f);
        if (id < 0) return k;
        if (ids_out) ids_out[k] = id;
    }
    return n;
}

/* ----------------------------------------------------------- the stages */

/* init_iframe_host, ps.cpp:1574-1606 */
void pso_init_iframe(pso_system *s)
{
    memset(s->chunkgrid, 0, sizeof(int) * (size_t)s->d.num_chunks * (1 + (size_t)s->d.max_per_chunk));
    memset(s->cellgrid, 0, sizeof(int) * (size_t)s->d.num_cells * (1 + (size_t)s->d.max_per_cell));
    s->gridmax[0] = 0; s->gridmax[1] = 0;
}

/* build_grid_host, ps.cpp:1468-1537: one pass over all slots in slot order */
void pso_build_grid(pso_system *s)
{
    const pso_derived *d = &s->d;
    int tid;
    for (tid = 0; tid < d->container_size; tid++) {
        pso_particle *p = &s->particles[tid];
        int c, i, old;
        if (!(p->cell >= 0 && p->cell < d->num_cells)) continue;

        s->tdata[tid].id = p->id;
        s->tdata[tid].x = p->x; s->tdata[tid].y = p->y; s->tdata[tid].z = p->z;
        s->tdata[tid].w = p->w; s->tdata[tid].age = p->age;

        c = 1 + d->max_per_chunk; i = p->chunk;
        old = s->chunkgrid[(size_t)i * c]++;
        if (old < d->max_per_chunk) {
            s->chunkgrid[(size_t)i * c + (old + 1)] = p->id;
            if (old + 1 > s->gridmax[0]) s->gridmax[0] = old + 1;
        }
        c = 1 + d->max_per_cell; i = p->cell;
        old = s->cellgrid[(size_t)i * c]++;
        if (old < d->max_per_cell) {
            s->cellgrid[(size_t)i * c + (old + 1)] = p->id;
            if (old + 1 > s->gridmax[1]) s->gridmax[1] = old + 1;
        } else {
            /* ps.cpp:1517-1526: the cell is full, the particle is killed.  The
             * reference resets first and then frees with the (now -1,-1)
             * segment, which get_info_rloc maps to queue record 0. */
            s->cellgrid[(size_t)i * c]--;
            pso_reset_particle(p);
            pso_q_insert(s->queue_info, s->queue, d, p->seg_type, p->seg_tid, p->id);
            s->ctr.cell_overflow_kills++;
        }
    }
}

/* neighbour id list of a cell: fill_cells + fill_particles, app.cu:370-452 */
static int gather_neighbours(const pso_system *s, int cell, int *neib)
{
    const pso_derived *d = &s->d;
    const int cstride = 1 + d->max_per_cell;
    int cells[27], ncell = pso_fill_cells(d, cell, cells), nn = 0, i;
    for (i = 0; i < ncell; i++) {
        const int *cl = s->cellgrid + (size_t)cells[i] * cstride;
        int t, cnt = cl[0];
        for (t = 1; t <= cnt; t++)
            if (nn < d->max_neib_particles) neib[nn++] = cl[t];
    }
    return nn;
}

/* death + collision scan (ps.cpp:1182-1208) and force loop (ps.cpp:1247-1259) of one
 * particle over the gathered list; nothing is modified.  Returns the collision flag. */
static int scan_and_accumulate(const pso_system *s, const pso_particle *me, const int *neib, int nn, float acc[3],
                               int *died_of_age)
{
    const pso_config *c = &s->cfg;
    const pso_derived *d = &s->d;
    int collision_flag = 0, i;
    acc[0] = acc[1] = acc[2] = 0.0f;
    *died_of_age = 0;
    if ((double)me->age > d->particle_life) { *died_of_age = 1; return 2; }
    for (i = 0; i < nn; i++) {
        const pso_tdata *nb = &s->tdata[neib[i]];
        int flag = 0;
        if (me->id != nb->id) flag = pso_body_body_collision(c, d, me, nb);
        if (flag > collision_flag) collision_flag = flag;
        if (collision_flag == 2) break;
    }
    if (collision_flag > 0) return collision_flag;
    for (i = 0; i < nn; i++) {
        const pso_tdata *nb = &s->tdata[neib[i]];
        if (me->id != nb->id) pso_body_body_interaction(c, d, me, nb, acc);
    }
    return 0;
}

/* everything calc_forces does with one particle once flag and acceleration are known:
 * kill / survive (ps.cpp:1210-1242), integrate (1261-1302), explosion (1306-1333),
 * relocation (1335-1374) */
static void finish_particle(pso_system *s, pso_particle *me, int collision_flag, int died_of_age, const float acc[3])
{
    const pso_config *c = &s->cfg;
    const pso_derived *d = &s->d;
    const int id = me->id;
    int seg[2];

    if (collision_flag == 2) {
        if (died_of_age) s->ctr.deaths_age++; else s->ctr.deaths_collision++;
        pso_get_id_info(d, id, seg);
        pso_reset_particle(me);
        pso_q_insert(s->queue_info, s->queue, d, seg[0], seg[1], id);
        return;
    }
    if (collision_flag == 1) { pso_survive_particle(me); s->ctr.survives++; return; }

    me->ax = acc[0]; me->ay = acc[1]; me->az = acc[2];
    pso_integrate(c, d, me);
    s->ctr.integrated++;

    if (s->explosions && (me->age >= me->fertility_age) && !me->is_parent) {
        if (!s->rng) {
            s->ctr.explosions_skipped++;
        } else {
            int ri[3], nid; double u = 0.0;
            float ux, uy, uz, mag, vx, vy, vz;
            s->rng(s->rng_user, id, s->step, ri, &u);
            ux = (float)(ri[0] * 1.0); uy = (float)(ri[1] * 1.0); uz = (float)(ri[2] * 1.0);
            mag = sqrtf((float)(ux * ux * 1.0 + uy * uy * 1.0 + uz * uz * 1.0)); /* ps.cpp:50 */
            ux /= mag; uy /= mag; uz /= mag;
            vx = (float)(ux * c->explosion_speed);
            vy = (float)(uy * c->explosion_speed);
            vz = (float)(uz * c->explosion_speed);
            me->is_parent = 1;
            me->vx = vx; me->vy = vy; me->vz = vz;
            nid = pso_q_remove(s->queue_info, s->queue, d, me->seg_type, me->seg_tid);
            if (nid >= 0) {
                float lo = (float)d->min_fertility_age, hi = (float)d->max_fertility_age;
                float fert = (float)(lo + u * (hi - lo)); /* ps.cpp:29-36 */
                create_particle(c, d, &s->particles[nid], (float)c->particle_weight, 0.0f, fert,
                                me->x, me->y, me->z,
                                (float)(-1.0 * vx), (float)(-1.0 * vy), (float)(-1.0 * vz));
                s->ctr.births++;
            } else s->ctr.births_failed++;
        }
    }

    if (me->seg_fault) {
        int nid;
        pso_get_id_info(d, id, seg);
        nid = pso_q_remove(s->queue_info, s->queue, d, me->seg_type, me->seg_tid);
        if (nid >= 0) {
            pso_particle *dst = &s->particles[nid];
            int keep = dst->id;          /* copy_particle, app.cu:232-237 */
            *dst = *me; dst->id = keep;
            dst->seg_fault = 0;
            s->ctr.relocations++;
        } else s->ctr.relocations_lost++;
        pso_reset_particle(me);
        pso_q_insert(s->queue_info, s->queue, d, seg[0], seg[1], id);
    }
}

/* calc_forces_host for one chunk, ps.cpp:1120-1383 */
void pso_calc_forces_chunk(pso_system *s, int chunk, int subtask_elems)
{
    const pso_derived *d = &s->d;
    const int *row = s->chunkgrid + (size_t)chunk * (1 + d->max_per_chunk);
    int tid;
    for (tid = 0; tid < subtask_elems; tid++) {
        int chunk_size = row[0], pid, nn, flag, aged;
        float acc[3];
        pso_particle *me;
        if (tid > chunk_size - 1) continue;
        if (tid >= d->max_per_chunk) continue; /* beyond what build_grid stored */
        pid = row[tid + 1];
        if (pid < 0) continue;
        me = &s->particles[pid];
        if (!(me->cell >= 0 && me->cell < d->num_cells)) continue;
        nn = gather_neighbours(s, me->cell, s->neib);
        flag = scan_and_accumulate(s, me, s->neib, nn, acc, &aged);
        finish_particle(s, me, flag, aged, acc);
    }
}

/* ---- the same stage cut in two (test support for the sharded multi-GPU path) ---- */

int pso_sorted_count(const pso_system *s)
{
    int c, n = 0;
    for (c = 0; c < s->d.num_cells; c++) n += s->cellgrid[(size_t)c * (1 + s->d.max_per_cell)];
    return n;
}

static void calc_pairs_range(const pso_system *s, int lo, int hi, float *force4, int *neib)
{
    const pso_derived *d = &s->d;
    const int cstride = 1 + d->max_per_cell;
    int c, gi = 0;
    for (c = 0; c < d->num_cells; c++) {
        const int *cl = s->cellgrid + (size_t)c * cstride;
        int t, cnt = cl[0];
        if (gi + cnt <= lo || gi >= hi) { gi += cnt; continue; }
        for (t = 1; t <= cnt; t++, gi++) {
            const pso_particle *me = &s->particles[cl[t]];
            int nn, flag, aged;
            float acc[3];
            union { int i; float f; } bits;
            if (gi < lo || gi >= hi) continue;
            nn = gather_neighbours(s, c, neib);
            flag = scan_and_accumulate(s, me, neib, nn, acc, &aged);
            bits.i = flag;
            force4[4 * gi] = acc[0]; force4[4 * gi + 1] = acc[1]; force4[4 * gi + 2] = acc[2];
            force4[4 * gi + 3] = bits.f;
        }
    }
}

void pso_calc_pairs(pso_system *s, int lo, int hi, float *force4)
{
    calc_pairs_range(s, lo, hi, force4, s->neib);
}

/* The same read-only pass on several host threads (contiguous shares of [lo, hi), one
 * scratch list each): what pmlib gets from running chunk subtasks on all cores.  Used by
 * bench.py for the all-core CPU figure; results are those of pso_calc_pairs. */
typedef struct { const pso_system *s; int lo, hi; float *force4; int *neib; } pairs_job;

static void *pairs_worker(void *arg)
{
    pairs_job *j = (pairs_job *)arg;
    calc_pairs_range(j->s, j->lo, j->hi, j->force4, j->neib);
    return NULL;
}

int pso_calc_pairs_threads(pso_system *s, int lo, int hi, float *force4, int nthreads)
{
    pthread_t *tid;
    pairs_job *job;
    int k, started = 0, rc = 0;
    if (nthreads < 1) nthreads = 1;
    if (hi < lo) hi = lo;
    tid = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    job = (pairs_job *)calloc((size_t)nthreads, sizeof(pairs_job));
    if (!tid || !job) { free(tid); free(job); return -1; }
    for (k = 0; k < nthreads; k++) {
        job[k].s = s; job[k].force4 = force4;
        job[k].lo = lo + (int)((long long)(hi - lo) * k / nthreads);
        job[k].hi = lo + (int)((long long)(hi - lo) * (k + 1) / nthreads);
        job[k].neib = (int *)malloc(sizeof(int) * (size_t)s->d.max_neib_particles);
        if (!job[k].neib || pthread_create(&tid[k], NULL, pairs_worker, &job[k]) != 0) { rc = -1; break; }
        started++;
    }
    for (k = 0; k < started; k++) pthread_join(tid[k], NULL);
    for (k = 0; k < nthreads; k++) free(job[k].neib);
    free(tid); free(job);
    return rc;
}

void pso_apply_forces(pso_system *s, const float *force4)
{
    const pso_derived *d = &s->d;
    const int cstride = 1 + d->max_per_cell;
    /* sorted index of every slot that is in a cell list */
    int *rank = (int *)malloc(sizeof(int) * (size_t)d->container_size);
    int c, gi = 0, ch, biggest = s->gridmax[0];
    if (!rank) return;
    for (c = 0; c < d->container_size; c++) rank[c] = -1;
    for (c = 0; c < d->num_cells; c++) {
        const int *cl = s->cellgrid + (size_t)c * cstride;
        int t;
        for (t = 1; t <= cl[0]; t++) rank[cl[t]] = gi++;
    }
    /* the reference's serial order: chunk by chunk, chunk list order (ps.cpp:1900-1912, 1140-1163) */
    for (ch = 0; ch < d->num_chunks; ch++) {
        const int *row = s->chunkgrid + (size_t)ch * (1 + d->max_per_chunk);
        int tid;
        for (tid = 0; tid < biggest; tid++) {
            pso_particle *me;
            union { int i; float f; } bits;
            float acc[3];
            int pid, k, aged;
            if (tid > row[0] - 1 || tid >= d->max_per_chunk) continue;
            pid = row[tid + 1];
            if (pid < 0) continue;
            me = &s->particles[pid];
            if (!(me->cell >= 0 && me->cell < d->num_cells)) continue;
            k = rank[pid];
            if (k < 0) continue;
            acc[0] = force4[4 * k]; acc[1] = force4[4 * k + 1]; acc[2] = force4[4 * k + 2];
            bits.f = force4[4 * k + 3];
            aged = (double)me->age > d->particle_life;
            finish_particle(s, me, bits.i, aged, acc);
        }
    }
    free(rank);
}

/* ---- the life cycle with its queue operations deferred (test support for the slab-partitioned
 * multi-GPU path).  pso_apply_collect does what pso_apply_forces does, in the same serial order,
 * except that nothing touches a free-slot queue and nothing is placed in a new slot: every
 * q_insert / q_remove the reference would execute is recorded as an operation keyed by its
 * place in that serial order (chunk, slot, sub-step: birth remove 0, relocation remove 1,
 * insert 2 -- ps.cpp:1232, 1319, 1355, 1369).  pso_replay_ops then executes a set of operations
 * queue by queue in key order and places the particles.  Within one queue that is exactly the
 * order pso_apply_forces executes them in, and queues do not interact, so
 *     collect + replay  ==  pso_apply_forces                (tests/test_oracle_deferred.py)
 * and the operations on a queue may come from several systems that each hold a slab, as long
 * as all of them are handed to the queue's owner. */
static unsigned long long op_key(int chunk, int slot, int sub)
{
    return ((unsigned long long)(unsigned)(chunk + 1) << 34) | ((unsigned long long)(unsigned)slot << 2) | (unsigned)sub;
}

int pso_apply_collect(pso_system *s, const float *force4, pso_op *ops, int cap)
{
    const pso_config *c = &s->cfg;
    const pso_derived *d = &s->d;
    const int cstride = 1 + d->max_per_cell;
    int *rank = (int *)malloc(sizeof(int) * (size_t)d->container_size);
    int cc, gi = 0, ch, biggest = s->gridmax[0], n = 0;
    if (!rank) return -1;
    for (cc = 0; cc < d->container_size; cc++) rank[cc] = -1;
    for (cc = 0; cc < d->num_cells; cc++) {
        const int *cl = s->cellgrid + (size_t)cc * cstride;
        int t;
        for (t = 1; t <= cl[0]; t++) rank[cl[t]] = gi++;
    }
    for (ch = 0; ch < d->num_chunks; ch++) {
        const int *row = s->chunkgrid + (size_t)ch * (1 + d->max_per_chunk);
        int tid;
        for (tid = 0; tid < biggest; tid++) {
            pso_particle *me;
            union { int i; float f; } bits;
            int pid, k, aged, seg[2], old_cell;
            if (tid > row[0] - 1 || tid >= d->max_per_chunk) continue;
            pid = row[tid + 1];
            if (pid < 0) continue;
            me = &s->particles[pid];
            if (!(me->cell >= 0 && me->cell < d->num_cells)) continue;
            k = rank[pid];
            if (k < 0) continue;
            bits.f = force4[4 * k + 3];
            aged = (double)me->age > d->particle_life;
            old_cell = me->cell;
            if (n + 3 > cap) { free(rank); return -2; }
            if (bits.i == 2) {                                   /* kill, ps.cpp:1210-1235 */
                if (aged) s->ctr.deaths_age++; else s->ctr.deaths_collision++;
                pso_get_id_info(d, pid, seg);
                pso_reset_particle(me);
                ops[n].key = op_key(ch, pid, 2); ops[n].rec = pso_get_info_rloc(d, seg[0], seg[1]);
                ops[n].kind = 0; ops[n].slot = pid; ops[n].dst = -1; ops[n].old_cell = old_cell; n++;
                continue;
            }
            if (bits.i == 1) { pso_survive_particle(me); s->ctr.survives++; continue; }
            me->ax = force4[4 * k]; me->ay = force4[4 * k + 1]; me->az = force4[4 * k + 2];
            pso_integrate(c, d, me);
            s->ctr.integrated++;
            if (s->explosions && (me->age >= me->fertility_age) && !me->is_parent) {   /* ps.cpp:1306-1333 */
                if (!s->rng) s->ctr.explosions_skipped++;
                else {
                    int ri[3]; double u = 0.0;
                    float ux, uy, uz, mag, vx, vy, vz, lo, hi;
                    s->rng(s->rng_user, pid, s->step, ri, &u);
                    ux = (float)(ri[0] * 1.0); uy = (float)(ri[1] * 1.0); uz = (float)(ri[2] * 1.0);
                    mag = sqrtf((float)(ux * ux * 1.0 + uy * uy * 1.0 + uz * uz * 1.0));
                    ux /= mag; uy /= mag; uz /= mag;
                    vx = (float)(ux * c->explosion_speed); vy = (float)(uy * c->explosion_speed); vz = (float)(uz * c->explosion_speed);
                    me->is_parent = 1;
                    me->vx = vx; me->vy = vy; me->vz = vz;
                    lo = (float)d->min_fertility_age; hi = (float)d->max_fertility_age;
                    ops[n].key = op_key(ch, pid, 0); ops[n].rec = pso_get_info_rloc(d, me->seg_type, me->seg_tid);
                    ops[n].kind = 2; ops[n].slot = pid; ops[n].dst = -1; ops[n].old_cell = old_cell;
                    memset(&ops[n].body, 0, sizeof(pso_particle));
                    create_particle(c, d, &ops[n].body, (float)c->particle_weight, 0.0f, (float)(lo + u * (hi - lo)),
                                    me->x, me->y, me->z, (float)(-1.0 * vx), (float)(-1.0 * vy), (float)(-1.0 * vz));
                    n++;
                }
            }
            if (me->seg_fault) {                                /* ps.cpp:1335-1374 */
                pso_get_id_info(d, pid, seg);
                ops[n].key = op_key(ch, pid, 1); ops[n].rec = pso_get_info_rloc(d, me->seg_type, me->seg_tid);
                ops[n].kind = 1; ops[n].slot = pid; ops[n].dst = -1; ops[n].old_cell = old_cell; ops[n].body = *me; n++;
                pso_reset_particle(me);
                ops[n].key = op_key(ch, pid, 2); ops[n].rec = pso_get_info_rloc(d, seg[0], seg[1]);
                ops[n].kind = 0; ops[n].slot = pid; ops[n].dst = -1; ops[n].old_cell = old_cell; n++;
            }
        }
    }
    free(rank);
    return n;
}

static int op_order(const void *a, const void *b)
{
    const pso_op *x = (const pso_op *)a, *y = (const pso_op *)b;
    if (x->rec != y->rec) return x->rec < y->rec ? -1 : 1;
    return x->key < y->key ? -1 : x->key > y->key ? 1 : 0;
}

void pso_replay_ops(pso_system *s, pso_op *ops, int n)
{
    const pso_config *c = &s->cfg;
    const pso_derived *d = &s->d;
    int i;
    qsort(ops, (size_t)n, sizeof(pso_op), op_order);
    for (i = 0; i < n; i++) {
        pso_op *o = &ops[i];
        pso_queue_info *q = &s->queue_info[o->rec];
        if (o->kind == 0) {                                      /* q_insert, app_common.cu:346-376 */
            if (q->count == q->seg_size) continue;
            if (q->count == 0) { q->front = q->rloc; q->rear = q->rloc; }
            else if (q->rear == q->rloc + q->seg_size - 1) q->rear = q->rloc;
            else q->rear++;
            q->count++;
            s->queue[q->rear] = o->slot;
        } else {                                                 /* q_remove, app_common.cu:305-339 */
            int item = -1;
            if (q->count > 0) {
                int pos = q->front;
                if (q->count == 1) { q->front = -1; q->rear = -1; }
                else if (q->front == q->rloc + q->seg_size - 1) q->front = q->rloc;
                else q->front++;
                q->count--;
                item = s->queue[pos];
                s->queue[pos] = -1;
            }
            o->dst = item;
            if (o->kind == 1) {
                if (item >= 0) {
                    pso_particle *dst = &s->particles[item];
                    int keep = dst->id;
                    *dst = o->body; dst->id = keep; dst->seg_fault = 0;
                    memset((char *)dst + 22, 0, 2);          /* the two pad bytes: an op list that went through numpy carries noise there */
                    s->ctr.relocations++;
                } else s->ctr.relocations_lost++;
            } else {
                if (item >= 0) {
                    const pso_particle *b = &o->body;
                    create_particle(c, d, &s->particles[item], b->w, b->age, b->fertility_age, b->x, b->y, b->z, b->vx, b->vy, b->vz);
                    s->ctr.births++;
                } else s->ctr.births_failed++;
            }
        }
    }
}

void pso_advance_step(pso_system *s) { s->step++; }

/* the batches of ps.cpp:1900-1912 visit chunks 0..NUM_CHUNKS-1 in order */
void pso_calc_forces(pso_system *s)
{
    int biggest = s->gridmax[0], ch;
    if (biggest > 0)
        for (ch = 0; ch < s->d.num_chunks; ch++) pso_calc_forces_chunk(s, ch, biggest);
}

void pso_step(pso_system *s, int nsteps) /* ps.cpp:1843-1928 */
{
    int k;
    for (k = 0; k < nsteps; k++) {
        pso_init_iframe(s);
        pso_build_grid(s);
        pso_calc_forces(s);
        s->step++;
    }
}
