// pairs.hip -- calc_forces' two neighbour loops: collision flags, then forces (ps.cpp:1182-1263)
#include "kernels_common.hpp"

namespace psamd {

// ------------------------------------------------------------------ pair kernel
// Correctly rounded fp32 sqrt and reciprocal without the range/denormal scaffolding
// the compiler wraps around them: valid for normal inputs well inside the exponent
// range (the host only selects them when eps2^3 .. (3 (2L)^2 + eps2)^3 lies in
// [2^-60, 2^60]).  Each is one hardware estimate (v_rsq_f32 / v_rcp_f32, 1 ulp) plus
// one residual correction, and each is checked against the compiler's correctly
// rounded form over EVERY float of [2^-62, 2^62] by psamd_selftest_math
// (tests/test_gpu_math.py): zero mismatches.
__device__ __forceinline__ float sqrt_rn_short(float a)
{
    const float r = __builtin_amdgcn_rsqf(a);
    const float g = a * r;                      // ~sqrt(a)
    const float h = 0.5f * r;                   // ~1 / (2 sqrt(a))
    const float d = __builtin_fmaf(-g, g, a);   // exact residual
    return __builtin_fmaf(d, h, g);
}

__device__ __forceinline__ float rcp_rn_newton(float q)
{
    const float x = __builtin_amdgcn_rcpf(q);
    const float e = __builtin_fmaf(-q, x, 1.0f);
    return __builtin_fmaf(e, x, x);
}

// RN(1 / RN(sqrt(a))): the reference's 1.0f / sqrtf(a), two roundings.
__device__ __forceinline__ float inv_sqrt_selected(float six)
{
    return rcp_rn_newton(sqrt_rn_short(six));
}

// One transcendental instead of two: the reciprocal's Newton step starts from the rsq estimate
// itself (r ~ 1/sqrt(a) ~ 1/s).  That is RN(1/s) for every float of the range EXCEPT where s has
// an all-ones mantissa (1/s lies a hair above a rounding tie and the step lands on the tie: 124
// inputs in [2^-62, 2^62]); there the residual e is exactly 2^-24, which is what `tie` reports so
// that the caller can redo the group with inv_sqrt_selected.  Checked for every float of the
// range by psamd_selftest_math: no mismatch that is not reported.  v_rcp_f32 costs 3.3 issue
// slots on gfx950 (profiles/r1_microbench_valu_rates.txt), the compare one.
__device__ __forceinline__ float inv_sqrt_guarded(float a, bool &tie)
{
    const float r = __builtin_amdgcn_rsqf(a);
    const float g = a * r, h = 0.5f * r;
    const float s = __builtin_fmaf(__builtin_fmaf(-g, g, a), h, g);
    const float e = __builtin_fmaf(-s, r, 1.0f);
    tie = tie || e == 0x1p-24f;
    return __builtin_fmaf(e, r, r);
}

// A tempting shortcut that is NOT exact, kept only so the self test can show it: start
// the reciprocal's Newton step from the rsq estimate (2h ~ 1/g) instead of a second
// transcendental.  124 of the 1.04e9 floats in range come out one ulp off.
__device__ __forceinline__ float inv_sqrt_one_transcendental(float a)
{
    const float r = __builtin_amdgcn_rsqf(a);
    float g = a * r, h = 0.5f * r;
    const float e = __builtin_fmaf(-h, g, 0.5f);
    h = __builtin_fmaf(h, e, h);
    g = __builtin_fmaf(g, e, g);
    const float q = __builtin_fmaf(__builtin_fmaf(-g, g, a), h, g);
    float x = h + h;
    for (int k = 0; k < 2; k++) x = __builtin_fmaf(__builtin_fmaf(-q, x, 1.0f), x, x);
    return x;
}

// bodyBodyInteraction, app_common.cu:236-267, for a snapshot body q = (x,y,z,w_eff).
__device__ __forceinline__ float pair_exact(float xi, float yi, float zi, const float4 q, double eps2,
                                            float &ax, float &ay, float &az)
{
    const float rx = q.x - xi, ry = q.y - yi, rz = q.z - zi;
    const float d2 = rx * rx + ry * ry + rz * rz;
    const float dsq = (float)((double)d2 + eps2);      // EPS2 is a double literal
    const float six = dsq * dsq * dsq;
    const float inv = 1.0f / sqrtf(six);               // correctly rounded sqrt, then divide
    const float s = q.w * inv;
    ax += rx * s; ay += ry * s; az += rz * s;
    return d2;
}

// Same physics with fused multiply-adds and the hardware reciprocal square root:
// differs from the reference in the last bits (PSAMD_FLAG_FAST_MATH).
__device__ __forceinline__ float pair_fast(float xi, float yi, float zi, const float4 q, float eps2,
                                           float &ax, float &ay, float &az)
{
    const float rx = q.x - xi, ry = q.y - yi, rz = q.z - zi;
    const float d2 = fmaf(rz, rz, fmaf(ry, ry, rx * rx));
    const float dsq = d2 + eps2;
    const float rinv = __builtin_amdgcn_rsqf(dsq);
    const float s = q.w * (rinv * rinv * rinv);
    ax = fmaf(rx, s, ax); ay = fmaf(ry, s, ay); az = fmaf(rz, s, az);
    return d2;
}

// bodyBodyCollision, app_common.cu:269-301, evaluated exactly for the few pairs whose
// squared distance passes the gate.  0 none, 1 survive (higher id), 2 kill (lower id).
__device__ __forceinline__ int collide_exact(const DevParams &P, float d2, float age_i, int id_i,
                                             float age_j, int id_j)
{
    const float dist = sqrtf(d2);
    if ((double)dist > P.coll_radius || (double)age_i < P.kid_age || (double)age_j < P.kid_age) return 0;
    if ((double)age_i > P.life || (double)age_j > P.life) return 0;
    if (id_i > id_j) return 1;
    if (id_i < id_j) return 2;
    return 0;
}

// Lean exact pair arithmetic for k_pairs<1>.  The reference adds the double literal EPS2
// in double and rounds to float; from eps_f32_from upwards a plain fp32 add gives the same
// bits (checked for every such float when the context is created).  A wave takes the
// slow branch only when one of its lanes holds a pair closer than `slow_below` =
// max(eps_f32_from, collision gate): there EPS2 is added in double and the exact collision
// rule is evaluated for the pairs inside the gate, so the common path carries neither.
struct PairCtx {
    float xi, yi, zi, age_i;
    int id_i, gi;
    bool scan;
};

typedef float v2f __attribute__((ext_vector_type(2)));

// Two pairs per instruction slot: gfx950 has packed fp32 add/mul/fma, and the SoA tile hands
// (x_j, x_j+1) over in one aligned register pair, so nothing is shuffled between registers.
// Every packed operation rounds each half exactly like its scalar form.
__device__ __forceinline__ v2f inv_sqrt_selected2(v2f six)
{
    v2f r; r.x = __builtin_amdgcn_rsqf(six.x); r.y = __builtin_amdgcn_rsqf(six.y);
    const v2f g = six * r, h = 0.5f * r;
    const v2f s = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, g, six), h, g);   // sqrt_rn_short
    v2f x; x.x = __builtin_amdgcn_rcpf(s.x); x.y = __builtin_amdgcn_rcpf(s.y);
    const v2f one = {1.0f, 1.0f};
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-s, x, one), x, x);          // rcp_rn_newton
}

// inv_sqrt_guarded on two pairs
__device__ __forceinline__ v2f inv_sqrt_guarded2(v2f six, bool &tie)
{
    v2f r; r.x = __builtin_amdgcn_rsqf(six.x); r.y = __builtin_amdgcn_rsqf(six.y);
    const v2f g = six * r, h = 0.5f * r;
    const v2f s = __builtin_elementwise_fma(__builtin_elementwise_fma(-g, g, six), h, g);
    const v2f one = {1.0f, 1.0f};
    const v2f e = __builtin_elementwise_fma(-s, r, one);
    tie = tie || e.x == 0x1p-24f || e.y == 0x1p-24f;
    return __builtin_elementwise_fma(e, r, r);
}

// NQ pairs in two stages, so that a caller can start fetching the next group's bodies
// between them: distances first (the only use of the positions), then everything else.
template <int NQ>
struct PairRows {
    v2f rx[NQ / 2], ry[NQ / 2], rz[NQ / 2], d[NQ / 2];
    float dm;                                   // smallest d of the group
};

// SOFTENED: d = fma chain started at eps2 (fast math); else the reference's unfused r.r
template <int NQ, bool SOFTENED>
__device__ __forceinline__ void pairs_dist(const PairCtx &c, const v2f (&qx)[NQ / 2], const v2f (&qy)[NQ / 2],
                                           const v2f (&qz)[NQ / 2], float eps2, PairRows<NQ> &r)
{
    const v2f xi = {c.xi, c.xi}, yi = {c.yi, c.yi}, zi = {c.zi, c.zi}, eps = {eps2, eps2};
    r.dm = 3.0e38f;
#pragma unroll
    for (int i = 0; i < NQ / 2; i++) {
        r.rx[i] = qx[i] - xi; r.ry[i] = qy[i] - yi; r.rz[i] = qz[i] - zi;
        if (SOFTENED)
            r.d[i] = __builtin_elementwise_fma(r.rz[i], r.rz[i], __builtin_elementwise_fma(r.ry[i], r.ry[i], __builtin_elementwise_fma(r.rx[i], r.rx[i], eps)));
        else
            r.d[i] = r.rx[i] * r.rx[i] + r.ry[i] * r.ry[i] + r.rz[i] * r.rz[i];
        r.dm = fminf(fminf(r.dm, r.d[i].x), r.d[i].y);
    }
}

// ONE_T: one transcendental per pair (inv_sqrt_guarded2) -- fewer issue slots, for passes that are
// throughput-bound (four or more waves per SIMD: -3.7 % on the N = 2^20 force pass); the two-
// transcendental form has the shorter dependency chain and wins where a SIMD holds one or two waves
// (a 1/8 slab's tile walk: 0.57 against 0.64 ms).
// In two halves, so that a walk with a SIMD (almost) to itself can put the NEXT group's distances between them:
// pairs_scale_exact -- every pair's w / d^3 (the branches are in here) -- and pairs_add, the ordered additions.
template <int NQ, bool ONE_T>
__device__ __forceinline__ void pairs_scale_exact(const DevParams &P, const PairCtx &c, const PairRows<NQ> &r,
                                                  const v2f (&qw)[NQ / 2], int gj0,
                                                  const float *__restrict__ snap_age,
                                                  const int *__restrict__ sorted_id,
                                                  v2f (&sc)[NQ / 2], int &flag)
{
    constexpr int H = NQ / 2;
    v2f e[H];
    // One-pass stage (c.scan; a compile-time false in the two-pass force pass): a distance that is not a number --
    // the particle's own position or a body's is not one -- passes the reference's collision test, but the
    // group's minimum does not see it (fminf drops it): such a group takes the branch with the exact rule too.
    bool wild = false;
    if (c.scan) {
        v2f t = r.d[0];
#pragma unroll
        for (int i = 1; i < H; i++) t = t + r.d[i];
        const float tt = t.x + t.y;
        wild = tt != tt;
    }
    if (__any(r.dm < P.slow_below) || __any(wild)) {
#pragma unroll
        for (int i = 0; i < H; i++) {
            e[i].x = (float)((double)r.d[i].x + P.eps2);
            e[i].y = (float)((double)r.d[i].y + P.eps2);
        }
        if (c.scan && (wild || !(r.dm > P.coll_d2_gate))) {
#pragma unroll
            for (int i = 0; i < NQ; i++) {
                const float di = (i & 1) ? r.d[i >> 1].y : r.d[i >> 1].x;
                if (!(di > P.coll_d2_gate) && gj0 + i != c.gi)
                    flag = max(flag, collide_exact(P, di, c.age_i, c.id_i, snap_age[gj0 + i], sorted_id[gj0 + i]));
            }
        }
    } else {
        const v2f eps = {P.eps2f, P.eps2f};
#pragma unroll
        for (int i = 0; i < H; i++) e[i] = r.d[i] + eps;
    }
    if (!ONE_T) {
#pragma unroll
        for (int i = 0; i < H; i++) sc[i] = qw[i] * inv_sqrt_selected2(e[i] * e[i] * e[i]);
    } else {
        bool tie = false;
#pragma unroll
        for (int i = 0; i < H; i++) {
            sc[i] = inv_sqrt_guarded2(e[i] * e[i] * e[i], tie);
            // (keeps the step's last fma above the branch: sunk below it, its operands -- 16 VGPRs --
            // stay live across the branch and the kernel drops from 6 to 5 waves per SIMD)
            asm volatile("" : "+v"(sc[i]));
        }
        if (__any(tie)) {                               // about one group in 500
#pragma unroll
            for (int i = 0; i < H; i++) sc[i] = inv_sqrt_selected2(e[i] * e[i] * e[i]);
        }
#pragma unroll
        for (int i = 0; i < H; i++) sc[i] = qw[i] * sc[i];
    }
}

template <int NQ>
__device__ __forceinline__ void pairs_add(const PairRows<NQ> &r, const v2f (&sc)[NQ / 2], float &ax, float &ay, float &az)
{
#pragma unroll
    for (int i = 0; i < NQ / 2; i++) {                  // sums in list order
        const v2f px = r.rx[i] * sc[i], py = r.ry[i] * sc[i], pz = r.rz[i] * sc[i];
        ax += px.x; ay += py.x; az += pz.x;
        ax += px.y; ay += py.y; az += pz.y;
    }
}

template <int NQ, bool ONE_T>
__device__ __forceinline__ void pairs_finish_exact(const DevParams &P, const PairCtx &c, const PairRows<NQ> &r,
                                                   const v2f (&qw)[NQ / 2], int gj0,
                                                   const float *__restrict__ snap_age,
                                                   const int *__restrict__ sorted_id,
                                                   float &ax, float &ay, float &az, int &flag)
{
    v2f sc[NQ / 2];
    pairs_scale_exact<NQ, ONE_T>(P, c, r, qw, gj0, snap_age, sorted_id, sc, flag);
    pairs_add<NQ>(r, sc, ax, ay, az);
}

// Fast-math finish (FMA + v_rsq) on softened distances.
template <int NQ>
__device__ __forceinline__ void pairs_finish_fast(const PairRows<NQ> &r, const v2f (&qw)[NQ / 2],
                                                  float &ax, float &ay, float &az)
{
    constexpr int H = NQ / 2;
    v2f sc[H];
    // the group's transcendentals back to back: going from a transcendental to plain VALU work and back costs a
    // couple of cycles each way on gfx950 (profiles/r4_microbench_trans_overlap.txt: 8 v_rsq among 32 v_fma, one to
    // four, take 18 % longer than the same instructions grouped)
    v2f q[H];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < H; i++) { q[i].x = __builtin_amdgcn_rsqf(r.d[i].x); q[i].y = __builtin_amdgcn_rsqf(r.d[i].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < H; i++) sc[i] = qw[i] * (q[i] * q[i] * q[i]);
#pragma unroll
    for (int i = 0; i < H; i++) {
        ax = fmaf(r.rx[i].x, sc[i].x, ax); ay = fmaf(r.ry[i].x, sc[i].x, ay); az = fmaf(r.rz[i].x, sc[i].x, az);
        ax = fmaf(r.rx[i].y, sc[i].y, ax); ay = fmaf(r.ry[i].y, sc[i].y, ay); az = fmaf(r.rz[i].y, sc[i].y, az);
    }
}

#if defined(PSAMD_TWO_TRANSCENDENTALS)      // (A/B builds)
constexpr bool ONE_T_DEFAULT = false;
#else
constexpr bool ONE_T_DEFAULT = true;
#endif
template <int NQ, bool ONE_T = ONE_T_DEFAULT>
__device__ __forceinline__ void pairsN_exact_lean(const DevParams &P, const PairCtx &c, const v2f (&qx)[NQ / 2],
                                                  const v2f (&qy)[NQ / 2], const v2f (&qz)[NQ / 2],
                                                  const v2f (&qw)[NQ / 2], int gj0,
                                                  const float *__restrict__ snap_age,
                                                  const int *__restrict__ sorted_id,
                                                  float &ax, float &ay, float &az, int &flag)
{
    PairRows<NQ> r;
    pairs_dist<NQ, false>(c, qx, qy, qz, 0.f, r);
    pairs_finish_exact<NQ, ONE_T>(P, c, r, qw, gj0, snap_age, sorted_id, ax, ay, az, flag);
}

// returns the smallest softened squared distance (d2 + eps2) of the group, for the collision gate
template <int NQ>
__device__ __forceinline__ float pairsN_fast(const PairCtx &c, const v2f (&qx)[NQ / 2], const v2f (&qy)[NQ / 2],
                                             const v2f (&qz)[NQ / 2], const v2f (&qw)[NQ / 2], float eps2,
                                             float &ax, float &ay, float &az)
{
    PairRows<NQ> r;
    pairs_dist<NQ, true>(c, qx, qy, qz, eps2, r);
    pairs_finish_fast<NQ>(r, qw, ax, ay, az);
    return r.dm;
}

__device__ __forceinline__ void pair1_exact_lean(const DevParams &P, const PairCtx &c, const float4 q, int gj,
                                                 const float *__restrict__ snap_age,
                                                 const int *__restrict__ sorted_id,
                                                 float &ax, float &ay, float &az, int &flag)
{
    const float rx = q.x - c.xi, ry = q.y - c.yi, rz = q.z - c.zi;
    const float d2 = rx * rx + ry * ry + rz * rz;
    const float e = (float)((double)d2 + P.eps2);
    if (__any(c.scan && !(d2 > P.coll_d2_gate))) {
        if (c.scan && !(d2 > P.coll_d2_gate) && gj != c.gi)
            flag = max(flag, collide_exact(P, d2, c.age_i, c.id_i, snap_age[gj], sorted_id[gj]));
    }
    const float s = q.w * inv_sqrt_selected(e * e * e);
    ax += rx * s; ay += ry * s; az += rz * s;
}

// ------------------------------------------------------------------ two-pass pair stage
// The reference scans a particle's neighbours for collisions first and runs the force loop
// only if there was none (ps.cpp:1182-1263): a particle that dies or "survives" a collision
// this step is not integrated and its acceleration is never looked at.  In a dense cloud
// that is a large share (42 % in the first step of the N = 2^20 benchmark cloud).  The lean
// modes do the same: k_collide_cell settles every particle's flag from the few bodies that can
// reach it -- its own cell and the neighbours' bodies near the shared faces (the halo lists
// k_sort_cells filled) -- then k_build_active lists, per cell, the particles that still need
// a force, and the force pass walks the 27-cell stencil for those only.
//
// Bodies in the stencil of local cell (i1, i2, i3) that are no kids, counted by one wave.  For the particle
// whose own position is not a number: the lean force walks let a particle meet itself and the kids because
// r * 0 adds nothing -- not so when r is no number.  The reference skips both (ps.cpp:1258,
// app_common.cu:240-243): with no other body in the stencil the particle's sum is +0 (this count is 1:
// itself), with one it is no number either way.
__device__ __forceinline__ int stencil_adults(const DevParams &P, int i1, int i2, int i3, const int *__restrict__ cell_start,
                                              const float *__restrict__ snap_age)
{
    const int lane = threadIdx.x & 63;
    int total = 0;
    for (int k = 0; k < STENCIL; k++) {
        const int nc = __builtin_amdgcn_readfirstlane(local_cell(P, i3 + c_stencil[k][2], i1 + c_stencil[k][1], i2 + c_stencil[k][0]));
        if (nc < 0) continue;
        const int b = __builtin_amdgcn_readfirstlane(cell_start[nc]);
        const int n = __builtin_amdgcn_readfirstlane(min(cell_start[nc + 1] - b, P.max_per_cell));
        for (int j0 = 0; j0 < n; j0 += 64) {
            const int j = j0 + lane;
            total += __popcll(__ballot(j < n && !(snap_age[b + (j < n ? j : 0)] < P.kid_thr)));
        }
    }
    return total;
}

// One particle against `n` bodies given as arrays (wave-uniform pointers, so the loads are
// scalar loads).  bodyBodyCollision (app_common.cu:269-301) without a branch: the reference's
// test (double)sqrtf(r.r) > COLLISION_RADIUS is, sqrtf being correctly rounded and monotone,
// r.r > coll_d2_max for a float found by bisection when the context is created; a body that can
// never collide (kid, over age) carries cid = -1, otherwise its slot id; and "flag = max over
// the hits of (id_i > id_j ? 1 : 2)" is two lane masks: met someone with a higher id (2, the
// lower id dies), met someone with a lower one (1).  The particle itself drops out because
// neither id comparison holds for it.
__device__ __forceinline__ void collide_scan(const DevParams &P, float xi, float yi, float zi, int id_i, bool scan,
                                             const float *__restrict__ bx, const float *__restrict__ by,
                                             const float *__restrict__ bz, const int *__restrict__ bcid, int n,
                                             unsigned long long &hi_mask, unsigned long long &lo_mask)
{
    // hi_mask / lo_mask: lanes that met a body with a higher / lower id (wave-uniform words: the
    // bookkeeping is scalar work).  A group's sixteen bodies AND their ids arrive in one batch of scalar
    // loads; a body's test is its distance arithmetic and one compare, the two id compares happen only
    // for the body some lane is within reach of (about one in eight at the benchmark's density), behind
    // a scalar branch.  (Before: a group minimum first, then -- nearly every group has a hit -- a loop of
    // sixteen compare-and-branch steps and a scalar load of the hit's id that the walk had to wait
    // for; that bookkeeping cost as much as the arithmetic.  Two groups of loads in flight were
    // tried and were slower.)
    constexpr int NB = 16;
    const v2f x2 = {xi, xi}, y2 = {yi, yi}, z2 = {zi, zi};
    const float dmax = P.coll_d2_max;
    const unsigned uid = (unsigned)id_i;
    const unsigned long long scan_mask = __builtin_amdgcn_ballot_w64(scan);
    auto hit = [&](float d2, int cj) {
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(!(d2 > dmax)) & scan_mask;
        if (hm) {
            hi_mask |= hm & __builtin_amdgcn_ballot_w64(cj > id_i);
            lo_mask |= hm & __builtin_amdgcn_ballot_w64((unsigned)cj < uid);      // (a body that never collides carries -1: not below any id)
        }
    };
    int j = 0;
    // (Two groups of eight in flight -- the next group's loads issued before the current one is worked
    // through, since the wave spends half its cycles parked at s_waitcnt -- were measured again with the
    // ids in the batch: 189 us against 147 for flags + plan.  The double set of bodies costs SGPR spills.)
    for (; j + NB <= n; j += NB) {
        // all the group's loads and distances first (one batch of scalar loads, one wait), then the tests
        int cid[NB];
        v2f d[NB / 2];
#pragma unroll
        for (int i = 0; i < NB; i++) cid[i] = bcid[j + i];
#pragma unroll
        for (int i = 0; i < NB / 2; i++) {
            const v2f rx = v2f{bx[j + 2 * i], bx[j + 2 * i + 1]} - x2, ry = v2f{by[j + 2 * i], by[j + 2 * i + 1]} - y2,
                      rz = v2f{bz[j + 2 * i], bz[j + 2 * i + 1]} - z2;
            d[i] = rx * rx + ry * ry + rz * rz;
        }
#pragma unroll
        for (int i = 0; i < NB / 2; i++) { asm volatile("" : "+v"(d[i])); }      // (keeps the tests below the arithmetic: the loads stay one batch)
#pragma unroll
        for (int i = 0; i < NB / 2; i++) { hit(d[i].x, cid[2 * i]); hit(d[i].y, cid[2 * i + 1]); }
    }
    for (; j < n; j++) {
        const float rx = bx[j] - xi, ry = by[j] - yi, rz = bz[j] - zi;
        hit(rx * rx + ry * ry + rz * rz, bcid[j]);
    }
}

// The collision flags of every particle of the computed cells and, for the particles that will not be
// integrated or feel no force (kids), the final force4 record.  One workgroup per cell, with the
// candidates culled first.  A collision needs the two within COLLISION_RADIUS (0.4 against a 5.0 cell): of
// the ~400 bodies a cell's particle could meet (its cell's and the halo list's) a handful are near enough
// to be worth the arithmetic.  The workgroup bins those bodies (the ones that can collide at all: cid >= 0)
// on a grid of up to 10^3 bins over the cell's box grown by the halo reach -- a counting sort in LDS:
// census with the body's rank in its bin from the atomic's return, prefix, scatter of (x, y, z, id) rows,
// the bodies held in registers between the passes -- and a particle then tests the bodies of its bin and
// the bins around it only: nine runs (a row of three bins along x is one run of the sorted rows), nine
// bodies in all at the benchmark's density.  A bin is wider than the reach, so two bodies within it of
// each other are never more than one bin apart on any axis (the bin coordinate is a monotone function of
// the position, clamped into the grid); the test itself is the arithmetic of collide_scan on the same
// operands, and "any hit with a higher / a lower id" does not depend on the order the candidates come
// in: the flags are the same bits.  A cell with more bodies than the LDS rows hold, or whose halo list
// overflowed, takes collide_scan over everything.
// (Until round 3 this was one wave per 64-particle slice running collide_scan over all ~400 bodies, 123 us
// at N = 2^20; a workgroup per cell with the 400 bodies in LDS read back as broadcast rows was 133 us --
// a broadcast ds_read_b128 still occupies the LDS pipe for its 64 lanes.  With the bins: 41 us, of which
// the runs are 18.  Steps on the way, flags + plan: 146 us -> 98 (bins) -> 91 (bodies kept in registers,
// run bounds read in one batch) -> 69 (two bodies a turn, ids by max / min instead of a branch at a hit)
// -> 67 (the particle's own position and id from the binning registers); profiles/r3_ab_collide.txt.)
constexpr int COLL_NB = 10;
template <int CAP>
__global__ __launch_bounds__(256, CAP <= 1024 ? 6 : 3) void k_collide_cell(DevParams P, const int *__restrict__ cell_start,
                                                      const float *__restrict__ snap_soa, const float *__restrict__ snap_age,
                                                      const int *__restrict__ sorted_id, const int *__restrict__ snap_cid,
                                                      const int *__restrict__ halo_count, const float *__restrict__ halo_f,
                                                      const int *__restrict__ halo_id, int *__restrict__ active_list,
                                                      int *__restrict__ active_count, int *__restrict__ task_cost,
                                                      const ForceBuf force4)
{
    constexpr int KB = CAP / 256;                                 // bodies a thread bins (held in registers between the passes)
    __shared__ float4 s_body[CAP];
    __shared__ int s_bin[COLL_NB * COLL_NB * COLL_NB + 1];
    __shared__ int s_wtot[4];
    const int c = comp_cell(P, blockIdx.x);
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int base = __builtin_amdgcn_readfirstlane(cell_start[c]);
    const int cnt = __builtin_amdgcn_readfirstlane(min(cell_start[c + 1] - base, P.max_per_cell));
    if (cnt <= 0) return;
    const int nh = __builtin_amdgcn_readfirstlane(halo_count[c]);
    const size_t cap = (size_t)P.sorted_cap;
    const size_t hat = (size_t)c * HALO_CAP, hplane = (size_t)P.n_local_cells * HALO_CAP;
    int i1, i2, i3;
    cell_coords(P, c, i1, i2, i3);
    // the bins: nb per axis over [-reach, cell + reach) in the cell's own coordinates
    const float cs = (float)P.cell_size, reach = P.halo_reach * 1.01f + 1e-3f, box = cs + 2.0f * reach;
    const int nb = max(1, min(COLL_NB, (int)(box / (reach * 1.05f))));
    const float per_unit = (float)nb / box;
    const float ox = ((float)i2 - (float)(P.G / 2)) * cs - reach, oy = ((float)(P.G / 2) - (float)i1) * cs + reach,
                oz = ((float)(P.G / 2) - (float)i3) * cs + reach;      // u = x - ox, oy - y, oz - z: offsets into the grown box
    auto bin1 = [&](float u) { return max(0, min(nb - 1, (int)(u * per_unit))); };
    bool binned = nh <= HALO_CAP && cnt + nh <= CAP;
    const int nbins = nb * nb * nb;
    // what one force task of this cell walks: the population of its stencil (the last wave, while the others' loads fly)
    if (wv == 3) {
        int n = 0;
        if (lane < STENCIL) {
            const int nc = local_cell(P, i3 + c_stencil[lane][2], i1 + c_stencil[lane][1], i2 + c_stencil[lane][0]);
            if (nc >= 0) n = min(cell_start[nc + 1] - cell_start[nc], P.max_per_cell);
        }
        n = wave_incl_scan(n);
        if (lane == 63) task_cost[c] = n;
    }
    float4 q[KB];
    if (binned) {
        const int nbody = cnt + nh;
        // all the loads in one batch (the coordinates do not wait for the ids), the bins zeroed meanwhile
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int e = tid + 256 * k;
            const bool own = e < cnt;
            q[k] = make_float4(0.f, 0.f, 0.f, __int_as_float(-1));
            if (e < nbody) {
                q[k].w = __int_as_float(own ? snap_cid[base + e] : halo_id[hat + (e - cnt)]);
                q[k].x = own ? snap_soa[base + e] : halo_f[hat + (e - cnt)];
                q[k].y = own ? snap_soa[cap + base + e] : halo_f[hplane + hat + (e - cnt)];
                q[k].z = own ? snap_soa[2 * cap + base + e] : halo_f[2 * hplane + hat + (e - cnt)];
            }
        }
        // A candidate whose position is not a number passes the reference's distance test against every particle
        // that scans it (halo_dirs): no bins for this cell, collide_scan meets it with everything.
        bool wild = false;
#pragma unroll
        for (int k = 0; k < KB; k++) wild |= __float_as_int(q[k].w) >= 0 && !finite3(q[k].x, q[k].y, q[k].z);
        for (int b = tid; b <= nbins; b += 256) s_bin[b] = 0;
        if (__syncthreads_or(wild)) binned = false;
    }
    if (binned) {
        int bin[KB], rank[KB];
#pragma unroll
        for (int k = 0; k < KB; k++) {
            bin[k] = -1; rank[k] = 0;
            if (__float_as_int(q[k].w) >= 0) {
                bin[k] = (bin1(oz - q[k].z) * nb + bin1(oy - q[k].y)) * nb + bin1(q[k].x - ox);
                rank[k] = atomicAdd(&s_bin[bin[k]], 1);
            }
        }
        __syncthreads();
        // exclusive prefix over the bins: a run of bins per thread, the runs' totals through the waves
        const int per = (nbins + 255) / 256;
        const int b0 = min(nbins, tid * per), b1 = min(nbins, b0 + per);
        int mine = 0;
        for (int b = b0; b < b1; b++) mine += s_bin[b];
        const int incl = wave_incl_scan(mine);
        if (lane == 63) s_wtot[wv] = incl;
        __syncthreads();
        int run = incl - mine;
        for (int k = 0; k < wv; k++) run += s_wtot[k];
        for (int b = b0; b < b1; b++) { const int n = s_bin[b]; s_bin[b] = run; run += n; }
        if (tid == 255) s_bin[nbins] = run;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KB; k++)
            if (bin[k] >= 0) s_body[s_bin[bin[k]] + rank[k]] = q[k];
        __syncthreads();
    }
    const float dmax = P.coll_d2_max;
    // the flag, the force record of the particles the force pass does not visit, and the list of the ones it does
    // (flag 0 and not a kid), packed at active_list[cell_start[c] ...] in whatever order the cell's waves arrive
    auto finish = [&](bool valid, int gi, bool dead, bool kid, bool met_higher, bool met_lower, bool alone = false) {
        int flag = met_higher ? 2 : met_lower ? 1 : 0;
        if (dead) flag = 2;                                          // ps.cpp:1183
        const bool on = valid && flag == 0 && !kid && !alone;       // (alone: stencil_adults)
        if (valid && !on) force4.put(P, c, gi, make_float4(0.f, 0.f, 0.f, __int_as_float(flag)));   // (the force pass writes the records of the particles it visits)
        const unsigned long long m = __ballot(on);
        if (m) {
            int off = 0;
            if (lane == 0) off = atomicAdd(&active_count[c], __popcll(m));
            off = __builtin_amdgcn_readfirstlane(off);
            if (on) active_list[base + off + __popcll(m & ((1ull << lane) - 1ull))] = gi;
        }
    };
    if (binned) {
        // thread tid's k-th body is the cell's particle tid + 256 k (the cell's own come first): position and id are
        // in registers already; only a particle that cannot collide needs its age looked up (dead or kid?)
#pragma unroll
        for (int k = 0; k < KB; k++) {
            const int first = wv * 64 + 256 * k;
            if (first >= cnt) break;
            const bool valid = lane < cnt - first;
            const int gi = base + first + (valid ? lane : 0);
            const int id_i = __float_as_int(q[k].w);
            const bool scan = valid && id_i >= 0;
            bool dead = false, kid = false, met_higher = false, met_lower = false;
            if (valid && id_i < 0) { const float age_i = snap_age[gi]; dead = age_i > P.life_thr; kid = age_i < P.kid_thr; }
            if (scan) {
                const float xi = q[k].x, yi = q[k].y, zi = q[k].z;
                const unsigned uid = (unsigned)id_i;
                const int bx = bin1(xi - ox), by = bin1(oy - yi), bz = bin1(oz - zi);
                const int x0 = max(bx - 1, 0), x1 = min(bx + 1, nb - 1);
                // the nine runs' bounds first (one batch of LDS reads), then the runs
                int j0[9], j1[9];
#pragma unroll
                for (int r = 0; r < 9; r++) {
                    const int z = bz + r / 3 - 1, y = by + r % 3 - 1;
                    const bool in = z >= 0 && z < nb && y >= 0 && y < nb;
                    const int row = (z * nb + y) * nb;
                    j0[r] = in ? s_bin[row + x0] : 0;
                    j1[r] = in ? s_bin[row + x1 + 1] : 0;
                }
                // two bodies a turn (an odd run's last body twice: the result is an OR over the hits), no branch
                // at a hit: the highest id met as a signed number and the lowest as an unsigned one say, against
                // the particle's own, whether there was one above and one below (a miss counts as id -1: neither)
                const v2f x2 = {xi, xi}, y2 = {yi, yi}, z2 = {zi, zi};
                int hi = -1;
                unsigned lo = ~0u;
#pragma unroll
                for (int r = 0; r < 9; r++)
                    for (int j = j0[r]; j < j1[r]; j += 2) {
                        const float4 qa = s_body[j], qb = s_body[min(j + 1, j1[r] - 1)];
                        const v2f rx = v2f{qa.x, qb.x} - x2, ry = v2f{qa.y, qb.y} - y2, rz = v2f{qa.z, qb.z} - z2;
                        const v2f d2 = rx * rx + ry * ry + rz * rz;
                        const int ca = !(d2.x > dmax) ? __float_as_int(qa.w) : -1, cb = !(d2.y > dmax) ? __float_as_int(qb.w) : -1;
                        hi = max(hi, max(ca, cb));
                        lo = min(lo, min((unsigned)ca, (unsigned)cb));
                    }
                met_higher = hi > id_i;
                met_lower = lo < uid;
            }
            finish(valid, gi, dead, kid, met_higher, met_lower);
        }
    } else {
        for (int first = wv * 64; first < cnt; first += 256) {
            const bool valid = lane < cnt - first;
            const int gi = base + first + (valid ? lane : 0);
            const float xi = snap_soa[gi], yi = snap_soa[cap + gi], zi = snap_soa[2 * cap + gi];
            const float age_i = snap_age[gi];
            const int id_i = sorted_id[gi];
            const bool dead = age_i > P.life_thr, kid = age_i < P.kid_thr;
            const bool scan = valid && !dead && !kid;
            unsigned long long hi_mask = 0, lo_mask = 0;
            collide_scan(P, xi, yi, zi, id_i, scan, snap_soa + base, snap_soa + cap + base, snap_soa + 2 * cap + base,
                         snap_cid + base, cnt, hi_mask, lo_mask);
            // (a particle whose own position is not a number passes the distance test against EVERY body of its
            // stencil, not only the ones near the faces: its wave walks the whole stencil)
            if (nh <= HALO_CAP && !__any(scan && !finite3(xi, yi, zi))) {
                collide_scan(P, xi, yi, zi, id_i, scan, halo_f + hat, halo_f + hplane + hat, halo_f + 2 * hplane + hat, halo_id + hat, nh,
                             hi_mask, lo_mask);
            } else {
                // the halo list overflowed (denser than the container admits in steady state): whole stencil
                for (int k = 1; k < 27; k++) {
                    const int nc = __builtin_amdgcn_readfirstlane(local_cell(P, i3 + c_stencil[k][2], i1 + c_stencil[k][1], i2 + c_stencil[k][0]));
                    if (nc < 0) continue;
                    const int nbase = __builtin_amdgcn_readfirstlane(cell_start[nc]);
                    const int n = __builtin_amdgcn_readfirstlane(min(cell_start[nc + 1] - nbase, P.max_per_cell));
                    collide_scan(P, xi, yi, zi, id_i, scan, snap_soa + nbase, snap_soa + cap + nbase, snap_soa + 2 * cap + nbase, snap_cid + nbase, n,
                                 hi_mask, lo_mask);
                }
            }
            // (a particle whose position is not a number and that has no other adult in its stencil: its sum is +0)
            bool alone = false;
            if (__any(scan && !finite3(xi, yi, zi))) {
                const int adults = stencil_adults(P, i1, i2, i3, cell_start, snap_age);   // (all lanes: the count is a wave's work)
                alone = scan && !finite3(xi, yi, zi) && adults <= 1;
            }
            finish(valid, gi, dead, kid, (hi_mask >> lane) & 1ull, (lo_mask >> lane) & 1ull, alone);
        }
    }
}

// The plan of the balanced force pass, one launch of eight workgroups (one per XCD run of wave
// slots).  Every workgroup works out, for itself, in LDS:
//   (1) the prefix over the computed cells (the lent ones first: their results travel back to the
//       rank that owns them) of the 64-slices of the active lists and of what those tasks walk
//       (a task of cell c walks task_cost[c] bodies, the population of the cell's stencil);
//       with `merge`, only full slices become ordinary tasks and the leftovers (a cell's last,
//       partly filled slice: 20 of 64 lanes on average once the collided particles are gone) are
//       packed, up to four cells to a wave, into merged tasks;
//   (2) where every wave slot of ITS run starts: the pass's work is the list of (task, stencil
//       step) units -- task-major, 27 steps per task -- a unit costs the bodies of the neighbour
//       cell it visits, and wave slot s takes the units from wave_pos[s] up to wave_pos[s + 1]:
//       equal shares of the cost, cut at unit boundaries.  The eight runs start at whole tasks, so a
//       task that is cut is always continued by a workgroup of the same run.
// The task list, the packs and the frame scalars are the same whichever workgroup writes them; each
// writes a share.  (These were three launches, k_build_active / k_active_tasks / k_split_tasks, 60 us
// of mostly one-workgroup latency on the step's critical path; the prefixes are cheap enough to
// be recomputed eight times.)
// merge: 0 every slice is an ordinary task; 1 the packs are the merged tasks of k_pairs_merged (run
// beside the balanced pass); 2 the packs are tasks of the balanced pass itself (tile walk): pack m is
// task n_tasks2 + m, with one virtual "cell" ncomp + m in the prefix arrays.
constexpr int PLAN_LDS = 6144;        // prefix entries (computed cells + virtual pack cells + 1) kept in LDS
__global__ __launch_bounds__(1024) void k_plan_force(DevParams P, int nw, int merge, const int *__restrict__ cell_start_g,
                                                     const int *__restrict__ active_count, const int *__restrict__ task_cost,
                                                     int *__restrict__ task_list2, int *__restrict__ ctask_start_g,
                                                     long long *__restrict__ cost_start_g, int4 *__restrict__ merged_tasks,
                                                     long long *__restrict__ wave_pos, FrameScalars *fs, unsigned long long *trace,
                                                     StepState *st, int pass)
{
#ifdef PSAMD_PLAN_TRACE    // diagnostic build: time stamps (100 MHz) of workgroup x's phases in trace[8 x ...]
#define PT(i) do { if (threadIdx.x == 0) trace[8 * blockIdx.x + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PT(i) do {} while (0)
#endif
    PT(0);
    __shared__ long long s_cost[PLAN_LDS + 1];
    __shared__ int s_task[PLAN_LDS + 1];
    __shared__ int s_ac[PLAN_LDS];                     // active_count | task_cost << 13 of the j-th computed cell
    __shared__ long long wave_tot[16], wave_cost[16], wave_pcost[16];
    __shared__ long long s_run[2];
    __shared__ long long s_runcost[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, x = blockIdx.x;
    const int ncomp = comp_count(P);
    const bool ext = merge == 2;
    const bool in_lds = (ext ? 2 * ncomp : ncomp) + 1 <= PLAN_LDS && P.max_per_cell < (1 << 13);
    long long *cost_start = in_lds ? s_cost : cost_start_g;
    int *ctask_start = in_lds ? s_task : ctask_start_g;
    if (in_lds) for (int j = tid; j < ncomp; j += 1024) { const int c = comp_cell(P, j), n = active_count[c]; s_ac[j] = n | ((n ? task_cost[c] : 0) << 13); }
    __syncthreads();
    PT(1);
    auto act_of = [&](int j) { return in_lds ? (s_ac[j] & 0x1fff) : active_count[comp_cell(P, j)]; };
    auto cost_of = [&](int j) { return in_lds ? (s_ac[j] >> 13) : task_cost[comp_cell(P, j)]; };

    // ---- (1) prefixes, task list, packs ----
    const int per = (ncomp + 1023) / 1024;
    const int c0 = min(ncomp, tid * per), c1 = min(ncomp, c0 + per);
    // the leftovers are packed greedily, in cell order, a run of cells per thread: six (two packs of
    // three 20-lane leftovers) where there are threads enough -- each step of the greedy walk is a
    // dependent LDS round trip, and this walk is done twice
    const int pper = max(6, (ncomp + 1023) / 1024);
    const int p0 = min(ncomp, tid * pper), p1 = min(ncomp, p0 + pper);
    // out / cost_out (may be null): the packs and, per pack, what its wave walks (its longest stencil)
    auto pack = [&](int4 *out, long long *cost_out, long long cost_base, long long *cost_sum) -> int {
        int npack = 0, used = 0, ng = 0, pc = 0;
        long long acc = 0;
        int4 cur = make_int4(-1, -1, -1, -1);
        auto flush = [&]() {
            if (out) out[npack] = cur;
            if (cost_out) cost_out[npack] = cost_base + acc;
            acc += pc; npack++;
            cur = make_int4(-1, -1, -1, -1); used = 0; ng = 0; pc = 0;
        };
        for (int j = p0; j < p1; j++) {
            const int r = act_of(j) & 63;
            if (r == 0) continue;
            const int c = comp_cell(P, j);
            if (ng == 4 || used + r > 64) flush();
            if (ng == 0) cur.x = c; else if (ng == 1) cur.y = c; else if (ng == 2) cur.z = c; else cur.w = c;
            ng++; used += r; pc = max(pc, cost_of(j));
        }
        if (ng) flush();
        if (cost_sum) *cost_sum = acc;
        return npack;
    };
    long long mine = 0, mycost = 0, mypcost = 0;   // tasks (low word) and packs (high word); bodies the tasks walk; ... the packs walk
    for (int j = c0; j < c1; j++) {
        const int n = act_of(j), nt = merge ? (n >> 6) : ((n + 63) >> 6);
        mine += nt;
        mycost += (long long)nt * cost_of(j);
    }
    if (merge) mine |= (long long)pack(nullptr, nullptr, 0, &mypcost) << 32;
    long long incl = mine, cincl = mycost, pincl = mypcost;
    for (int d = 1; d < 64; d <<= 1) {
        const long long o = __shfl_up(incl, d), oc = __shfl_up(cincl, d), op = __shfl_up(pincl, d);
        if (lane >= d) { incl += o; cincl += oc; pincl += op; }
    }
    if (lane == 63) { wave_tot[wv] = incl; wave_cost[wv] = cincl; wave_pcost[wv] = pincl; }
    __syncthreads();
    long long run2 = incl - mine, total2 = 0, crun = cincl - mycost, ctotal = 0, prun = pincl - mypcost, ptotal = 0;
    for (int k = 0; k < 16; k++) {
        if (k < wv) { run2 += wave_tot[k]; crun += wave_cost[k]; prun += wave_pcost[k]; }
        total2 += wave_tot[k]; ctotal += wave_cost[k]; ptotal += wave_pcost[k];
    }
    int run = (int)(run2 & 0xffffffffll);
    const int total = (int)(total2 & 0xffffffffll), npacks = (int)(total2 >> 32);
    PT(2);
    const bool my_share = (tid & 7) == x;           // the lists in memory: each workgroup writes an eighth
    for (int j = c0; j < c1; j++) {
        const int n = merge ? (act_of(j) >> 6) : ((act_of(j) + 63) >> 6);
        ctask_start[j] = run; cost_start[j] = crun;
        if (my_share && n) { const int c = comp_cell(P, j); for (int sl = 0; sl < n; sl++) task_list2[run + sl] = c * P.slices + sl; }
        run += n;
        crun += (long long)n * cost_of(j);
    }
    if (merge) {
        const int m0 = (int)(run2 >> 32);
        const int np = pack(x == 0 ? merged_tasks + m0 : nullptr, ext ? cost_start + ncomp + m0 : nullptr, ctotal + prun, nullptr);
        if (ext) for (int m = 0; m < np; m++) ctask_start[ncomp + m0 + m] = total + m0 + m;
    }
    const int ncells = ncomp, nent = ncomp + (ext ? npacks : 0), ntask = total + (ext ? npacks : 0);
    const long long T = ctotal + (ext ? ptotal : 0);
    if (tid == 0) {
        ctask_start[nent] = ntask;
        cost_start[nent] = T;
        if (x == 0) { fs->n_tasks2 = total; fs->n_merged = npacks; fs->cost_total = T; }
    }
    if (!in_lds) __threadfence();                    // (every workgroup wrote the same values; this one reads its own)
    __syncthreads();
    PT(3);
    if (nw <= 0) return;                             // (unbalanced pass: only the lists were wanted)

    // ---- (2) the wave slots of run x ----
    // A position in the pass's work is (task, cost already walked inside the task): which stencil step
    // that is depends on the populations of the task's stencil, which the wave that starts (or stops)
    // there looks up anyway -- k_pairs_balanced turns the residual into a step.  (Walking the 27 counts
    // here, per wave slot, was most of this kernel's 40 us.)  For a merged pack the residual IS the
    // step (its steps are taken as equally long), marked by bit 30.  whole = round up to the next task start.
    auto pos_at = [&](long long v, bool whole) -> long long {
        if (v >= T) return (long long)ntask << 32;
        int a = 0, b = nent - 1;                          // last entry whose tasks start at or before v
        while (a < b) { const int m = (a + b + 1) >> 1; if (cost_start[m] <= v) a = m; else b = m - 1; }
        const int nt = ctask_start[a + 1] - ctask_start[a];
        if (a >= ncells) {                                // a merged pack: one task
            const long long S = cost_start[a + 1] - cost_start[a], off = v - cost_start[a];
            const int k = S > 0 ? (int)min((long long)(STENCIL - 1), off * STENCIL / S) : 0;
            if (whole) return (long long)(ctask_start[a] + (off > 0 ? 1 : 0)) << 32;
            return ((long long)ctask_start[a] << 32) | (long long)(k | (1 << 30));
        }
        const int S = cost_of(a);
        if (nt == 0 || S <= 0) return (long long)ctask_start[a + 1] << 32;     // (v < T: cannot be the last cell)
        const long long off = v - cost_start[a];
        const int q = (int)min((long long)(nt - 1), off / S);
        const int r = (int)(off - (long long)q * S);
        const int t = ctask_start[a] + q;
        if (whole) return (long long)(t + (r > 0 ? 1 : 0)) << 32;
        return ((long long)t << 32) | (long long)r;
    };
    auto cost_of_task_start = [&](int t) -> long long {
        if (t >= ntask) return T;
        int a = 0, b = nent - 1;
        while (a < b) { const int mm = (a + b + 1) >> 1; if (ctask_start[mm] <= t) a = mm; else b = mm - 1; }
        if (a >= ncells) return cost_start[a];             // a merged pack is one task
        return cost_start[a] + (long long)(t - ctask_start[a]) * cost_of(a);
    };
    const int m = nw >> 3;                                // wave slots per XCD run (nw is a multiple of 32)
    if (tid < 2) {
        s_run[tid] = pos_at(T * (x + tid) / 8, true);
        s_runcost[tid] = cost_of_task_start((int)(s_run[tid] >> 32));
    }
    __syncthreads();
    PT(4);
    const long long run_lo = s_run[0], run_hi = s_run[1];
    const long long lo = s_runcost[0], hi = s_runcost[1];
    for (int j = tid; j < m; j += 1024)                   // equal shares of the run's own cost range
        wave_pos[x * m + j] = j == 0 ? run_lo : max(run_lo, min(run_hi, pos_at(lo + (hi - lo) * j / m, false)));
    if (x == 7 && tid == 0) wave_pos[nw] = run_hi;        // = (ntask, 0)
    if (x == 0 && tid == 0) {
        // the clock of the pass that follows (WavePace): how long the last one took, and when this one was planned
        const unsigned long long a = st->pairs_t0[pass], b = st->pairs_end[pass];
        st->pairs_ticks[pass] = (b > a && b - a < (1ull << 30)) ? (int)(b - a) : 0;
        st->pairs_t0[pass] = __builtin_amdgcn_s_memrealtime();
        st->pairs_end[pass] = 0;
    }
    __syncthreads();
    PT(5);
#undef PT
}

// wave_pos -> (task, stencil step) unit, by the wave that starts (or stops) there.  The stencil step of a position
// (task, cost already walked inside the task) is the number of leading stencil cells the residual covers whole:
// 27 lanes look the cells' populations up, one scan, one ballot.  Every wave of the balanced pass resolves its own two
// boundaries as its first instructions (until round 5 a launch of its own did it, k_resolve_steps: 5.5 us on the step's
// critical path for what a wave does in the shadow of its first loads).  Returns a wave-uniform number.
__device__ __forceinline__ int resolve_unit(const DevParams &P, long long pos, const int *__restrict__ cell_start, const int *__restrict__ task_list)
{
    const int lane = (int)(threadIdx.x & 63);
    const int t = __builtin_amdgcn_readfirstlane((int)(pos >> 32)), r = __builtin_amdgcn_readfirstlane((int)(pos & 0xffffffffll));
    int k = 0;
    if (r & (1 << 30)) k = r & 63;                           // a merged pack: the residual is the step
    else if (r > 0) {
        const int c = __builtin_amdgcn_readfirstlane(task_list[t]) / P.slices;
        int i1, i2, i3, cnt = 0;
        cell_coords(P, c, i1, i2, i3);
        if (lane < STENCIL) {
            const int nc = local_cell(P, i3 + c_stencil[lane][2], i1 + c_stencil[lane][1], i2 + c_stencil[lane][0]);
            if (nc >= 0) cnt = min(cell_start[nc + 1] - cell_start[nc], P.max_per_cell);
        }
        const int cum = wave_incl_scan(cnt);
        k = __popcll(__ballot(lane < STENCIL - 1 && cum <= r));
    }
    return t * STENCIL + k;
}

// One wave = 64 consecutive particles of one cell (four independent waves per workgroup).
// Neighbour cells are visited in the reference's stencil order, their bodies in list
// order, and every lane adds each body to its own particle's sum: each particle sees
// exactly the reference's sequence of fp32 additions (ps.cpp:1247-1259).
// MODE 0: exact with the compiler's correctly rounded sqrt/divide (any EPS2);
//      1: exact with the short sqrt/reciprocal above, NQ pairs per slow-branch test;
//      2: fast math (FMA + v_rsq), not bit-exact.
// SHARDED: the launch covers only this rank's run of the task list.
//
// Modes 1 and 2 never stage neighbour data at all.  It is the same for all 64 lanes, the
// ranges are wave-uniform, so the loads are scalar loads (s_load_dwordx8 from the SoA
// snapshot, straight out of L2 into SGPRs) and the packed fp32 instructions take the SGPR
// pairs as operands: no LDS, no vector registers for the bodies.  (An LDS tile read with
// ds_read_b128 by four waves per CU kept the LDS pipe ~70 % busy -- 16 cycles per wave
// read, scripts/microbench/lds_groups.hip -- and cost 4 % more time.)
// (Round 4 tried a third way -- every row of 16 lanes holds 16 bodies in VGPRs and the arithmetic takes them through
// DPP, `v_sub_f32_dpp rx, tile_x, xi row_newbcast:j`: no LDS, no scalar loads, the compiler fuses every broadcast.
// Bit-identical and 9-17 % slower everywhere: a DPP-modified v_sub / v_mul issues at half rate on gfx950.
// profiles/r4_ab_dpp_walk.txt, commit 2592be9.)
// Mode 0, the fallback for softening lengths outside the lean range, streams 64-body
// tiles through 1 KiB of LDS per wave.  No s_barrier: a wave only ever touches its own
// tile, and a wave's LDS operations complete in issue order, so a compiler-level fence
// is all the ordering needed.
#ifndef PSAMD_BALANCED_WAVES
#define PSAMD_BALANCED_WAVES 7      // resident waves per SIMD the scalar-walk force pass is built for (70 VGPRs; measured, exact / tolerance arithmetic: 6 waves 2.15 / 1.25 ms, 7 waves 2.11 / 1.22 ms)
#endif

#ifdef PSAMD_WAVE_TRACE   // diagnostic build only: when and where did this wave run
#define PS_TRACE_BEGIN() const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime()
#define PS_TRACE_END() do { if ((threadIdx.x & 63) == 0) { \
        unsigned long long *t_ = trace + (size_t)3 * (blockIdx.x * 4 + (threadIdx.x >> 6)); \
        t_[0] = trace_t0; t_[1] = __builtin_amdgcn_s_memrealtime(); \
        t_[2] = ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 0xf) << 32)   /* XCC_ID */ \
                | __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4); } } while (0)               /* HW_ID */
#else
#define PS_TRACE_BEGIN() do {} while (0)
#define PS_TRACE_END() do {} while (0)
#endif
#ifndef TILE_PIPELINED
#define TILE_PIPELINED 1            // the tile walk reads a group of bodies from LDS while it works through the one before (A/B builds: 0)
#endif

// One task: 64 consecutive particles of one cell against the cell's stencil.
// Hand-off of a task's partial sums between the wave that walked the first stencil steps and
// the one that continues (balanced force pass).  Follows the guide's inter-workgroup recipe
// (cdna_hip_programming.md, Guideline 16): the payload is stored write-through with agent-scope
// atomic stores, the storing wave drains its stores, ONE lane raises the flag with an agent-scope
// atomic store; the consumer polls that one word relaxed and reads the payload with agent-scope
// atomic loads (they bypass the CU's L1, so no acquire fence is needed).  The flags are zeroed
// with the frame, before the launch.
typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

// The flag word carries the stencil step the published sums stand at, so that a task cut in
// three or more pieces hands on correctly at every cut (each consumer waits for ITS step).
__device__ __forceinline__ void handoff_publish(float4 *slot, float ax, float ay, float az, int flag, bool valid, int *ready, int step)
{
    if (valid) {
        gu64 *p = (gu64 *)(unsigned long long *)slot;
        __hip_atomic_store(p, ((unsigned long long)__float_as_uint(ay) << 32) | __float_as_uint(ax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, ((unsigned long long)(unsigned)flag << 32) | __float_as_uint(az), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((threadIdx.x & 63) == 0) __hip_atomic_store((gu32 *)(unsigned int *)ready, (unsigned)step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// false: the flag never came (bounded spin; the caller raises a sticky error)
__device__ __forceinline__ bool handoff_consume(const float4 *slot, float &ax, float &ay, float &az, int &flag, bool valid, const int *ready, int step)
{
    int ok = 0;
    if ((threadIdx.x & 63) == 0) {
        for (unsigned spins = 0; spins < (1u << 22); spins++) {
            if (__hip_atomic_load((gu32 *)(unsigned int *)ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)step) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    ok = __builtin_amdgcn_readfirstlane(ok);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       // compiler-only: the loads below stay below the poll
    if (ok && valid) {
        gu64 *p = (gu64 *)(unsigned long long *)slot;
        const unsigned long long a = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ax = __uint_as_float((unsigned)a); ay = __uint_as_float((unsigned)(a >> 32));
        az = __uint_as_float((unsigned)b); flag = (int)(unsigned)(b >> 32);
    }
    return ok != 0;
}

// Pacing of the balanced force pass.  All its waves are resident and have the same amount of work, but the SIMD issues
// oldest-first: the seven waves of a SIMD do not advance together, they END one after the other (wave trace, round 4:
// the workgroups dispatched first end at 37 % of the kernel's span, the next at 47 %, ... the last at 92-100 %), and
// for the last 40 % of the launch a SIMD holds fewer than four waves -- at the end a lone one, which cannot cover its
// scalar-load round trips (15 % of the kernel's issue slots idle).  So every wave keeps itself on schedule: at each
// stencil step it compares the share of its work it has done with the share of the pass's expected duration that has
// gone by (the duration of the last such pass, kept in StepState by the planning kernel and the waves themselves) and
// sets its issue priority accordingly -- behind schedule: up, ahead: down.  Waves then advance together and end
// together, whatever their age.  Nothing but issue order changes: same instructions, same results.
struct WavePace {
    unsigned long long t0 = 0;      // when the pass was planned (100 MHz counter)
    float per_tick = 0.f;           // 1 / expected duration of the pass, in ticks; 0: no pacing (no history yet)
    float per_unit = 0.f;           // 1 / the wave's (task, stencil step) units
    int done = 0;                   // units done so far
    int band = 20;                  // how far off schedule (1/1024 of the pass) before the priority goes to an end of its range
    __device__ __forceinline__ void step()
    {
        done++;
        if (per_tick == 0.f) return;
        const float lag = (float)(long long)(__builtin_amdgcn_s_memrealtime() - t0) * per_tick - (float)done * per_unit;
        // (s_setprio is a scalar instruction: it executes whatever EXEC says, so the choice must be a scalar branch --
        // the lag as a wave-uniform integer, in 1/1024 of the pass)
        const int q = __builtin_amdgcn_readfirstlane((int)(lag * 1024.0f));
        if (q > band) __builtin_amdgcn_s_setprio(3);
        else if (q > 0) __builtin_amdgcn_s_setprio(2);
        else if (q > -band) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
};

// Stencil steps [k0, k1) of a task.  resume: the sums of steps < k0 come from the wave that
// walked them (ready != nullptr); a walk that stops before step 27 publishes its sums instead
// of finishing the particle.  The whole task is k0 = 0, k1 = 27, ready = nullptr.
// SETTLED: the collision flags are known already (two-pass mode: the balanced pass) -- nothing tracks distances for them
template <int MODE, int NQ, bool SETTLED = false>
__device__ __forceinline__ void pairs_task(const DevParams &P, const int *__restrict__ cell_start,
                                           const SnapSoa snap4, const float *__restrict__ snap_soa,
                                           const float *__restrict__ snap_age, const int *__restrict__ sorted_id,
                                           const ForceBuf force4, int task,
                                           float4 *tile, unsigned long long *trace,
                                           const int *__restrict__ active_list = nullptr,
                                           const int *__restrict__ active_count = nullptr,
                                           int k0 = 0, int k1 = STENCIL, int *ready = nullptr, FrameScalars *fs = nullptr,
                                           WavePace *pace = nullptr)
{
    PS_TRACE_BEGIN();
    const int c = task / P.slices, slice = task - c * P.slices;
    const int base = cell_start[c];
    // two-pass mode: the slice is cut from the cell's list of particles that need a force
    const int cnt = active_list ? active_count[c] : min(cell_start[c + 1] - base, P.max_per_cell);
    const int first = slice * 64;
    if (first >= cnt) return;
    const int nvalid = min(64, cnt - first);
    const int lane = threadIdx.x & 63;
    const bool valid = lane < nvalid;
    const int gi = active_list ? active_list[base + first + (valid ? lane : 0)] : base + first + (valid ? lane : 0);
    const float4 me = snap4[gi];
    const float age_i = snap_age[gi];
    const int id_i = sorted_id[gi];
    const bool dead = age_i > P.life_thr;                      // ps.cpp:1183
    const bool kid = age_i < P.kid_thr;
    const bool scan = SETTLED ? false : (valid && !dead && !kid && !active_list);   // two-pass mode: flags are settled already

    int i1, i2, i3;
    cell_coords(P, c, i1, i2, i3);
    float ax = 0.f, ay = 0.f, az = 0.f;
    int flag = 0;
    const float eps2f = (float)P.eps2;

    // Lane k (< 27) looks up neighbour cell k of the stencil once: its range in the
    // sorted order, or an empty range if it lies outside the grid.
    int my_nb = 0, my_cnt = 0;
    if (lane < 27) {
        const int nc = local_cell(P, i3 + c_stencil[lane][2], i1 + c_stencil[lane][1], i2 + c_stencil[lane][0]);
        if (nc >= 0) {
            my_nb = cell_start[nc];
            my_cnt = min(cell_start[nc + 1] - my_nb, P.max_per_cell);
        }
    }
    if (MODE != 0) {
        const PairCtx ctx = {me.x, me.y, me.z, age_i, id_i, gi, scan};
        const size_t cap = (size_t)P.sorted_cap;
        if (k0 > 0 && !handoff_consume(force4 + gi, ax, ay, az, flag, valid, ready, k0)) {
            if (lane == 0) atomicOr(&fs->error, ERR_HANDOFF_TIMEOUT);
        }
        // The bodies [nb, nb + n) of one cell, from four planes of a snapshot (wave-uniform pointers: scalar loads).
        auto walk_cell = [&](const float *__restrict__ sx, const float *__restrict__ sy, const float *__restrict__ sz,
                             const float *__restrict__ sw, int nb, int n) {
            float dmin = 3.0e38f;
            int jj = 0;
            // NQ bodies per group.  (Fetching the next group between the distance stage and
            // the rest -- scalar loads return out of order, so it cannot go out any earlier --
            // was measured 3 % slower for the exact arithmetic on a full GPU and no faster
            // on a 1/8 share.)
            for (; jj + NQ <= n; jj += NQ) {
                v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], qw[NQ / 2];
#pragma unroll
                for (int i = 0; i < NQ / 2; i++) {
                    qx[i] = v2f{sx[jj + 2 * i], sx[jj + 2 * i + 1]};
                    qy[i] = v2f{sy[jj + 2 * i], sy[jj + 2 * i + 1]};
                    qz[i] = v2f{sz[jj + 2 * i], sz[jj + 2 * i + 1]};
                    qw[i] = v2f{sw[jj + 2 * i], sw[jj + 2 * i + 1]};
                }
                // (The compiler lets the masses' load sink to its use, behind the reciprocal square roots; pinned up
                // here with the coordinates' loads -- four in one batch -- the pass took the same time, 2.13 ms.)
                if (MODE == 1)
                    pairsN_exact_lean<NQ>(P, ctx, qx, qy, qz, qw, nb + jj, snap_age, sorted_id, ax, ay, az, flag);
                else
                    dmin = fminf(dmin, pairsN_fast<NQ>(ctx, qx, qy, qz, qw, eps2f, ax, ay, az));
            }
            for (; jj < n; jj++) {
                const float4 q = make_float4(sx[jj], sy[jj], sz[jj], sw[jj]);
                if (MODE == 1)
                    pair1_exact_lean(P, ctx, q, nb + jj, snap_age, sorted_id, ax, ay, az, flag);
                else
                    dmin = fminf(dmin, pair_fast(me.x, me.y, me.z, q, eps2f, ax, ay, az) + eps2f);
            }
            // fast math, rare: someone in this cell is within the (widened) collision gate of
            // one of my lanes; the exact rule is then evaluated on unfused distances
            const float gate_soft = (P.coll_d2_gate + eps2f) * 1.0001f;
            if (MODE == 2 && __any(scan && !(dmin > gate_soft))) {
                if (scan && !(dmin > gate_soft)) {
                    for (int j = 0; j < n; j++) {
                        const float rx = sx[j] - me.x, ry = sy[j] - me.y, rz = sz[j] - me.z;
                        const float d2 = rx * rx + ry * ry + rz * rz;
                        if (!(d2 > P.coll_d2_gate) && nb + j != gi)
                            flag = max(flag, collide_exact(P, d2, age_i, id_i, snap_age[nb + j], sorted_id[nb + j]));
                    }
                }
            }
        };
        // the stencil, in the reference's order
        for (int k = k0; k < k1; k++) {
            const int nb = __builtin_amdgcn_readlane(my_nb, k), n = __builtin_amdgcn_readlane(my_cnt, k);
            const float *sx = snap_soa + nb;
            walk_cell(sx, sx + cap, sx + 2 * cap, sx + 3 * cap, nb, n);
            if (pace) pace->step();
        }
    } else {
        // Generic exact mode: tiles of 64 snapshot entries, in stencil order then list order.
        // The next tile's global load is issued before the current tile is consumed.  The lean
        // modes let a particle meet itself (r = 0 adds +0, exactly nothing) because
        // 1/sqrt(eps2^3) is finite on the range they are allowed on; this one also serves
        // softening lengths where it is not, so it skips the self pair explicitly, as the
        // reference does by id (ps.cpp:1258), and a kid neighbour too (app_common.cu:240: ai
        // comes back unchanged; its zeroed mass times an infinite 1/r^3 would be a NaN).
        int k = 0, t0 = 0;
        int nb = __shfl(my_nb, 0), ncnt = __shfl(my_cnt, 0);
        while (ncnt == 0 && ++k < 27) { nb = __shfl(my_nb, k); ncnt = __shfl(my_cnt, k); }
        bool have = k < 27;
        float4 pre = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have && lane < min(64, ncnt)) pre = snap4[nb + lane];
        while (have) {
            const int c_nb = nb, c_t0 = t0, n = min(64, ncnt - t0);
            PS_WAVE_SYNC();                           // previous tile fully consumed
            if (lane < n) tile[lane] = pre;
            PS_WAVE_SYNC();
            t0 += 64;                                 // advance to the next non-empty tile
            if (t0 >= ncnt) {
                t0 = 0; ncnt = 0;
                while (ncnt == 0 && ++k < 27) { nb = __shfl(my_nb, k); ncnt = __shfl(my_cnt, k); }
            }
            have = k < 27;
            // issued after the fences (they drain outstanding loads), consumed a tile later
            if (have && lane < min(64, ncnt - t0)) pre = snap4[nb + t0 + lane];
            float dmin = 3.0e38f, dsum = 0.0f;          // (dsum: a distance that is not a number passes the collision test; fminf drops it)
#pragma unroll 4
            for (int jj = 0; jj < n; jj++) {
                if (c_nb + c_t0 + jj == gi) continue;
                const float4 q = tile[jj];
                if (q.w == 0.0f) {                     // kid (or massless) neighbour: no force term, still a distance
                    const float rx = q.x - me.x, ry = q.y - me.y, rz = q.z - me.z;
                    const float d2 = rx * rx + ry * ry + rz * rz;
                    dmin = fminf(dmin, d2); dsum += d2;
                    continue;
                }
                const float d2 = pair_exact(me.x, me.y, me.z, q, P.eps2, ax, ay, az);
                dmin = fminf(dmin, d2); dsum += d2;
            }
            // rare: someone in this tile is within the collision gate of one of my lanes
            const bool close = scan && (!(dmin > P.coll_d2_gate) || dsum != dsum);
            if (__any(close)) {
                if (close) {
                    for (int jj = 0; jj < n; jj++) {
                        const float4 q = tile[jj];
                        const float rx = q.x - me.x, ry = q.y - me.y, rz = q.z - me.z;
                        const float d2 = rx * rx + ry * ry + rz * rz;
                        const int gj = c_nb + c_t0 + jj;
                        if (!(d2 > P.coll_d2_gate) && gj != gi)
                            flag = max(flag, collide_exact(P, d2, age_i, id_i, snap_age[gj], sorted_id[gj]));
                    }
                }
            }
        }
    }
    if (MODE != 0 && k1 < STENCIL) {             // not the end of the walk: hand the sums on
        handoff_publish(force4 + gi, ax, ay, az, flag, valid, ready, k1);
        PS_TRACE_END();
        return;
    }
    if (dead) flag = 2;
    if (kid) { ax = 0.f; ay = 0.f; az = 0.f; }   // every term is skipped for a kid (app_common.cu:240)
    if (MODE != 0 && !SETTLED) {
        // one-pass lean stage: a particle whose own position is not a number met itself and the kids (stencil_adults;
        // the two-pass stage settles this in k_collide_cell, the generic mode skips both explicitly)
        const bool lost = valid && !kid && !finite3(me.x, me.y, me.z);
        if (__any(lost)) {
            const int adults = stencil_adults(P, i1, i2, i3, cell_start, snap_age);       // (all lanes: the count is a wave's work)
            if (lost && adults <= 1) { ax = 0.f; ay = 0.f; az = 0.f; }
        }
    }
    if (valid) force4.put_id(P, c, gi, id_i, make_float4(ax, ay, az, __int_as_float(flag)));
    PS_TRACE_END();
}

template <int MODE, int NQ>
__global__ __launch_bounds__(256) void k_pairs(DevParams P, const int *__restrict__ cell_start,
                                               const SnapSoa snap4,
                                               const float *__restrict__ snap_soa,
                                               const float *__restrict__ snap_age,
                                               const int *__restrict__ sorted_id,
                                               const int *__restrict__ task_list,
                                               const ForceBuf force4,
                                               FrameScalars *fs, unsigned long long *trace,
                                               const int *__restrict__ active_list, const int *__restrict__ active_count)
{
    // Workgroups of four INDEPENDENT waves (no workgroup barrier anywhere): the hardware
    // deals a workgroup's waves over the four SIMDs of its CU and workgroups over the
    // CUs, which keeps even a small share (a few waves per CU) evenly spread.
    __shared__ float4 tiles[MODE == 0 ? 4 : 1][MODE == 0 ? 64 : 1];   // mode 0 only
    const int wave = threadIdx.x >> 6;
    // The work list holds only non-empty (cell, slice) tasks, cell-major.  Workgroups are dealt
    // round-robin over the eight XCDs (b and b + 8 share an L2), so workgroup b takes its four
    // tasks from XCD (b & 7)'s contiguous eighth of the list: neighbouring cells' snapshots then
    // sit in that XCD's L2.
    // (Eighths of equal WORK instead of equal length -- the outer planes of the grid have
    // fewer neighbours, so the two XCDs holding them go idle for the last sixth of the
    // launch -- were tried: the XCDs then finish together, yet the launch was only 1 %
    // shorter and the extra prefix sum cost k_scan 10 us.)
    const int ntask = active_list ? fs->n_tasks2 : fs->n_tasks;
    const int nwg = (ntask + 3) >> 2;
    if ((int)blockIdx.x >= nwg) return;
    const int slot = xcd_contiguous(blockIdx.x, nwg) * 4 + wave;
    if (slot >= ntask) return;
    pairs_task<MODE, NQ>(P, cell_start, snap4, snap_soa, snap_age, sorted_id, force4,
                         task_list[slot], tiles[MODE == 0 ? wave : 0], trace, active_list, active_count, 0, STENCIL, nullptr, fs);
}

// ------------------------------------------------------------------ all-pairs forces (PSAMD_FLAG_ALL_PAIRS, not in the reference)
// A particle's acceleration = the stencil's chain, exactly the cutoff pass above (the reference's order), plus every
// other cell of the box in GLOBAL index order.  The far field is 99 % of the work and the same for every particle but
// for the 27 cells it must leave out, so it does not go by (cell, slice) tasks -- whose last slices are mostly empty
// lanes: a quarter of all lanes at 64 particles per cell -- but by DENSE tasks: the particles that need a force, in
// cell order, 64 to a wave whatever their cells.  A wave (dense task, part) walks the cells of its part (a sixteenth
// of the box, by blocks of 64 cells) in chunks of ALLP_CHUNK consecutive cells: a chunk's bodies are ONE chain of
// additions started at +0 (an fp32 sum of a quarter of a million terms in one chain would carry 4e-5 of rounding,
// measured at N = 2^18), the chunk's sum is added to the part's, k_allpairs_combine adds the parts to the stencil's
// chain in part order.  A lane whose own stencil holds a cell of the chunk leaves that cell's bodies out (its sums are
// put back after the cell's walk); a chunk no lane has in its stencil, its cells adjacent in the buffer -- nearly
// all -- is walked in one go, with one ragged tail per chunk instead of one per cell.  The association depends on
// nothing but the global cell order: the same bits on one GPU and on any number of ranks, where far_buf is the
// all-gathered snapshot of all ranks with its index by global cell (k_allg_index) instead of the own snapshot.
#ifndef PSAMD_ALLP_CHUNK
#define PSAMD_ALLP_CHUNK 4
#endif
constexpr int ALLP_CHUNK = PSAMD_ALLP_CHUNK;          // (divides 64)

// act_start[j]: how many particles need a force in the pass's cells before its j-th; [comp_count]: in all.  One workgroup.
__global__ __launch_bounds__(1024) void k_allp_prefix(DevParams P, const int *__restrict__ active_count, int *__restrict__ act_start)
{
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ncomp = comp_count(P), per = (ncomp + 1023) / 1024;
    const int c0 = min(ncomp, tid * per), c1 = min(ncomp, c0 + per);
    int mine = 0;
    for (int j = c0; j < c1; j++) mine += active_count[comp_cell(P, j)];
    const int incl = wave_incl_scan(mine);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    int run = incl - mine, total = 0;
    for (int k = 0; k < 16; k++) { if (k < wv) run += wave_tot[k]; total += wave_tot[k]; }
    for (int j = c0; j < c1; j++) { act_start[j] = run; run += active_count[comp_cell(P, j)]; }
    if (tid == 0) act_start[ncomp] = total;
}

// the dense order: sorted index and cell of the r-th particle that needs a force.  One wave per cell of the pass.
__global__ __launch_bounds__(256) void k_allp_dense(DevParams P, const int *__restrict__ cell_start, const int *__restrict__ active_list,
                                                    const int *__restrict__ active_count, const int *__restrict__ act_start,
                                                    int *__restrict__ dense_gi, int *__restrict__ dense_cell)
{
    const int j = blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= comp_count(P)) return;
    const int c = comp_cell(P, j), base = cell_start[c], n = active_count[c], o = act_start[j];
    for (int i = lane; i < n; i += 64) { dense_gi[o + i] = active_list[base + i]; dense_cell[o + i] = c; }
}

// n bodies from four planes of a snapshot (wave-uniform pointers: scalar loads), added to (ax, ay, az) in list order
template <int MODE, int NQ>
__device__ __forceinline__ void walk_far(const DevParams &P, const PairCtx &ctx, const float *__restrict__ sx, const float *__restrict__ sy,
                                         const float *__restrict__ sz, const float *__restrict__ sw, int n, float eps2f,
                                         float &ax, float &ay, float &az)
{
    int flag = 0, jj = 0;
    for (; jj + NQ <= n; jj += NQ) {
        v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], qw[NQ / 2];
#pragma unroll
        for (int i = 0; i < NQ / 2; i++) {
            qx[i] = v2f{sx[jj + 2 * i], sx[jj + 2 * i + 1]};
            qy[i] = v2f{sy[jj + 2 * i], sy[jj + 2 * i + 1]};
            qz[i] = v2f{sz[jj + 2 * i], sz[jj + 2 * i + 1]};
            qw[i] = v2f{sw[jj + 2 * i], sw[jj + 2 * i + 1]};
        }
        if (MODE == 1) pairsN_exact_lean<NQ>(P, ctx, qx, qy, qz, qw, 0, nullptr, nullptr, ax, ay, az, flag);
        else (void)pairsN_fast<NQ>(ctx, qx, qy, qz, qw, eps2f, ax, ay, az);
    }
    for (; jj < n; jj++) {
        const float4 q = make_float4(sx[jj], sy[jj], sz[jj], sw[jj]);
        if (MODE == 1) pair1_exact_lean(P, ctx, q, 0, nullptr, nullptr, ax, ay, az, flag);
        else (void)pair_fast(ctx.xi, ctx.yi, ctx.zi, q, eps2f, ax, ay, az);
    }
}

template <int MODE, int NQ>
__global__ __launch_bounds__(256, PSAMD_BALANCED_WAVES) void k_allp_far(DevParams P, const SnapSoa snap4, const int *__restrict__ act_start,
                                                                        const int *__restrict__ dense_gi, const int *__restrict__ dense_cell,
                                                                        const FarCells far, const float *__restrict__ far_buf,
                                                                        const int *__restrict__ far_start, const int *__restrict__ far_n)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_act = act_start[comp_count(P)];
    const int ntask = min((n_act + 63) >> 6, (int)(far.part_plane >> 6));      // (the partial sums' room: never short, see capi.hip)
    const int nitem = ntask * ALLP_PARTS, nwg = (nitem + 3) >> 2;
    // (the launch is sized from the host's bound of the live count, the items from the device's own count: a launch
    // that is too small for them -- it should not be -- takes several rounds instead of leaving particles out)
    for (int b = blockIdx.x; b < nwg; b += gridDim.x) {
    // part-major: an XCD's contiguous eighth of the items is two parts -- an eighth of the far bodies, which then sit in its L2
    const int slot = xcd_contiguous(b, nwg) * 4 + wave;
    if (slot >= nitem) continue;
    const int part = slot / ntask, T = slot - part * ntask;
    const int r = T * 64 + lane;
    const bool valid = r < n_act;
    const int rr = valid ? r : T * 64;                      // (a lane past the end rides along on the task's first particle; nothing of it is stored)
    const int gi = dense_gi[rr], c = dense_cell[rr];
    const float4 me = snap4[gi];
    int i1, i2, i3;
    cell_coords(P, c, i1, i2, i3);
    const PairCtx ctx = {me.x, me.y, me.z, 0.f, 0, gi, false};
    const float eps2f = (float)P.eps2;
    const size_t plane = (size_t)far.plane;
    const int nblk = (P.num_cells_global + 63) >> 6, GG = P.G * P.G;
    const int blk_lo = nblk * part / ALLP_PARTS, blk_hi = nblk * (part + 1) / ALLP_PARTS;
    float px = 0.f, py = 0.f, pz = 0.f;                     // the part's sum
    for (int blk = blk_lo; blk < blk_hi; blk++) {
        // the block's 64 cell ranges in one vector load, lane = cell (a scalar load per cell, and the body loads
        // behind it, were two dependent round trips for 64 bodies of work); with each cell's grid coordinates
        const int c2 = blk * 64 + lane;
        int f_nb = 0, f_cnt = 0, f_j = 0;
        if (c2 < P.num_cells_global) {
            const int j3 = c2 / GG, rem = c2 - j3 * GG, j1 = rem / P.G, j2 = rem - j1 * P.G;
            f_nb = far_start[c2];
            f_cnt = far_n ? far_n[c2] : min(far_start[c2 + 1] - f_nb, P.max_per_cell);
            f_j = (j3 << 20) | (j1 << 10) | j2;
        }
        for (int q0 = 0; q0 < 64; q0 += ALLP_CHUNK) {
            int n[ALLP_CHUNK], nb[ALLP_CHUNK];
            int total = 0, first = 0;
            bool adjacent = true, hit[ALLP_CHUNK], any_hit = false;
#pragma unroll
            for (int q = 0; q < ALLP_CHUNK; q++) {
                n[q] = __builtin_amdgcn_readlane(f_cnt, q0 + q);
                nb[q] = __builtin_amdgcn_readlane(f_nb, q0 + q);
                const int j = __builtin_amdgcn_readlane(f_j, q0 + q);
                if (n[q] > 0) {
                    if (total == 0) first = nb[q]; else adjacent = adjacent && nb[q] == first + total;
                    total += n[q];
                }
                hit[q] = n[q] > 0 && abs((j >> 20) - i3) <= 1 && abs(((j >> 10) & 1023) - i1) <= 1 && abs((j & 1023) - i2) <= 1;
                any_hit |= hit[q];
            }
            if (total == 0) continue;
            float ax = 0.f, ay = 0.f, az = 0.f;
            if (adjacent && !__any(any_hit)) {
                const float *sx = far_buf + first;
                walk_far<MODE, NQ>(P, ctx, sx, sx + plane, sx + 2 * plane, sx + 3 * plane, total, eps2f, ax, ay, az);
            } else {
#pragma unroll
                for (int q = 0; q < ALLP_CHUNK; q++) {
                    if (n[q] == 0) continue;
                    const float kx = ax, ky = ay, kz = az;
                    const float *sx = far_buf + nb[q];
                    walk_far<MODE, NQ>(P, ctx, sx, sx + plane, sx + 2 * plane, sx + 3 * plane, n[q], eps2f, ax, ay, az);
                    if (hit[q]) { ax = kx; ay = ky; az = kz; }      // a cell of this lane's own stencil: the cutoff pass has it
                }
            }
            px += ax; py += ay; pz += az;
        }
    }
    if (valid) far.part_acc[(size_t)part * far.part_plane + (size_t)r] = make_float4(px, py, pz, 0.f);
    }
}

// All-pairs: a particle's acceleration = (((stencil chain + part 0) + part 1) + ...) + part 15, the same
// association on one GPU and on any number of ranks.  One thread per particle that needs a force, in the dense order.
__global__ void k_allpairs_combine(DevParams P, const int *__restrict__ act_start, const int *__restrict__ dense_gi,
                                   const int *__restrict__ dense_cell, const FarCells far, const ForceBuf force4)
{
    const int n = min(act_start[comp_count(P)], (int)far.part_plane);
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < n; r += gridDim.x * blockDim.x) {
        const int gi = dense_gi[r], lc = dense_cell[r];
        float4 a = force4.get(P, lc, gi);                   // (flag 0, not a kid: it is on the active list)
#pragma unroll
        for (int p = 0; p < ALLP_PARTS; p++) {
            const float4 b = far.part_acc[(size_t)p * far.part_plane + (size_t)r];
            a.x += b.x; a.y += b.y; a.z += b.z;
        }
        force4.put(P, lc, gi, a);
    }
}

constexpr int MERGE_TILE = 4 * 64 + 4;          // floats per lane group: x[64] y[64] z[64] w[64] + skew

// The same walk for a wave that has its SIMD (almost) to itself -- a slab of a multi-GPU run has
// about 1.5 force tasks per SIMD.  There the scalar-load walk of pairs_task is latency-bound (one
// wave cannot cover its own s_load round trips: 1.6x slower per task, PSAMD_WAVES sweep in
// DESIGN.md), so the bodies come as 64-body tiles instead: one vector load per lane, issued a
// whole tile ahead (vector loads retire in order, so they pipeline), through LDS (SoA, no
// barrier: a wave reads only its own tiles and its LDS operations complete in order), read back
// as broadcast 16-byte rows.  Same arithmetic, same order: short last tiles are padded with
// massless bodies far outside the box (r * 0 = +-0 added to a sum that started at +0 changes
// nothing, as for kids).  Two-pass mode only (flags are settled), lean arithmetic.
//
// A wave serves up to four lane GROUPS, each a run of one cell's particles with its own stencil
// and its own tile (the groups' tiles skewed by 16 bytes onto different banks): one group of up
// to 64 lanes = an ordinary (cell, slice) task; several = the partly filled last slices of up to
// four cells packed into one wave (a cell's list of ~148 particles fills two slices and a third
// of another).  All groups walk stencil step k together, tile by tile, for as many rows as the
// longest of their lists.
struct TileGroups {
    int ng;
    int cell[4], first[4], count[4];      // group g: particles active_list[cell_start[cell] + first ..][0 .. count)
};

// NG: how many groups the code is built for (1: an ordinary task, nothing per-group left in it; 4: a pack)
template <int MODE, int NQ, int NG, bool ONE_T>
__device__ __forceinline__ void pairs_task_tile(const DevParams &P, const int *__restrict__ cell_start,
                                                const SnapSoa snap4, const ForceBuf force4,
                                                const TileGroups &G, float *tile, const int *__restrict__ active_list,
                                                int k0, int k1, int *ready, FrameScalars *fs)
{
    const int lane = threadIdx.x & 63;
    int off[5] = {0, 0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 4; g++) off[g + 1] = off[g] + ((g < NG && g < G.ng) ? G.count[g] : 0);
    const int g = NG == 1 ? 0 : (lane >= off[1]) + (lane >= off[2]) + (lane >= off[3]);       // a lane past the last group: 3, invalid
    const bool valid = lane < off[4];
    const int gc = valid ? (g == 0 ? G.cell[0] : g == 1 ? G.cell[1] : g == 2 ? G.cell[2] : G.cell[3]) : G.cell[0];
    const int gf = valid ? (g == 0 ? G.first[0] : g == 1 ? G.first[1] : g == 2 ? G.first[2] : G.first[3]) : G.first[0];
    const int l = valid ? lane - (g == 0 ? off[0] : g == 1 ? off[1] : g == 2 ? off[2] : off[3]) : 0;
    const int gi = active_list[cell_start[gc] + gf + l];
    const float4 me = snap4[gi];
    const float eps2f = (float)P.eps2;
    // neighbour ranges of all groups: entry e = group * 27 + stencil step, held by lane e % 64
    int tab_nb[2] = {0, 0}, tab_cnt[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < (NG == 1 ? 1 : 2); r++) {
        const int e = lane + 64 * r, eg = e / STENCIL, ek = e - eg * STENCIL;
        const int ec = (eg < NG && eg < G.ng) ? (eg == 0 ? G.cell[0] : eg == 1 ? G.cell[1] : eg == 2 ? G.cell[2] : G.cell[3]) : -1;
        if (ec >= 0) {
            int i1, i2, i3;
            cell_coords(P, ec, i1, i2, i3);
            const int nc = local_cell(P, i3 + c_stencil[ek][2], i1 + c_stencil[ek][1], i2 + c_stencil[ek][0]);
            if (nc >= 0) {
                tab_nb[r] = cell_start[nc];
                tab_cnt[r] = min(cell_start[nc + 1] - tab_nb[r], P.max_per_cell);
            }
        }
    }
    float ax = 0.f, ay = 0.f, az = 0.f;
    int flag = 0;
    const PairCtx ctx = {me.x, me.y, me.z, 0.f, 0, gi, false};
    if (k0 > 0 && !handoff_consume(force4 + gi, ax, ay, az, flag, valid, ready, k0)) {
        if (lane == 0) atomicOr(&fs->error, ERR_HANDOFF_TIMEOUT);
    }
    const float far = 1.0e6f;                                       // padding body, mass 0
    const float *tx = tile + (valid ? g : 0) * MERGE_TILE, *ty = tx + 64, *tz = tx + 128, *tw = tx + 192;
    int nbs[4] = {0, 0, 0, 0}, cnts[4] = {0, 0, 0, 0};
    // ranges of stencil step k for every group; returns the longest list
    auto step_ranges = [&](int k) -> int {
        int longest = 0;
#pragma unroll
        for (int gg = 0; gg < NG; gg++) {
            const int e = gg * STENCIL + k;
            nbs[gg] = __builtin_amdgcn_readlane(e < 64 ? tab_nb[0] : tab_nb[1], e & 63);
            cnts[gg] = gg < G.ng ? __builtin_amdgcn_readlane(e < 64 ? tab_cnt[0] : tab_cnt[1], e & 63) : 0;
            longest = max(longest, cnts[gg]);
        }
        return longest;
    };
    float4 pre[NG];
    auto fetch = [&](int t0) {                                      // this lane's body of every group's tile at row t0
#pragma unroll
        for (int gg = 0; gg < NG; gg++) {
            pre[gg] = make_float4(far, far, far, 0.f);
            if (gg < G.ng && lane < cnts[gg] - t0) pre[gg] = snap4[nbs[gg] + t0 + lane];
        }
    };
    // first non-empty step from k0 on, its first tiles fetched ahead
    int k = k0, t0 = 0, longest = 0;
    while (k < k1 && (longest = step_ranges(k)) == 0) k++;
    bool have = k < k1;
    if (have) fetch(0);
    while (have) {
        const int n = (min(64, longest - t0) + NQ - 1) & ~(NQ - 1);
        PS_WAVE_SYNC();                               // previous tiles fully consumed
#pragma unroll
        for (int gg = 0; gg < NG; gg++)
            if (gg < G.ng) {
                float *t = tile + gg * MERGE_TILE + lane;
                t[0] = pre[gg].x; t[64] = pre[gg].y; t[128] = pre[gg].z; t[192] = pre[gg].w;
            }
        PS_WAVE_SYNC();
        t0 += 64;                                     // advance to the next non-empty row of tiles
        if (t0 >= longest) {
            t0 = 0; longest = 0; k++;
            while (k < k1 && (longest = step_ranges(k)) == 0) k++;
        }
        have = k < k1;
        // issued after the fences (they drain outstanding loads), consumed a tile later
        if (have) fetch(t0);
        float dmin = 3.0e38f;
        // The tile's groups of NQ bodies, the NEXT group's LDS reads in flight while the current one is worked through
        // (this walk runs one or two waves to a SIMD: nobody else covers a read's round trip, and with all eight
        // reads followed at once by s_waitcnt lgkmcnt(0) a quarter of the loop was that wait).  Two register sets,
        // used in turn: LDS reads return in order, so the wait before a group is for that group's reads only.
        struct Group { v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], qw[NQ / 2]; };
        auto read_group = [&](int jj, Group &g) {               // 16-byte LDS reads, NQ is a multiple of 4
#pragma unroll
            for (int i = 0; i < NQ / 2; i += 2) {
                const float4 vx = *reinterpret_cast<const float4 *>(tx + jj + 2 * i);
                const float4 vy = *reinterpret_cast<const float4 *>(ty + jj + 2 * i);
                const float4 vz = *reinterpret_cast<const float4 *>(tz + jj + 2 * i);
                const float4 vw = *reinterpret_cast<const float4 *>(tw + jj + 2 * i);
                g.qx[i] = v2f{vx.x, vx.y}; g.qx[i + 1] = v2f{vx.z, vx.w};
                g.qy[i] = v2f{vy.x, vy.y}; g.qy[i + 1] = v2f{vy.z, vy.w};
                g.qz[i] = v2f{vz.x, vz.y}; g.qz[i + 1] = v2f{vz.z, vz.w};
                g.qw[i] = v2f{vw.x, vw.y}; g.qw[i + 1] = v2f{vw.z, vw.w};
            }
        };
        auto work_group = [&](const Group &g) {
            if (MODE == 1)
                pairsN_exact_lean<NQ, ONE_T>(P, ctx, g.qx, g.qy, g.qz, g.qw, 0, nullptr, nullptr, ax, ay, az, flag);
            else
                dmin = fminf(dmin, pairsN_fast<NQ>(ctx, g.qx, g.qy, g.qz, g.qw, eps2f, ax, ay, az));
        };
        if (TILE_PIPELINED) {
            Group a, b;
            read_group(0, a);
            for (int jj = 0; jj < n; jj += 2 * NQ) {
                if (jj + NQ < n) read_group(jj + NQ, b);
                work_group(a);
                if (jj + NQ < n) {
                    if (jj + 2 * NQ < n) read_group(jj + 2 * NQ, a);
                    work_group(b);
                }
            }
        } else {
            for (int jj = 0; jj < n; jj += NQ) {
                Group g;
                read_group(jj, g);
                work_group(g);
            }
        }
    }
    if (k1 < STENCIL) { handoff_publish(force4 + gi, ax, ay, az, flag, valid, ready, k1); return; }
    if (valid) force4.put(P, gc, gi, make_float4(ax, ay, az, __int_as_float(flag)));
}

// The force pass, balanced: `nw` waves (all resident), wave slot s walks the (task, stencil step)
// units from wave_pos[s] up to wave_pos[s + 1] -- the same number of bodies for every wave
// (k_split_tasks).  Most of a wave's share is whole tasks; the task its share ends in is started
// FIRST (steps 0 .. k-1, sums published), then the whole tasks, and LAST the task its share
// begins in is finished from the sums the previous wave slot published at the very start of its
// own work -- so nobody waits in practice, and a particle's sum is still one serial chain of
// fp32 additions in the reference's order.  A share that lies inside one task (few tasks, many
// waves) is one middle piece: consume, walk, publish.
// Wave slots are dealt XCD by XCD like the tasks of k_pairs; k_split_tasks starts every XCD's
// run at a whole task, so the wave that continues a task runs in a workgroup that was
// dispatched no later (block b - 8) or is the same workgroup.
// WALK 0: scalar-load walk, ordinary tasks only (packs, if any, run in k_pairs_merged beside it);
//      1: tile walk for everything, packs of partial slices included (few waves per SIMD);
//      2: scalar-load walk for the ordinary tasks, tile walk for the packs, all in one balanced list.
template <int MODE, int NQ>
__device__ __forceinline__ void merged_pack_task(const DevParams &P, const int *__restrict__ cell_start,
                                                 const SnapSoa snap4,
                                                 const int *__restrict__ active_list,
                                                 const int *__restrict__ active_count,
                                                 const int4 *__restrict__ merged_tasks,
                                                 const ForceBuf force4, int slot, float *tile, WavePace *pace = nullptr);

// nmb (WALK 0, a multiple of 8 so that the XCD dealing is undisturbed): the first nmb workgroups of the
// launch serve the merged packs of partly filled slices instead (merged_pack_task) -- dispatched first,
// their waves are the oldest on their SIMDs and are served first, which is what lets these long,
// stall-prone waves finish well inside the pass.  (As a kernel of their own on a second stream they
// needed a head start to get that: forked at the same moment as the balanced pass they ended with it,
// and the stage took 0.1 ms longer.)
template <int MODE, int NQ, int WALK>
__global__ __launch_bounds__(256, WALK == 0 ? PSAMD_BALANCED_WAVES : WALK == 1 ? 2 : 4) void k_pairs_balanced(DevParams P, const int *__restrict__ cell_start,
                                                        const SnapSoa snap4,
                                                        const float *__restrict__ snap_soa,
                                                        const float *__restrict__ snap_age,
                                                        const int *__restrict__ sorted_id,
                                                        const int *__restrict__ task_list,
                                                        const ForceBuf force4,
                                                        FrameScalars *fs, unsigned long long *trace,
                                                        const int *__restrict__ active_list, const int *__restrict__ active_count,
                                                        const long long *__restrict__ wave_pos, int *__restrict__ task_ready,
                                                        const int4 *__restrict__ merged_tasks, int nmb, StepState *st, int pass, int paced)
{
    __shared__ __attribute__((aligned(16))) float tiles[4][4 * MERGE_TILE];   // up to four 1-KiB tiles per wave
    const int wave = threadIdx.x >> 6;
#ifdef PSAMD_END_TRACE    // (diagnostic build: when every wave ended, and nothing else -- one store at its end: scripts/r5_end_trace.sh)
    struct EndNote {
        unsigned long long *trace;
        __device__ ~EndNote() { if ((threadIdx.x & 63) == 0) trace[blockIdx.x * 4 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memrealtime(); }
    } end_note{trace};
#endif
#ifdef PSAMD_WAVE_TRACE   // (diagnostic build: the wave's whole life, first instruction to last piece -- overwrites what its pieces noted)
    const unsigned long long wave_t0 = __builtin_amdgcn_s_memrealtime();
    struct WholeWave {
        unsigned long long *trace; unsigned long long t0;
        __device__ ~WholeWave() { if ((threadIdx.x & 63) == 0) { unsigned long long *t_ = trace + (size_t)3 * (blockIdx.x * 4 + (threadIdx.x >> 6)); t_[0] = t0; t_[1] = __builtin_amdgcn_s_memrealtime(); } }
    } whole_wave{trace, wave_t0};
#endif
    if (WALK == 0 && (int)blockIdx.x < nmb) {
        // The packs of partly filled slices, dealt round-robin to the 4 * nmb pack waves: a pack wave takes every
        // (4 * nmb)-th pack, one after the other, and paces itself over all of them -- so the launch holds the number of
        // pack workgroups that the packs' share of the WORK asks for, whatever their number (N = 2^22 in 24^3 cells has
        // 6 900 packs: one workgroup per four of them would be the whole GPU).
        const int first = blockIdx.x * 4 + wave, stride = nmb * 4, npack = fs->n_merged;
        if (first < npack) {
            WavePace pace;                       // a pack is 27 steps of (up to) four cells' stencils
            if (paced) {
                const int ticks = st->pairs_ticks[pass];
                pace.t0 = st->pairs_t0[pass];
                pace.per_tick = ticks > 0 ? 1.0f / (float)ticks : 0.f;
                pace.per_unit = 1.0f / (float)(STENCIL * ((npack - first + stride - 1) / stride));
                pace.band = paced;
            }
            for (int pack = first; pack < npack; pack += stride)     // (4 bodies per group: the 8-wide form costs this kernel its sixth wave per SIMD)
                merged_pack_task<MODE, (NQ > 4 ? 4 : NQ)>(P, cell_start, snap4, active_list, active_count, merged_tasks, force4, pack, tiles[wave], &pace);
        }
        return;
    }
    const int slot = xcd_contiguous((int)blockIdx.x - nmb, (int)gridDim.x - nmb) * 4 + wave;
    const long long pos_b = wave_pos[slot], pos_e = wave_pos[slot + 1];
    if (pos_e <= pos_b) return;                              // (positions order like units: task-major, cost inside the task)
    const int ub = resolve_unit(P, pos_b, cell_start, task_list), ue = resolve_unit(P, pos_e, cell_start, task_list);
    if (ue <= ub) return;
    WavePace pace;
    if (WALK == 0 && paced) {
        const int ticks = st->pairs_ticks[pass];
        pace.t0 = st->pairs_t0[pass];
        pace.per_tick = ticks > 0 ? 1.0f / (float)ticks : 0.f;
        pace.per_unit = 1.0f / (float)(ue - ub);
        pace.band = paced;
    }
    struct PassEnd {        // the pass's end, for the next one's clock: the latest wave's last instruction
        StepState *st; int pass; bool on;
        __device__ ~PassEnd() { if (on && (threadIdx.x & 63) == 0) atomicMax(&st->pairs_end[pass], (unsigned long long)__builtin_amdgcn_s_memrealtime()); }
    } pass_end{st, pass, WALK == 0 && paced != 0};
    const int tb = ub / STENCIL, lb = ub - tb * STENCIL;            // first unit: task tb, step lb
    const int tl = (ue - 1) / STENCIL, le = ue - tl * STENCIL;      // last task tl, its steps [.., le)
    // one call site, so one copy of the walk: the pieces in the order they are done
    const bool single = tb == tl;
    const int has_head = (!single && le < STENCIL) ? 1 : 0, has_tail = (!single && lb > 0) ? 1 : 0;
    const int first_whole = tb + has_tail, last_whole = tl + (has_head ? 0 : 1);      // tasks walked whole: [first, last)
    const int nwhole = single ? 0 : last_whole - first_whole;
    const int pieces = single ? 1 : has_head + nwhole + has_tail;
    for (int i = 0; i < pieces; i++) {
        int t, k0 = 0, k1 = STENCIL;
        if (single) { t = tb; k0 = lb; k1 = le; }
        else if (has_head && i == 0) { t = tl; k1 = le; }                 // the head of the last task first: publish early
        else if (i - has_head < nwhole) t = first_whole + (i - has_head);
        else { t = tb; k0 = lb; }                                         // the tail of the first task last: its head was published long ago
        const int nord = fs->n_tasks2;
        if (WALK == 1 || (WALK == 2 && t >= nord)) {
            // task t: an ordinary (cell, slice) task, or -- past them -- merged pack t - n_tasks2
            TileGroups G;
            if (t < nord) {
                const int task = task_list[t], c = task / P.slices, slice = task - c * P.slices;
                G.ng = 1; G.cell[0] = c; G.first[0] = slice * 64; G.count[0] = min(64, active_count[c] - slice * 64);
                G.cell[1] = G.cell[2] = G.cell[3] = c; G.first[1] = G.first[2] = G.first[3] = 0; G.count[1] = G.count[2] = G.count[3] = 0;
            } else {
                const int4 pk = merged_tasks[t - nord];
                const int cells[4] = {pk.x, pk.y, pk.z, pk.w};
                G.ng = 0;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const bool on = cells[q] >= 0;
                    G.cell[q] = on ? cells[q] : pk.x;
                    G.first[q] = on ? (active_count[cells[q]] & ~63) : 0;
                    G.count[q] = on ? (active_count[cells[q]] & 63) : 0;
                    if (on) G.ng = q + 1;
                }
            }
            if (t < nord) pairs_task_tile<MODE, NQ, 1, WALK != 1>(P, cell_start, snap4, force4, G, tiles[wave], active_list, k0, k1, task_ready + t, fs);
            else pairs_task_tile<MODE, NQ, 4, WALK != 1>(P, cell_start, snap4, force4, G, tiles[wave], active_list, k0, k1, task_ready + t, fs);
        } else
            pairs_task<MODE, NQ, true>(P, cell_start, snap4, snap_soa, snap_age, sorted_id, force4, task_list[t], nullptr, trace,
                                       active_list, active_count, k0, k1, task_ready + t, fs,
                                       WALK == 0 ? &pace : nullptr);          // (per_tick == 0: no pacing)
    }
}

// Merged task of the two-pass force pass: the partly filled last slices of up to four cells
// share one wave, each cell's particles in their own run of lanes.  Every lane group has its
// own stencil, so the bodies cannot come as scalar operands here: each group's current 64
// bodies sit in its own LDS tile (SoA, the groups' tiles skewed by 16 bytes so that they use
// different banks -- scripts/microbench/lds_groups.hip) and a lane reads its group's tile.
// All groups walk stencil step k together, tile by tile, for as many rows as the longest of
// their lists; shorter lists are padded with massless bodies far outside the box: such a
// row adds r * 0 = +-0 to a sum that started at +0 (bit-identical, as for kids).  Launched
// on its own (different register budget from k_pairs).

template <int MODE, int NQ>
__device__ __forceinline__ void merged_pack_task(const DevParams &P, const int *__restrict__ cell_start,
                                                 const SnapSoa snap4,
                                                 const int *__restrict__ active_list,
                                                 const int *__restrict__ active_count,
                                                 const int4 *__restrict__ merged_tasks,
                                                 const ForceBuf force4, int slot, float *tile, WavePace *pace)
{
    const int lane = threadIdx.x & 63;
    // (Raising these waves' issue priority -- they run one per SIMD among six of the balanced
    // pass -- was tried: s_setprio(3) ended them 0.6 ms earlier and the
    // balanced pass 0.5 ms later, 2.26 -> 2.48 ms for the stage.)
    const int4 pk = merged_tasks[slot];
    const int cells[4] = {pk.x, pk.y, pk.z, pk.w};
    // lane ranges of the groups
    int off[5] = {0, 0, 0, 0, 0}, ng = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int r = cells[k] >= 0 ? (active_count[cells[k]] & 63) : 0;
        off[k + 1] = off[k] + r;
        if (cells[k] >= 0) ng = k + 1;
    }
    const int g = (lane >= off[1]) + (lane >= off[2]) + (lane >= off[3]);       // a lane past the last group: 3, invalid
    const bool valid = lane < off[4];
    const int c = valid ? (g == 0 ? cells[0] : g == 1 ? cells[1] : g == 2 ? cells[2] : cells[3]) : cells[0];
    const int l = valid ? lane - (g == 0 ? off[0] : g == 1 ? off[1] : g == 2 ? off[2] : off[3]) : 0;
    const int gi = active_list[cell_start[c] + (active_count[c] & ~63) + l];
    const float4 me = snap4[gi];
    const float eps2f = (float)P.eps2;

    // neighbour ranges of all groups: entry e = group * 27 + stencil step, held by lane e % 64
    int tab_nb[2] = {0, 0}, tab_cnt[2] = {0, 0};
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int e = lane + 64 * r, eg = e / 27, ek = e - eg * 27;
        const int ec = eg == 0 ? cells[0] : eg == 1 ? cells[1] : eg == 2 ? cells[2] : eg == 3 ? cells[3] : -1;
        if (ec >= 0) {
            int i1, i2, i3;
            cell_coords(P, ec, i1, i2, i3);
            const int nc = local_cell(P, i3 + c_stencil[ek][2], i1 + c_stencil[ek][1], i2 + c_stencil[ek][0]);
            if (nc >= 0) {
                tab_nb[r] = cell_start[nc];
                tab_cnt[r] = min(cell_start[nc + 1] - tab_nb[r], P.max_per_cell);
            }
        }
    }
    const float *tx = tile + (valid ? g : 0) * MERGE_TILE, *ty = tx + 64, *tz = tx + 128, *tw = tx + 192;
    const float far = 1.0e6f;                                       // padding body, mass 0
    const PairCtx ctx = {me.x, me.y, me.z, 0.f, 0, gi, false};
    float ax = 0.f, ay = 0.f, az = 0.f;
    int flag = 0;
    for (int k = 0; k < 27; k++) {
        int nbs[4], cnts[4], longest = 0;
#pragma unroll
        for (int gg = 0; gg < 4; gg++) {
            const int e = gg * 27 + k;
            nbs[gg] = __builtin_amdgcn_readlane(e < 64 ? tab_nb[0] : tab_nb[1], e & 63);
            cnts[gg] = gg < ng ? __builtin_amdgcn_readlane(e < 64 ? tab_cnt[0] : tab_cnt[1], e & 63) : 0;
            longest = max(longest, cnts[gg]);
        }
        for (int t0 = 0; t0 < longest; t0 += 64) {
            PS_WAVE_SYNC();                                         // previous tiles fully consumed
#pragma unroll
            for (int gg = 0; gg < 4; gg++) {
                if (gg < ng) {
                    float4 v = make_float4(far, far, far, 0.f);
                    if (lane < cnts[gg] - t0) v = snap4[nbs[gg] + t0 + lane];
                    float *t = tile + gg * MERGE_TILE + lane;
                    t[0] = v.x; t[64] = v.y; t[128] = v.z; t[192] = v.w;
                }
            }
            PS_WAVE_SYNC();
            const int n = (min(64, longest - t0) + NQ - 1) & ~(NQ - 1);
            float dmin = 3.0e38f;
            for (int jj = 0; jj < n; jj += NQ) {
                v2f qx[NQ / 2], qy[NQ / 2], qz[NQ / 2], qw[NQ / 2];   // 16-byte LDS reads, NQ is a multiple of 4
#pragma unroll
                for (int i = 0; i < NQ / 2; i += 2) {
                    const float4 vx = *reinterpret_cast<const float4 *>(tx + jj + 2 * i);
                    const float4 vy = *reinterpret_cast<const float4 *>(ty + jj + 2 * i);
                    const float4 vz = *reinterpret_cast<const float4 *>(tz + jj + 2 * i);
                    const float4 vw = *reinterpret_cast<const float4 *>(tw + jj + 2 * i);
                    qx[i] = v2f{vx.x, vx.y}; qx[i + 1] = v2f{vx.z, vx.w};
                    qy[i] = v2f{vy.x, vy.y}; qy[i + 1] = v2f{vy.z, vy.w};
                    qz[i] = v2f{vz.x, vz.y}; qz[i + 1] = v2f{vz.z, vz.w};
                    qw[i] = v2f{vw.x, vw.y}; qw[i + 1] = v2f{vw.z, vw.w};
                }
                if (MODE == 1)
                    pairsN_exact_lean<NQ>(P, ctx, qx, qy, qz, qw, 0, nullptr, nullptr, ax, ay, az, flag);
                else
                    dmin = fminf(dmin, pairsN_fast<NQ>(ctx, qx, qy, qz, qw, eps2f, ax, ay, az));
            }
        }
        if (pace) pace->step();
    }
    if (valid) force4.put(P, c, gi, make_float4(ax, ay, az, 0.f));
}

template <int MODE, int NQ>
__global__ __launch_bounds__(256, 6) void k_pairs_merged(DevParams P, const int *__restrict__ cell_start,
                                                      const SnapSoa snap4,
                                                      const int *__restrict__ active_list,
                                                      const int *__restrict__ active_count,
                                                      const int4 *__restrict__ merged_tasks,
                                                      const ForceBuf force4, const FrameScalars *__restrict__ fs)
{
    __shared__ __attribute__((aligned(16))) float tiles[4][4 * MERGE_TILE];
    const int wave = threadIdx.x >> 6;
    const int slot = blockIdx.x * 4 + wave;
    if (slot >= fs->n_merged) return;
    merged_pack_task<MODE, NQ>(P, cell_start, snap4, active_list, active_count, merged_tasks, force4, slot, tiles[wave]);
}

// ------------------------------------------------------------------ self test
// Compare the hand-written sqrt / reciprocal with the compiler's correctly rounded forms
// on every float whose bit pattern lies in [lo_bits, hi_bits].  out[0..3] = mismatch
// counts of sqrt_rn_short, rcp_rn_newton, their composition (what the pair kernel uses)
// and of the rejected one-transcendental shortcut; out[4] = mismatches of inv_sqrt_guarded that it
// did not report, out[5] = inputs it reported; out[8..15] / out[16..23] = first
// offending inputs of sqrt / composition; out[24], out[25] = cursors.
__global__ void k_selftest_math(uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out)
{
    const uint64_t span = (uint64_t)hi_bits - lo_bits + 1;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad[6] = {0, 0, 0, 0, 0, 0};
    for (; i < span; i += stride) {
        const float a = __uint_as_float(lo_bits + (uint32_t)i);
        const float s_ref = sqrtf(a), r_ref = 1.0f / a, c_ref = 1.0f / s_ref;
        if (__float_as_uint(sqrt_rn_short(a)) != __float_as_uint(s_ref)) {
            bad[0]++;
            const unsigned long long k = atomicAdd(&out[25], 1ull);
            if (k < 8) out[8 + k] = __float_as_uint(a);
        }
        if (__float_as_uint(rcp_rn_newton(a)) != __float_as_uint(r_ref)) bad[1]++;
        if (__float_as_uint(inv_sqrt_selected(a)) != __float_as_uint(c_ref)) {
            bad[2]++;
            const unsigned long long k = atomicAdd(&out[24], 1ull);
            if (k < 8) out[16 + k] = __float_as_uint(a);
        }
        if (__float_as_uint(inv_sqrt_one_transcendental(a)) != __float_as_uint(c_ref)) bad[3]++;
        bool tie = false;
        const float gq = inv_sqrt_guarded(a, tie);
        if (tie) bad[5]++;
        else if (__float_as_uint(gq) != __float_as_uint(c_ref)) bad[4]++;
    }
    for (int k = 0; k < 6; k++) if (bad[k]) atomicAdd(&out[k], bad[k]);
}

// out[0] += number of floats x with bits in [lo_bits, hi_bits] for which the fp32 add of
// eps2f differs from the reference's double add rounded to float
__global__ void k_validate_eps(uint32_t lo_bits, uint32_t hi_bits, double eps2, float eps2f, unsigned long long *out)
{
    const uint64_t span = (uint64_t)hi_bits - lo_bits + 1;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (; i < span; i += stride) {
        const float x = __uint_as_float(lo_bits + (uint32_t)i);
        if (__float_as_uint(x + eps2f) != __float_as_uint((float)((double)x + eps2))) bad++;
    }
    if (bad) atomicAdd(out, bad);
}

hipError_t launch_validate_eps(hipStream_t st, uint32_t lo_bits, uint32_t hi_bits, double eps2, float eps2f,
                               unsigned long long *out)
{
    k_validate_eps<<<2048, 256, 0, st>>>(lo_bits, hi_bits, eps2, eps2f, out);
    return hipGetLastError();
}

hipError_t launch_selftest_math(hipStream_t st, uint32_t lo_bits, uint32_t hi_bits, unsigned long long *out24)
{
    k_selftest_math<<<4096, 256, 0, st>>>(lo_bits, hi_bits, out24);
    return hipGetLastError();
}

// How one pass of the pair stage is launched, from the hint of its task count: everything that shapes the
// launches and is not read from device memory by the kernels themselves (what a captured graph is keyed by).
struct PairShape {
    bool two, merge, balanced, tile, packs_in_list;
    int nw;                  // wave slots of the balanced force pass
    int nmb;                 // workgroups of the same launch, ahead of them, that serve the packs of partly filled slices (WALK 0)
};

static PairShape pair_shape(const DevParams &P, bool lean, int64_t hint)
{
    const int64_t tasks_hint = hint & 0xffffffffll, packs_hint = hint >> 32;      // (capi.hip, pairs_hint)
    PairShape s{};
    s.two = lean && P.two_pass;
    // leftover slices of several cells in one wave (k_pairs_merged).  A merged wave is long and
    // stalls on its tile loads; a small share (a slab with fewer than ~3 tasks per SIMD) has too
    // little other work to cover that and it becomes the critical path (measured on 1/4 and 1/8
    // shares of the N = 2^20 cloud).
    static const bool merge_off = std::getenv("PSAMD_NO_MERGE") != nullptr;
    static const bool balance_off = std::getenv("PSAMD_NO_BALANCE") != nullptr;
    static const int waves_env = std::getenv("PSAMD_WAVES") ? std::atoi(std::getenv("PSAMD_WAVES")) : 0;
    s.merge = s.two && !merge_off && (P.world == 1 || tasks_hint >= 3000);
    s.balanced = s.two && !balance_off;
    // Balanced pass: a fixed number of waves, all resident, each walking the same number of
    // bodies.  At least four per SIMD when there are that many tasks (fewer cannot cover their
    // scalar-load latency: 1024 / 2048 / 4096 / 6144 waves took 3.73 / 2.54 / 2.27 / 2.29 ms on
    // the N = 2^20 cloud), but not more waves than tasks (a task cut in three or more pieces is
    // a chain of waves that wait for each other).
    if (s.balanced) {
        static const int waves_per_simd = std::getenv("PSAMD_WAVES_PER_SIMD") ? std::atoi(std::getenv("PSAMD_WAVES_PER_SIMD")) : PSAMD_BALANCED_WAVES;      // (A/B runs)
        s.nw = 1024 * (int)std::min<int64_t>(waves_per_simd, std::max<int64_t>(1, tasks_hint / 1024));
        if (waves_env >= 32) s.nw = std::min(waves_env & ~31, MAX_PAIR_WAVES);
    }
    // few waves per SIMD (a slab of a multi-GPU run): the scalar-load walk cannot cover its own
    // load latency, bodies come through LDS tiles fetched a tile ahead instead -- and the partly
    // filled last slices are packed into tasks of the same pass
    static const int tile_env = std::getenv("PSAMD_TILE") ? std::atoi(std::getenv("PSAMD_TILE")) : -1;
    static const bool unified_packs = std::getenv("PSAMD_UNIFIED_PACKS") != nullptr;
    static const bool tile_packs = std::getenv("PSAMD_TILE_PACKS") != nullptr;
    s.tile = s.balanced && (tile_env >= 0 ? tile_env != 0 : s.nw <= 2048);
    // The packs of partly filled last slices as tasks of the balanced pass itself (tile walk).
    // Measured (pair stage, N = 2^20): one GPU, 8200 tasks: beside the pass in k_pairs_merged 2.31 ms,
    // in the list 2.40 (one kernel holding both walks needs 99 VGPRs: 4 waves per SIMD, not 6);
    // half the cloud (a slab of two): 1.67 vs 1.40 -- the separate kernel's 512 waves end long after
    // a pass that has only 4 waves per SIMD; an eighth (tile walk): no packs 0.58, packs 0.60 -- a
    // pack's four-group walk costs more than the two tasks it saves.  So (rounds 2-3): in the list for the slabs
    // that use the scalar walk, beside the pass on one GPU, none with the tile walk.
    // (Round 4: with persistent pack workgroups sized by the packs' share of the work and the waves paced, a slab of two
    // is served better by the one-GPU form too -- pair stage 1.32 -> 1.12 ms -- so the packs are in the list only on request.)
    s.packs_in_list = s.balanced && !merge_off && (s.tile ? tile_packs : (s.merge && unified_packs));
    if (s.packs_in_list) { s.merge = false; s.nw = std::min(s.nw, 4096); }      // (98 VGPRs with the tile walk in: 4 resident waves per SIMD)
    if (s.tile) s.merge = false;                  // no separate merged kernel beside a tile-walk pass
    // The packs' workgroups are the first of the same launch and hold residency slots for about half of it: with
    // a wave slot for every resident wave besides, the workgroups dispatched last could only start when a pack ended
    // (wave trace, round 4: a quarter of the balanced waves started 0.6-0.9 ms into a 2.3-ms launch).  So the balanced
    // part gets as many wave slots as the packs leave free: everything is resident from the start.
    // How many pack workgroups: the packs' share of the pass's work (a pack costs about PACK_COST ordinary tasks: four lane
    // groups with their own LDS tiles), in workgroups of the resident set; a pack wave takes several packs one after the
    // other.  Without a hint (a context's first step) a quarter.
    static const bool nw_minus_packs = !(std::getenv("PSAMD_NW_PACKS") && std::atoi(std::getenv("PSAMD_NW_PACKS")) == 0);
    static const double pack_cost = std::getenv("PSAMD_PACK_COST") ? std::atof(std::getenv("PSAMD_PACK_COST")) : 1.4;
    s.nmb = 0;
    if (s.merge && !s.packs_in_list && s.balanced) {
        const int resident = s.nw / 4;                     // workgroups the launch keeps resident
        if (nw_minus_packs && waves_env < 32 && s.nw >= 4096) {
            const double pw = pack_cost * (double)packs_hint, tw = (double)std::max<int64_t>(tasks_hint - packs_hint, 1);
            const double share = packs_hint > 0 ? pw / (pw + tw) : 0.25;
            int wgs = ((int)(resident * share + 0.5) + 7) & ~7;
            wgs = std::max(8, std::min(wgs, resident / 2));
            if (packs_hint > 0) wgs = std::min(wgs, (int)(((packs_hint + 3) / 4 + 7) & ~7));      // (a pack wave with no pack is a wasted slot)
            s.nmb = wgs;
            s.nw = (s.nw - 4 * wgs) & ~255;
        } else
            s.nmb = (int)std::min<int64_t>(((packs_hint > 0 ? (packs_hint + 3) / 4 : resident / 4) + 7) & ~7, resident);
    }
    return s;
}

uint64_t launch_pairs_shape(const DevParams &P, int64_t tasks_hint)
{
    const PairShape s = pair_shape(P, P.lean_math != 0, tasks_hint);
    return (uint64_t)(s.nw / 32) | (s.merge ? 1ull << 10 : 0) | (s.tile ? 1ull << 11 : 0) | (s.packs_in_list ? 1ull << 12 : 0) | (s.balanced ? 1ull << 13 : 0) | ((uint64_t)(s.nmb / 8) << 14);
}

template <int MODE, int NQ>
static hipError_t launch_pairs_mode(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t ev_force, int64_t tasks_hint, int pass, int64_t live_bound)
{
    const int ncomp = comp_count(P);
    if (ncomp <= 0) return hipSuccess;
    const ForceBuf fbuf = force_buf(d);
    const int tasks = ncomp * P.slices;
    const PairShape shape = pair_shape(P, MODE != 0, tasks_hint);
    const bool two = shape.two, merge = shape.merge, balanced = shape.balanced, tile = shape.tile, packs_in_list = shape.packs_in_list;
    const int nw = shape.nw;
    if (two) {
        // collision flags and the per-cell lists of the particles that need a force, then the plan of the force pass
        if (P.max_per_cell + HALO_CAP / 2 <= 1024)
            k_collide_cell<1024><<<ncomp, 256, 0, st>>>(P, d.cell_start, d.snap_soa, d.snap_age, d.sorted_id, d.snap_cid, d.halo_count, d.halo_f,
                                                        d.halo_id, d.active_list, d.active_count, d.task_cost, fbuf);
        else
            k_collide_cell<2560><<<ncomp, 256, 0, st>>>(P, d.cell_start, d.snap_soa, d.snap_age, d.sorted_id, d.snap_cid, d.halo_count, d.halo_f,
                                                        d.halo_id, d.active_list, d.active_count, d.task_cost, fbuf);
        k_plan_force<<<8, 1024, 0, st>>>(P, balanced ? nw : 0, packs_in_list ? 2 : merge ? 1 : 0, d.cell_start, d.active_count, d.task_cost,
                                         d.task_list2, d.ctask_start, d.cost_start, d.merged_tasks, d.wave_pos, d.fs, d.trace, d.st, pass);
    }
    if (ev_force) (void)hipEventRecord(ev_force, st);      // timing: the force pass proper starts here
    const int *task_list = two ? d.task_list2 : d.task_list;
    const int *active_list = two ? d.active_list : nullptr, *active_count = two ? d.active_count : nullptr;
    // the hand-off flags are indexed by task number, which starts at 0 in every pass of a frame:
    // each pass has its own block of them (both zeroed with the frame)
    int *task_ready = d.task_ready + (size_t)pass * P.n_local_cells * P.slices;
    if (balanced) {
        constexpr int M = MODE == 0 ? 1 : MODE;
        static const int paced = std::getenv("PSAMD_PACE") ? std::atoi(std::getenv("PSAMD_PACE")) : 20;      // (A/B runs: 0 = no pacing of the waves; else WavePace::band)
        // the packs of partly filled slices (merge): the first nmb workgroups of the same launch
        const int nmb = merge ? std::max(8, shape.nmb) : 0;
#define PS_BALANCED_Q(W, Q) k_pairs_balanced<M, Q, W><<<nmb + nw / 4, 256, 0, st>>>(P, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_soa, d.snap_age, d.sorted_id, task_list, \
                                                                     fbuf, d.fs, d.trace, active_list, active_count, d.wave_pos, task_ready, d.merged_tasks, nmb, d.st, pass, paced)
#define PS_BALANCED(W) PS_BALANCED_Q(W, NQ)
        // The tile walk -- a wave with its SIMD (almost) to itself -- takes 16 bodies per group: every group costs such a
        // wave two branches on a vector compare and the tail of three chains of dependent additions, all of it exposed;
        // half as many groups: -5 % on the pair stage of an eighth of the N = 2^20 cloud, -6 % in the tolerance mode (profiles/r4_ab_tile_nq.txt).
        // (The next group's distances between a group's scale factors and its additions, in one basic block: 9 % SLOWER.)
        static const int tile_nq = std::getenv("PSAMD_TILE_NQ") ? std::atoi(std::getenv("PSAMD_TILE_NQ")) : 16;      // (A/B runs)
        // (The same in the scalar walk where a SIMD holds four waves -- N = 2^22 on eight ranks -- gave 1 %: not kept.)
        if (tile && NQ == 8 && tile_nq == 16) PS_BALANCED_Q(1, 16);
        else if (tile) PS_BALANCED(1); else if (packs_in_list) PS_BALANCED(2); else PS_BALANCED(0);
#undef PS_BALANCED
#undef PS_BALANCED_Q
    }
    else {
        k_pairs<MODE, NQ><<<(tasks + 3) / 4, 256, 0, st>>>(P, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.snap_soa, d.snap_age, d.sorted_id, task_list, fbuf,
                                                         d.fs, d.trace, active_list, active_count);
        // (unbalanced pass, A/B runs only: the packs as a kernel of their own behind it)
        if (merge) k_pairs_merged<MODE == 0 ? 1 : MODE, NQ><<<(ncomp + 3) / 4, 256, 0, st>>>(
                P, d.cell_start, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.active_list, d.active_count, d.merged_tasks, fbuf, d.fs);
    }
    if (MODE != 0 && (P.flags & PSAMD_FLAG_ALL_PAIRS)) {
        // All-pairs forces: what ran above is the stencil's chain; now every other cell (k_allp_far) and the sum.
        // (two == true here: all-pairs contexts are created only with the two-pass pair stage.)  Where the far cells are
        // found: the own snapshot (one GPU: local cell == global cell, lengths from consecutive starts) or the
        // all-gathered snapshot of all ranks with its index by global cell.
        FarCells far;
        const bool gathered = P.world > 1;
        const float *far_buf = gathered ? reinterpret_cast<const float *>(d.allg_in) : d.snap_soa;
        const int *far_start = gathered ? d.gstart : d.cell_start, *far_n = gathered ? d.gn : nullptr;
        far.plane = gathered ? (unsigned long long)P.allg_cap : (unsigned long long)P.sorted_cap;
        far.part_acc = d.part_acc; far.part_plane = (unsigned long long)d.part_tasks * 64;
        // dense tasks: at most the particles alive (the host's bound; a slab also computes its neighbour's lent layers:
        // every entry of the sorted order).  The kernels go by the device's own count.
        const int64_t dense_bound = std::min<int64_t>(d.part_tasks, ((live_bound >= 0 && P.world == 1) ? live_bound : (int64_t)P.sorted_cap) / 64 + 2);
        k_allp_prefix<<<1, 1024, 0, st>>>(P, d.active_count, d.act_start);
        k_allp_dense<<<(ncomp + 3) / 4, 256, 0, st>>>(P, d.cell_start, d.active_list, d.active_count, d.act_start, d.dense_gi, d.dense_cell);
        k_allp_far<MODE == 0 ? 1 : MODE, NQ><<<(unsigned)((dense_bound * ALLP_PARTS + 3) / 4), 256, 0, st>>>(P, SnapSoa{d.snap_soa, (size_t)P.sorted_cap}, d.act_start, d.dense_gi, d.dense_cell,
                                                                                                  far, far_buf, far_start, far_n);
        k_allpairs_combine<<<(unsigned)((dense_bound * 64 + 255) / 256), 256, 0, st>>>(P, d.act_start, d.dense_gi, d.dense_cell, far, fbuf);
    }
    return hipGetLastError();
}

hipError_t launch_pairs(hipStream_t st, const DevParams &P, const DeviceState &d, hipEvent_t ev_force, int64_t tasks_hint, int pass, int64_t live_bound)
{
    // fast math shares the lean modes' validity range (finite 1/sqrt(eps2^3))
    static const int fast_nq = std::getenv("PSAMD_FAST_NQ") ? std::atoi(std::getenv("PSAMD_FAST_NQ")) : 8;      // (A/B runs)
    if ((P.flags & PSAMD_FLAG_FAST_MATH) && P.lean_math)
        return fast_nq == 4 ? launch_pairs_mode<2, 4>(st, P, d, ev_force, tasks_hint, pass, live_bound) : launch_pairs_mode<2, 8>(st, P, d, ev_force, tasks_hint, pass, live_bound);
    // 8 pairs per slow-branch test: measured 3 % (full GPU) to 5 % (a 1/8 share) faster than 4
    if (P.lean_math) return launch_pairs_mode<1, 8>(st, P, d, ev_force, tasks_hint, pass, live_bound);
    return launch_pairs_mode<0, 4>(st, P, d, ev_force, tasks_hint, pass, live_bound);
}

}  // namespace psamd
