"""Randomised GPU-vs-oracle parity run (not part of the test suite: minutes of oracle time).
Each case draws a size, a density profile, velocities, ages (kids, adults, elders), dt, EPS2, the
life-cycle switches and a world size (1 = plain context, 2..4 = slab contexts on this GPU), runs a
few steps and compares every byte of the reference-layout state with the oracle after each.
usage: python scripts/fuzz_parity.py [--cases 40] [--seed 1] [--log gpurun_out/fuzz.log]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, os.path.abspath(d))
import oracle_py as O                     # noqa: E402  (the checker)
import particlesystem_amd as ps           # noqa: E402
from particlesystem_amd.slab import merge_owned, step_local   # noqa: E402
from util import assert_same_particles, explosion_rng, oracle_cfg_from   # noqa: E402


def draw_case(rng, sizes, max_steps=6, worlds=(1, 1, 2, 3, 4), nan_draw=True):
    """nan_draw=False: the draws as they were before the not-a-number particles were added (campaigns up to seed 3303 keep their cases)"""
    n = int(rng.choice(sizes))
    geo = [{}, {}, {"chunk_factor": 2, "chunk_dim": 6}, {"chunk_factor": 3, "chunk_dim": 4}, {"chunk_factor": 5, "chunk_dim": 4},
           {"chunk_factor": 4, "chunk_dim": 3, "cell_size": 2.5}, {"chunk_factor": 5, "chunk_dim": 3}][int(rng.integers(0, 7))]
    G = geo.get("chunk_factor", 4) * geo.get("chunk_dim", 4)
    cs = geo.get("cell_size", 5.0)
    L = 0.5 * G * cs * 0.9975
    half = float(rng.choice([L, L, L / 2, L / 5]))                 # whole box, or a denser blob (cell overflow, collapse)
    xyz = rng.uniform(-half, half, (n, 3)).astype(np.float32)
    if G % 2:                                                      # odd grids are not centred: i = floor(+-c / cs) + G // 2
        xyz += np.float32(0.5 * cs) * np.array([1, -1, -1], np.float32)
    if rng.random() < 0.3:                                         # a second, very dense clump
        m = n // 10
        xyz[:m] = (rng.normal(0, 1.5, (m, 3)) + rng.uniform(-0.7 * L, 0.7 * L, 3)).clip(-L, L).astype(np.float32)
    if rng.random() < 0.5:                                         # some coordinates exactly on cell faces, or one ulp beside them
        m = n // 8
        pick = rng.choice(n, m, replace=False)
        axis = rng.integers(0, 3, m)
        face = (np.round(xyz[pick, axis] / np.float32(cs)) * np.float32(cs)).astype(np.float32)
        nudge = rng.integers(-1, 2, m)
        face = np.where(nudge < 0, np.nextafter(face, np.float32(-1e9)), np.where(nudge > 0, np.nextafter(face, np.float32(1e9)), face)).astype(np.float32)
        lim = np.float32(0.5 * G * cs)
        lo_edge, hi_edge = (-(G // 2) * cs, (G - G // 2) * cs)
        xyz[pick, axis] = face
        # keep every point inside the box [lo, hi) of its axis (x runs with +, y and z with -)
        sgn = np.array([1.0, -1.0, -1.0], np.float32)
        u = xyz * sgn
        u = np.clip(u, np.float32(lo_edge), np.nextafter(np.float32(hi_edge), np.float32(-1e9)))
        xyz = (u * sgn).astype(np.float32)
    vmax = float(rng.choice([0.0, 5.0, 60.0, 300.0]))
    v = rng.uniform(-vmax, vmax, (n, 3)).astype(np.float32) if vmax else None
    over = dict(geo)
    if rng.random() < 0.4:
        over["dt"] = float(rng.choice([0.01, 0.05, 0.2]))
    # ages against the thresholds of this dt: PARTICLE_LIFE = 300 dt, KID_AGE = life / 10 (common.h:58-61)
    life = 300.0 * over.get("dt", 0.05)
    kid = life / 10.0
    age = rng.uniform(0.0, 0.6 * life, n).astype(np.float32)       # kids and adults
    sel = rng.random(n)
    ulps = lambda v, k: (np.float32(v).view(np.int32) + k).view(np.float32)      # v moved by k ulps
    around = rng.integers(-3, 4, n).astype(np.int32)
    age = np.where(sel < 0.08, ulps(kid, around), age)             # at the kid threshold, to the ulp
    age = np.where((sel >= 0.08) & (sel < 0.16), ulps(life, around), age)          # at the end of life
    age = np.where((sel >= 0.16) & (sel < 0.22), np.float32(life) - np.float32(over.get("dt", 0.05)) * rng.integers(0, 3, n).astype(np.float32), age)
    age = np.where((sel >= 0.22) & (sel < 0.26), rng.uniform(life, 2 * life, n), age).astype(np.float32)       # over age
    births = rng.random() < 0.5
    fert = rng.uniform(0.2 * life, 0.7 * life, n).astype(np.float32) if births else (1e6 + np.arange(n)).astype(np.float32)
    if births and rng.random() < 0.5:                              # fertility age exactly the age some steps from now
        k = rng.integers(0, 4, n).astype(np.float32)
        fert = np.where(rng.random(n) < 0.3, age + k * np.float32(over.get("dt", 0.05)), fert).astype(np.float32)
    if rng.random() < 0.3:
        over["eps2"] = float(rng.choice([1e-20, 0.01, 1.0]))
    if rng.random() < 0.2:
        over["collision_radius"] = float(rng.choice([0.0, 0.1, 1.0]))
    w = None
    if rng.random() < 0.3:                                         # other masses, some of them zero
        w = np.where(rng.random(n) < 0.1, 0.0, rng.uniform(1.0, 100.0, n)).astype(np.float32)
    world = int(rng.choice(list(worlds)))
    world = min(world, G // 2)
    cuts = None
    if world > 1 and rng.random() < 0.4:                           # the caller's own cuts: >= 2 layers per rank
        while True:
            inner = np.sort(rng.choice(np.arange(2, G - 1), world - 1, replace=False))
            cuts = [0] + [int(v) for v in inner] + [G]
            if min(b - a for a, b in zip(cuts, cuts[1:])) >= 2:
                break
    interior = bool(world > 1 and rng.random() < 0.3)
    reupload = bool(world == 1 and not births and rng.random() < 0.4)   # hand the state to a fresh context half way
    # (not with births: the birth RNG is keyed on the context's step counter, which a fresh context restarts)
    steps = int(rng.integers(2, max_steps + 1))
    replay = bool(rng.random() < 0.35)
    # (last draw, so that the cases of earlier campaigns keep their other draws) now and then two particles whose
    # velocity is not a number -- what a child born with the direction (0, 0, 0) gets (0/0, ps.cpp:1306-1333): a kid and
    # an adult; a step later their position is not a number either and the reference files them under cell 0
    if nan_draw and rng.random() < 0.15:
        if v is None:
            v = np.zeros((n, 3), np.float32)
        pick = rng.choice(n, 2, replace=False)
        v[pick] = np.nan
        age[pick[0]] = np.float32(0.3 * kid)
        age[pick[1]] = np.float32(0.4 * life)
    return dict(n=n, xyz=xyz, v=v, age=age, fert=fert, w=w, births=births, over=over, world=world, steps=steps,
                cuts=cuts, interior=interior, reupload=reupload, replay=replay,
                desc="n=%d G=%d half=%.1f vmax=%g births=%d masses=%d world=%d cuts=%r interior=%d reupload=%d nan=%d %r" %
                     (n, G, half, vmax, births, w is not None, world, cuts, interior, reupload, int(v is not None and bool(np.isnan(v).any())), over))


def run_case(c, seed, graphs=False):
    """graphs: the contexts replay their stage sequences as hipGraphs (psamd_set_graphs)"""
    has_nan = c["v"] is not None and bool(np.isnan(c["v"]).any())
    flags = ps.FLAG_EXPLOSIONS if (c["births"] or has_nan) else 0      # (births on: the far outbox exists in worlds of four or more)
    extra = dict(seed=seed) if c["births"] else {}
    W = c["world"]
    if c["cuts"]:
        extra["cuts"] = c["cuts"]
    mk = lambda r: ps.ParticleSystem(ps.default_config(rank=r, world=W, flags=flags, **extra, **c["over"]))
    def mk(r, mk0=mk):
        g = mk0(r)
        if graphs:
            g.set_graphs(True)
        return g
    try:
        ranks = [mk(r) for r in range(W)]
    except ps.PsamdError as e:
        if "no slab partition" in str(e):      # random cuts (or many ranks on a coarse chunk grid) may leave a rank's reads with a non-neighbour
            return "plan refused (world %d, cuts %r)" % (W, c["cuts"])
        raise
    carried = {}                                                 # counters of a context that handed its state on
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    if c["births"]:
        o.set_rng(explosion_rng(seed))
    try:
        ids_o = o.fill(c["xyz"], age=c["age"], fert_age=c["fert"], w=c["w"])
    except Exception as e:                                           # the segment of a dense clump is full: not a parity case
        for g in ranks:
            g.close()
        o.close()
        return "skipped (%s)" % str(e)[:60]
    if c["v"] is not None:
        p = o.particles
        p["vx"][ids_o], p["vy"][ids_o], p["vz"][ids_o] = c["v"].T
    for g in ranks:
        g.fill_particles(c["xyz"], age=c["age"], fert_age=c["fert"], vxyz=c["v"], w=c["w"])
        if c.get("replay"):
            g.snapshot_save()
    for k in range(c["steps"]):
        try:
            if W == 1:
                if c["reupload"] and k == c["steps"] // 2:
                    # the reference's buffers out of one context and into a fresh one
                    old = ranks[0]
                    p, (qi, q) = old.download_particles(), old.download_queues()
                    new = mk(0)
                    new.upload_particles(p); new.upload_queues(qi, q)
                    carried = dict(old.counters)
                    old.close()
                    ranks[0] = new
                ranks[0].step(1)
            else:
                step_local(ranks, overlap_interior=c["interior"])
                for g in ranks:
                    g.synchronize()
        except ps.PsamdError as e:
            # documented refusals of the slab path: more overflow kills than a status record carries, a message
            # smaller than what a fast dense cloud sends (halo_cap_cell, xfer_cap, the hop-two messages' 1024 records).
            # (Served since round 3, no longer refusals: the chunk-list capacity rule across ranks, a two-layer jump
            # over a rank whose state is a single layer.)
            # (Served since round 3 too: a particle whose position is not a number on a rank far from the cell the
            # reference files it under -- the far outbox.  ERR_SLAB_MISMATCH / ERR_FOREIGN_CELL are failures again.)
            if W > 1 and ("status message" in str(e) or "had no room" in str(e)):
                for g in ranks:
                    g.close()
                o.close()
                return "refused (%s)" % str(e)[:70]
            raise
        o.step(1)
        plans = [g.slab_plan() for g in ranks]
        got = ranks[0].download_particles() if W == 1 else merge_owned([g.download_particles() for g in ranks], plans)
        assert_same_particles(got, o.particles, "step %d" % (k + 1))
        qs = [g.download_queues() for g in ranks]
        qi = qs[0][0] if W == 1 else merge_owned([q[0] for q in qs], plans, "records")
        q = qs[0][1] if W == 1 else merge_owned([q[1] for q in qs], plans)
        assert qi.tobytes() == o.queue_info.tobytes() and np.array_equal(q, o.queue), "queues differ at step %d" % (k + 1)
    if c.get("replay") and not c["reupload"]:
        # what bench.py's timed loop relies on: restore the snapshot taken after the fill, run the same steps
        # again -- the state must come out the same, byte for byte (one context and slabs alike)
        want = [g.download_particles() for g in ranks], [g.download_queues() for g in ranks]
        for g in ranks:
            g.snapshot_restore()
        for k in range(c["steps"]):
            if W == 1:
                ranks[0].step(1)
            else:
                step_local(ranks, overlap_interior=c["interior"])
        for r, g in enumerate(ranks):
            g.synchronize()
            assert g.download_particles().tobytes() == want[0][r].tobytes(), "replay after snapshot_restore differs (rank %d)" % r
            qi, q = g.download_queues()
            assert qi.tobytes() == want[1][r][0].tobytes() and np.array_equal(q, want[1][r][1]), "queues differ after replay (rank %d)" % r
    cnt = {k: sum(g.counters[k] for g in ranks) + carried.get(k, 0) for k in ("relocations", "births", "deaths_collision", "cell_overflow_kills")}
    twice = 2 if (c.get("replay") and not c["reupload"]) else 1              # the event counters are cumulative: a replay counts again
    for k, v in cnt.items():
        assert v == twice * o.counters[k], (k, v, o.counters[k], twice)
    cnt = {k: v // twice for k, v in cnt.items()}
    if graphs:
        cnt["graph_replays"] = sum(g.graph_stats()[0] for g in ranks)       # (raises if the runtime refused a capture)
    for g in ranks:
        g.close()
    o.close()
    return "ok %r" % cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default=None)
    ap.add_argument("--sizes", default="3000,12000,40000,90000", help="particle counts to draw from")
    ap.add_argument("--max-steps", type=int, default=6)
    ap.add_argument("--worlds", default="1,1,2,3,4", help="world sizes to draw from (capped at half the grid's layers)")
    ap.add_argument("--graphs", action="store_true", help="every other case with the stage sequences as hipGraphs")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    log = open(a.log, "a") if a.log else sys.stdout
    bad = 0
    for i in range(a.cases):
        c = draw_case(rng, [int(v) for v in a.sizes.split(",")], a.max_steps, [int(v) for v in a.worlds.split(",")])
        if os.environ.get("FUZZ_ANNOUNCE"):
            print("start case %d [%s]" % (i, c["desc"]), file=log, flush=True)
        t = time.time()
        try:
            res = run_case(c, 1000 + i, graphs=a.graphs and i % 2 == 1)
        except AssertionError as e:
            res = "MISMATCH %s" % str(e)[:300]
            bad += 1
        print("case %d [%s] %.1fs: %s" % (i, c["desc"], time.time() - t, res), file=log, flush=True)
    print("fuzz done: %d cases, %d mismatches" % (a.cases, bad), file=log, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
