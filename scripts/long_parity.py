"""A long free-running parity run (not part of the suite): N particles with births on, `steps` steps, one context
and `world` slabs against the oracle, every byte every `every` steps.  usage: python scripts/long_parity.py [--n 20000] [--steps 200] [--world 4]"""
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


def run(n=20000, steps=200, world=4, every=10, seed=7, graphs=False, say=print):
    """n particles with births on, `steps` free-running steps: one context and `world` slabs against the oracle, every byte
    every `every` steps.  graphs: the slabs replay their stage sequences as hipGraphs.  Raises on the first difference."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-39.9, 39.9, (n, 3)).astype(np.float32)
    age = rng.uniform(2.0, 9.0, n).astype(np.float32)
    fert = rng.uniform(3.0, 12.0, n).astype(np.float32)
    v = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
    mk = lambda r, W: ps.ParticleSystem(ps.default_config(rank=r, world=W, flags=ps.FLAG_EXPLOSIONS, seed=seed))
    one = mk(0, 1)
    ranks = [mk(r, world) for r in range(world)] if world > 1 else []
    if graphs:
        for g in ranks:
            g.set_graphs(True)
    o = O.System(oracle_cfg_from(one.cfg))
    o.set_rng(explosion_rng(seed))
    ids = o.fill(xyz, age=age, fert_age=fert)
    p = o.particles
    p["vx"][ids], p["vy"][ids], p["vz"][ids] = v.T
    for g in [one] + ranks:
        g.fill_particles(xyz, age=age, fert_age=fert, vxyz=v)
    t0 = time.time()
    for k in range(1, steps + 1):
        one.step(1)
        if ranks:
            step_local(ranks)
        o.step(1)
        if k % every == 0 or k == steps:
            assert_same_particles(one.download_particles(), o.particles, "one context, step %d" % k)
            if ranks:
                plans = [g.slab_plan() for g in ranks]
                assert_same_particles(merge_owned([g.download_particles() for g in ranks], plans), o.particles, "%d slabs, step %d" % (world, k))
            live = int((o.particles["cell"] >= 0).sum())
            say("step %d: live %d, births %d, relocations %d, collisions %d, %.0f s: equal" %
                (k, live, o.counters["births"], o.counters["relocations"], o.counters["deaths_collision"], time.time() - t0))
    out = dict(live=int((o.particles["cell"] >= 0).sum()), births=int(o.counters["births"]), relocations=int(o.counters["relocations"]),
               graph_replays=sum(g.graph_stats()[0] for g in ranks) if graphs else 0)
    for g in [one] + ranks:
        g.close()
    o.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--world", type=int, default=4)
    ap.add_argument("--every", type=int, default=10)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--graphs", action="store_true")
    a = ap.parse_args()
    run(a.n, a.steps, a.world, a.every, a.seed, a.graphs, say=lambda m: print(m, flush=True))
    print("long parity done: %d steps, 0 mismatches" % a.steps)


if __name__ == "__main__":
    main()
