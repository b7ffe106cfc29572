"""Randomised campaign for the all-pairs forces (PSAMD_FLAG_ALL_PAIRS; not in the reference, parity unpinned): nothing
can say what the sums should be to the bit, but the union of W slabs must equal ONE context byte for byte, every step,
whatever the cloud, the grid and the world size (the far field's association depends on the global cell order only).
The clouds are fuzz_parity.py's draws (densities, faces, kids and elders, births, masses, odd grids).
usage: python scripts/fuzz_allpairs.py [--cases 30] [--seed 1] [--log gpurun_out/fuzz_allpairs.log]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, os.path.abspath(d))
import particlesystem_amd as ps           # noqa: E402
from particlesystem_amd.slab import merge_owned, step_local   # noqa: E402
from util import assert_same_particles    # noqa: E402
from fuzz_parity import draw_case         # noqa: E402


def run_case(c, seed):
    flags = ps.FLAG_ALL_PAIRS | (ps.FLAG_EXPLOSIONS if c["births"] else 0)
    extra = dict(seed=seed) if c["births"] else {}
    over = dict(c["over"])
    if over.get("collision_radius", 0.4) > 0.5:          # all-pairs contexts need the two-pass pair stage (radius small against the cell)
        over["collision_radius"] = 0.4
    W = max(2, c["world"])
    try:
        one = ps.ParticleSystem(ps.default_config(flags=flags, **extra, **over))
        ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=W, flags=flags, **extra, **over)) for r in range(W)]
    except ps.PsamdError as e:
        return "refused at creation (%s)" % str(e)[:70]
    everyone = [one] + ranks
    try:
        for g in everyone:
            g.fill_particles(c["xyz"], age=c["age"], fert_age=c["fert"], vxyz=c["v"], w=c["w"])
    except ps.PsamdError as e:
        for g in everyone:
            g.close()
        return "skipped (%s)" % str(e)[:60]
    plans = [g.slab_plan() for g in ranks]
    try:
        for k in range(c["steps"]):
            one.step(1)
            step_local(ranks, overlap_interior=c["interior"])
            union = merge_owned([g.download_particles() for g in ranks], plans)
            assert_same_particles(union, one.download_particles(), "all-pairs, %d slabs, step %d" % (W, k + 1))
    except ps.PsamdError as e:
        if "status message" in str(e) or "had no room" in str(e):
            for g in everyone:
                g.close()
            return "refused (%s)" % str(e)[:70]
        raise
    cnt = {k: one.counters[k] for k in ("relocations", "births", "deaths_collision", "integrated")}
    for g in everyone:
        g.close()
    return "ok %r" % cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default=None)
    ap.add_argument("--sizes", default="2000,6000,20000")
    ap.add_argument("--max-steps", type=int, default=4)
    ap.add_argument("--worlds", default="2,3,4,5,8")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    log = open(a.log, "a") if a.log else sys.stdout
    bad = 0
    for i in range(a.cases):
        c = draw_case(rng, [int(v) for v in a.sizes.split(",")], a.max_steps, [int(v) for v in a.worlds.split(",")])
        if c["v"] is not None and bool(np.isnan(c["v"]).any()):
            c["v"] = np.nan_to_num(c["v"], nan=0.0)        # (a force that is no number is no test of an association)
        t0 = time.time()
        try:
            res = run_case(c, a.seed * 1000 + i)
        except AssertionError as e:
            res = "MISMATCH %s" % str(e)[:200]
            bad += 1
        print("case %d [%s] %.1fs: %s" % (i, c["desc"], time.time() - t0, res), file=log, flush=True)
    print("all-pairs fuzz done: %d cases, %d mismatches" % (a.cases, bad), file=log, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
