"""Which cases of a fuzz campaign can run at all -- decided on the CPU: the oracle's own fill accepts the cloud (a clump
may fill a segment) and the slab partition admits the plan.  Used to pick the fixed-seed slice tests/test_gpu_fuzz.py runs.
usage: python scripts/fuzz_screen.py --seed S --cases K [--sizes ...] [--worlds ...] [--max-steps M] [--legacy]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, os.path.abspath(d))
import oracle_py as O                     # noqa: E402
import particlesystem_amd as ps           # noqa: E402
from fuzz_parity import draw_case          # noqa: E402
from util import oracle_cfg_from           # noqa: E402


def runnable(c):
    W = c["world"]
    extra = {"cuts": c["cuts"]} if c["cuts"] else {}
    try:
        cfgs = [ps.default_config(rank=r, world=W, **extra, **c["over"]) for r in range(W)]
        for cfg in cfgs:
            ps.slab_plan(cfg)
    except ps.PsamdError:
        return "plan refused"
    o = O.System(oracle_cfg_from(cfgs[0]))
    try:
        o.fill(c["xyz"], age=c["age"], fert_age=c["fert"], w=c["w"])
    except Exception as e:
        return "fill refused (%s)" % str(e)[:40]
    finally:
        o.close()
    return "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--sizes", default="3000,12000,40000,90000")
    ap.add_argument("--worlds", default="1,1,2,3,4")
    ap.add_argument("--max-steps", type=int, default=6)
    ap.add_argument("--legacy", action="store_true", help="draws as before the not-a-number particles were added")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    for i in range(a.cases):
        c = draw_case(rng, [int(v) for v in a.sizes.split(",")], a.max_steps, [int(v) for v in a.worlds.split(",")], nan_draw=not a.legacy)
        print("%d case %d steps=%d [%s]: %s" % (a.seed, i, c["steps"], c["desc"], runnable(c)), flush=True)


if __name__ == "__main__":
    main()
