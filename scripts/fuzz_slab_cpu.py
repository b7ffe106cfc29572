"""CPU-only twin of scripts/fuzz_parity.py for the slab protocol: the oracle-built stand-in rank
(tests/oracle_slab.py: holds only its slab, speaks the product's stage / message interface) in
worlds of 2..4 against the serial oracle, random cases from the same generator.  No GPU.
usage: python scripts/fuzz_slab_cpu.py [--cases 30] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for d in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.dirname(os.path.abspath(__file__))):
    sys.path.insert(0, os.path.abspath(d))
import oracle_py as O                      # noqa: E402
import particlesystem_amd as ps            # noqa: E402
from oracle_slab import OracleSlabRank     # noqa: E402
from particlesystem_amd.slab import merge_owned, step_local   # noqa: E402
from util import oracle_cfg_from           # noqa: E402

sys.argv_backup = sys.argv
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py")).read()
ns = {"np": np}
exec(src[src.index("def draw_case"):src.index("def run_case")], ns)     # the generator only (no GPU imports)
draw_case = ns["draw_case"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=30)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = ran = 0
    for i in range(a.cases):
        c = draw_case(rng, [1500, 3000, 6000])
        if c["world"] == 1 or c["births"] or c["w"] is not None:
            continue                       # (the stand-in's fill takes positions, ages, fertility ages)
        t = time.time()
        over = dict(c["over"])
        if c["cuts"]:
            over["cuts"] = c["cuts"]
        W = c["world"]
        try:
            ranks = [OracleSlabRank(ps.default_config(rank=r, world=W, **over), xfer_cap=1 << 16) for r in range(W)]
        except ps.PsamdError:
            continue
        ref = O.System(oracle_cfg_from(ranks[0].cfg))
        try:
            ids = ref.fill(c["xyz"], age=c["age"], fert_age=c["fert"])
        except Exception:
            continue
        for s in ranks:
            s.fill_particles(c["xyz"], c["age"], c["fert"])
        if c["v"] is not None:
            p = ref.particles
            p["vx"][ids], p["vy"][ids], p["vz"][ids] = c["v"].T
            for s in ranks:
                pp = s.o.particles
                mine = pp["cell"][ids] >= 0      # the same slots on the rank that owns them
                for k, f in enumerate(("vx", "vy", "vz")):
                    pp[f][ids[mine]] = c["v"][mine, k]
        ok = True
        for k in range(c["steps"]):
            step_local(ranks)
            ref.step(1)
            plans = [s.plan for s in ranks]
            got = merge_owned([s.download_particles() for s in ranks], plans)
            if got.tobytes() != ref.particles.tobytes():
                ok = False
                break
        ran += 1
        bad += 0 if ok else 1
        print("case %d [%s] %.1fs: %s" % (i, c["desc"], time.time() - t, "ok" if ok else "MISMATCH at step %d" % (k + 1)), flush=True)
    print("fuzz-slab-cpu done: %d cases run, %d mismatches" % (ran, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
