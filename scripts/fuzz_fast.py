"""Randomised check of the tolerance mode (PSAMD_FLAG_FAST_MATH) against the exact mode on the
same GPU: same random cases as scripts/fuzz_parity.py (one context), one frame each; collision
flags must be identical, the accelerations of the particles that get one are compared relative to
their norm.  usage: python scripts/fuzz_fast.py [--cases 40] [--seed 1] [--log file]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_parity import draw_case, ps          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default=None)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    log = open(a.log, "a") if a.log else sys.stdout
    worst, bad = 0.0, 0
    for i in range(a.cases):
        c = draw_case(rng, [3000, 12000, 40000, 90000])
        over = {k: v for k, v in c["over"].items()}
        res = []
        try:
            for flags in (0, ps.FLAG_FAST_MATH):
                g = ps.ParticleSystem(ps.default_config(flags=flags, **over))
                g.fill_particles(c["xyz"], age=c["age"], fert_age=np.float32(1e6), vxyz=c["v"], w=c["w"])
                g.init_iframe(); g.build_grid(); g.calc_forces_pairs()
                total = int(g.download_cellgrid()[:, 0].sum())
                res.append(g.download_force4(0, total))
                g.calc_forces_apply()
                g.close()
        except ps.PsamdError as e:
            print("case %d [%s]: skipped (%s)" % (i, c["desc"], str(e)[:60]), file=log, flush=True)
            continue
        ex, fa = res
        same_flags = np.array_equal(ex[:, 3].view(np.int32), fa[:, 3].view(np.int32))
        keep = ex[:, 3].view(np.int32) == 0
        x, y = ex[keep, :3].astype(np.float64), fa[keep, :3].astype(np.float64)
        nrm = np.linalg.norm(x, axis=1)
        rel = np.linalg.norm(x - y, axis=1) / np.maximum(nrm, 1e-30)
        rel = rel[nrm > 0]
        mx = float(rel.max()) if len(rel) else 0.0
        worst = max(worst, mx)
        if not same_flags:
            bad += 1
        print("case %d [%s]: flags %s, %d accelerations, max relative deviation %.3g, 99.9th percentile %.3g" %
              (i, c["desc"], "identical" if same_flags else "DIFFER", len(rel), mx, float(np.quantile(rel, 0.999)) if len(rel) else 0.0),
              file=log, flush=True)
    print("fuzz-fast done: %d cases, %d with different flags, worst relative deviation %.3g" % (a.cases, bad, worst), file=log, flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
