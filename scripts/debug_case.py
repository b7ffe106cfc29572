"""Re-run one case of a fuzz_parity campaign (same draws) and, at the first step whose state differs from the
oracle's, print what differs.  usage: python scripts/debug_case.py --seed S --case K [--sizes ..] [--worlds ..] [--world W] [--interior 0/1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from fuzz_parity import draw_case, ps, O, explosion_rng, oracle_cfg_from, merge_owned, step_local   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, required=True)
    ap.add_argument("--case", type=int, required=True)
    ap.add_argument("--sizes", default="3000,12000,40000,90000")
    ap.add_argument("--worlds", default="1,1,2,3,4")
    ap.add_argument("--max-steps", type=int, default=6)
    ap.add_argument("--world", type=int, default=0, help="override the case's world size")
    ap.add_argument("--interior", type=int, default=-1, help="override the interior-pass switch")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    sizes = [int(v) for v in a.sizes.split(",")]
    worlds = [int(v) for v in a.worlds.split(",")]
    for _ in range(a.case + 1):
        c = draw_case(rng, sizes, a.max_steps, worlds)
    if a.world:
        c["world"] = a.world
    if a.interior >= 0:
        c["interior"] = bool(a.interior)
    print("case", c["desc"], "steps", c["steps"], "-> world", c["world"], "interior", c["interior"], flush=True)
    seed = 1000 + a.case
    W = c["world"]
    flags = ps.FLAG_EXPLOSIONS if c["births"] else 0
    extra = dict(seed=seed) if c["births"] else {}
    if c["cuts"]:
        extra["cuts"] = c["cuts"]
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=W, flags=flags, **extra, **c["over"])) for r in range(W)]
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    if c["births"]:
        o.set_rng(explosion_rng(seed))
    ids_o = o.fill(c["xyz"], age=c["age"], fert_age=c["fert"], w=c["w"])
    if c["v"] is not None:
        p = o.particles
        p["vx"][ids_o], p["vy"][ids_o], p["vz"][ids_o] = c["v"].T
    for g in ranks:
        g.fill_particles(c["xyz"], age=c["age"], fert_age=c["fert"], vxyz=c["v"], w=c["w"])
    for r, g in enumerate(ranks):
        pl = g.slab_plan()
        print("rank", r, "compute layers", pl.cut_lo, pl.cut_hi, "state", pl.state_lo, pl.state_hi, flush=True)
    for k in range(c["steps"]):
        if W == 1:
            ranks[0].step(1)
        else:
            step_local(ranks, overlap_interior=c["interior"])
            for g in ranks:
                g.synchronize()
        o.step(1)
        plans = [g.slab_plan() for g in ranks]
        got = ranks[0].download_particles() if W == 1 else merge_owned([g.download_particles() for g in ranks], plans)
        want = o.particles
        cnt = {kk: sum(g.counters[kk] for g in ranks) for kk in ("relocations", "relocations_lost", "births", "births_failed", "deaths_collision", "deaths_age", "cell_overflow_kills", "survives", "integrated")}
        print("step", k + 1, "gpu", cnt, flush=True)
        print("step", k + 1, "ora", {kk: o.counters[kk] for kk in cnt}, flush=True)
        livew = want["cell"] >= 0
        print("   oracle: live", int(livew.sum()), "max|a|", float(np.abs(want["ax"][livew]).max()), "in cell 0:", int((want["cell"] == 0).sum()),
              "min pair dist proxy: particles with |a|>1e6:", int((np.abs(want["ax"][livew]) > 1e6).sum()), flush=True)
        bad = [f for f in got.dtype.names if got[f].tobytes() != want[f].tobytes()]
        if bad:
            print("fields that differ:", bad)
            d = np.nonzero(got["cell"] != want["cell"])[0]
            print(len(d), "slots differ in cell; live gpu", int((got["cell"] >= 0).sum()), "oracle", int((want["cell"] >= 0).sum()))
            for s in d[:40]:
                print("  slot", int(s), "gpu cell", int(got["cell"][s]), "id", int(got["id"][s]) if "id" in got.dtype.names else "", "| oracle cell", int(want["cell"][s]),
                      "x", float(want["x"][s]), float(want["y"][s]), float(want["z"][s]), "age", float(want["age"][s]))
            for s in d[-6:]:
                print("  GPU slot", int(s), {f: (float(got[f][s]) if got[f].dtype.kind == "f" else int(got[f][s])) for f in got.dtype.names})
                print("  ORA slot", int(s), {f: (float(want[f][s]) if want[f].dtype.kind == "f" else int(want[f][s])) for f in want.dtype.names})
            for name, arr in (("gpu", got), ("oracle", want)):
                live = arr["cell"] >= 0
                for f in ("x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az", "w", "age"):
                    v = arr[f][live]
                    nf = ~np.isfinite(v)
                    if nf.any():
                        print("  non-finite", name, f, int(nf.sum()), "slots", np.nonzero(live)[0][nf][:8])
                print("  ", name, "max |a|", float(np.nanmax(np.abs(arr["ax"][live]))), "max |x|", float(np.nanmax(np.abs(arr["x"][live]))))
            qs = [g.download_queues() for g in ranks]
            qi = qs[0][0] if W == 1 else merge_owned([q[0] for q in qs], plans, "records")
            oq = o.queue_info
            dq = [i for i in range(len(qi)) if qi[i].tobytes() != oq[i].tobytes()]
            print(len(dq), "queue records differ:", dq[:20])
            for i in dq[:10]:
                print("  record", i, "gpu", qi[i], "oracle", oq[i])
            return 1
    print("no mismatch")
    return 0


if __name__ == "__main__":
    sys.exit(main())
