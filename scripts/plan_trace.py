"""Diagnostic: where k_plan_force spends its time, per workgroup.
Build first with PSAMD_EXTRA_FLAGS=-DPSAMD_PLAN_TRACE python particlesystem_amd/build.py --force"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (its HIP runtime first)
import particlesystem_amd as ps
n = 1 << 20
g = ps.ParticleSystem(ps.default_config())
xyz = g.uniform_cloud(n, 2026)
age = np.random.default_rng(2026).uniform(15 / 7, 7.5, n).astype(np.float32)
g.fill_particles(xyz, age=age, fert_age=np.full(n, 1e6, np.float32))
g.snapshot_save()
for _ in range(3):
    g.snapshot_restore(); g.init_iframe(); g.build_grid(); g.calc_forces_pairs(); g.synchronize()
    t = np.asarray(g.wave_trace()).reshape(-1)[:64].reshape(8, 8).astype(np.int64)
    g.calc_forces_apply()
names = ["fill LDS", "prefix", "lists+packs", "run bounds", "split"]
d = np.diff(t[:, :6], axis=1) / 100.0
for i, nm in enumerate(names):
    print("%-12s us per workgroup: %s" % (nm, " ".join("%.1f" % v for v in d[:, i])))
print("whole: %s; span %.1f us" % (" ".join("%.1f" % v for v in (t[:, 5] - t[:, 0]) / 100.0), (t[:, 5].max() - t[:, 0].min()) / 100.0))
