"""Diagnostic: where k_replay_bucket spends its time, per queue record.
Build first with PSAMD_EXTRA_FLAGS=-DPSAMD_REPLAY_TRACE python particlesystem_amd/build.py --force"""
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
    g.snapshot_restore(); g.step(1)
g.synchronize()
nrec = g.sizes.queue_info_size
t = g.wave_trace()
t = np.asarray(t).reshape(-1)[: 8 * nrec].reshape(nrec, 8).astype(np.int64)
t = t[t[:, 6] > 0]
names = ["load", "sort", "copy+prefix", "bad-check", "replay", "tail"]
d = np.diff(t[:, :6], axis=1) / 100.0
print("records with ops", len(t), "ops: p50 %d max %d" % (np.median(t[:, 6]), t[:, 6].max()))
for i, nm in enumerate(names[:5]):
    print("%-12s us: p50 %.1f p99 %.1f max %.1f" % (nm, np.median(d[:, i]), np.percentile(d[:, i], 99), d[:, i].max()))
tot = (t[:, 5] - t[:, 0]) / 100.0
print("whole WG us: p50 %.1f max %.1f; kernel span %.1f us" % (np.median(tot), tot.max(), (t[:, 5].max() - t[:, 0].min()) / 100.0))
start = (t[:, 0] - t[:, 0].min()) / 100.0
print("WG start us: p50 %.1f p99 %.1f max %.1f" % (np.median(start), np.percentile(start, 99), start.max()))
