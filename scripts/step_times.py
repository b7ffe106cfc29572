"""Diagnostic: wall time of every step of a short timed region right after a synchronisation (where do the first steps lose time?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesystem_amd as ps
n = 1 << 20
g = ps.ParticleSystem(ps.default_config())
xyz = g.uniform_cloud(n, 2026)
age = np.random.default_rng(2026).uniform(15 / 7, 7.5, n).astype(np.float32)
g.fill_particles(xyz, age=age, fert_age=np.full(n, 1e6, np.float32))
g.snapshot_save()
def one():
    g.snapshot_restore(); g.step(1)
for _ in range(300): one()
g.synchronize()
for trial, idle in enumerate((0.0, 0.0, 0.002, 0.05)):
    for _ in range(20): one()
    g.synchronize()
    if idle: time.sleep(idle)
    t = [time.perf_counter()]
    for _ in range(24):
        one(); t.append(time.perf_counter())
    g.synchronize(); t.append(time.perf_counter())
    d = np.diff(np.array(t)) * 1e3
    print("idle %.3f s before: total %.3f ms for 24 steps = %.4f per step; step walls:" % (idle, (t[-1] - t[0]) * 1e3, (t[-1] - t[0]) * 1e3 / 24), " ".join("%.2f" % x for x in d))
