"""One GPU, N = 2^20, restored every step: psamd_step(1) with the host a step ahead (run-ahead 1, the default) and with every
stage call waiting for its own step's scalars (run-ahead 0: the reference's "stop in the step that failed")."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesystem_amd as ps
n = 1 << 20
g = ps.ParticleSystem(ps.default_config())
g.set_tdata_mirror(False)
xyz = g.uniform_cloud(n, 2026)
age = np.random.default_rng(2026).uniform(15 / 7, 7.5, n).astype(np.float32)
g.fill_particles(xyz, age=age, fert_age=np.full(n, 1e6, np.float32))
g.snapshot_save()
for rep in range(2):
    for ra in (1, 0):
        g.set_run_ahead(ra)
        for _ in range(60):
            g.snapshot_restore(); g.step(1)
        g.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            g.snapshot_restore(); g.step(1)
        g.synchronize()
        print("run-ahead %d: %.4f ms per step" % (ra, 1e3 * (time.perf_counter() - t0) / 200))
