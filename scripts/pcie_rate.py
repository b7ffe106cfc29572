"""Step rate when the caller fetches the reference's buffers back to the host after the
stages, as the reference's driver does with pFetchBack (ps.cpp:1874-1922)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesystem_amd as ps
n = 1 << 20
g = ps.ParticleSystem(ps.default_config())
xyz = g.uniform_cloud(n, 2026)
age = np.random.default_rng(2026).uniform(15 / 7, 7.5, n).astype(np.float32)
g.fill_particles(xyz, age=age, fert_age=np.full(n, 1e6, np.float32))
g.snapshot_save()
def run(fetch, steps=5):
    g.snapshot_restore(); g.step(1); g.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.snapshot_restore()
        g.init_iframe(); g.build_grid()
        if fetch: g.gridmax(); g.download_tdata()
        g.calc_forces()
        if fetch: g.download_particles(); g.download_queues()
    g.synchronize()
    return (time.perf_counter() - t0) / steps
for fetch in (False, True, True):
    dt = run(fetch)
    print("fetch_back=%s: %.2f ms/step, %.3g particle-updates/s" % (fetch, dt * 1e3, n / dt))
