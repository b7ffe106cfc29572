"""One GPU: the same step enqueued as psamd_step(1), as the four slab stage calls (what the C++ host issues), as the three
reference stage calls -- does the way the host cuts the step matter to the GPU's time?  (N = 2^20, restored every step.)"""
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
def fused(): g.step(1)
def slab(): g.slab_build(); g.slab_pairs(); g.slab_apply(); g.slab_finish()
def stages(): g.init_iframe(); g.build_grid(); g.calc_forces()
for rep in range(2):
    for name, fn in (("psamd_step(1)", fused), ("slab_build/pairs/apply/finish", slab), ("init_iframe/build_grid/calc_forces", stages)):
        for _ in range(60):
            g.snapshot_restore(); fn()
        g.synchronize()
        g.set_timing(True, period=8)
        t0 = time.perf_counter()
        for _ in range(200):
            g.snapshot_restore(); fn()
        g.synchronize()
        dt = (time.perf_counter() - t0) / 200
        med, _, _ = g.timing_stats()
        g.set_timing(False)
        print("%-36s %.4f ms per step, force pass %.1f us" % (name, 1e3 * dt, med["pairs"]))
