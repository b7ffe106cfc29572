import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesystem_amd as ps
n = 1 << 20
g = ps.ParticleSystem(ps.default_config(collision_radius=0.0))
xyz = g.uniform_cloud(n, 2026)
rng = np.random.default_rng(2026)
age = rng.uniform(15/7, 7.5, n).astype(np.float32)
g.fill_particles(xyz, age=age, fert_age=np.full(n, 1e6, np.float32))
prev = 0
g.set_timing(True)
for k in range(6):
    t0 = time.perf_counter()
    try:
        g.step(1); g.synchronize()
        err = None
    except ps.PsamdError as e:
        err = str(e)
    dt = time.perf_counter() - t0
    c = g.counters
    g.init_iframe(); g.build_grid()
    cg = g.download_cellgrid()
    p = g.download_particles()
    live = p[p["cell"] >= 0]
    sp = np.sqrt(live["vx"]**2 + live["vy"]**2 + live["vz"]**2)
    qi, q = g.download_queues()
    print("step", k, "ms %.2f" % (dt*1e3), "reloc", c["relocations"] - prev, "lost", c["relocations_lost"], "live", len(live),
          "maxcell", cg[:, 0].max(), "mean|v| %.2f" % sp.mean(), "min queue count", qi["count"].min(), "err", err, flush=True)
    prev = c["relocations"]
    tim, nl = g.timing(); print("   us:", {k: round(v / max(nl, 1), 1) for k, v in tim.items()}); g.set_timing(True)
