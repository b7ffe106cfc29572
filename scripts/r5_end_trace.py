"""Diagnostic: when every wave of the balanced force pass ends (one GPU, the benchmark cloud).
Build:  hipcc ... -DPSAMD_END_TRACE  (scripts/r5_end_trace.sh);  run with PSAMD_LIB pointing at that build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
import particlesystem_amd as ps
n = 1 << 20
g = ps.ParticleSystem(ps.default_config())
g.set_tdata_mirror(False)
xyz = g.uniform_cloud(n, 2026)
age = np.random.default_rng(2026).uniform(15 / 7, 7.5, n).astype(np.float32)
g.fill_particles(xyz, age=age, fert_age=np.full(n, 1e6, np.float32))
g.snapshot_save()
g.set_timing(True, period=1)
for _ in range(30):
    g.snapshot_restore(); g.step(1)
g.synchronize()
tim, _ = g.timing()
med, mx, _ = g.timing_stats()
nw = 7168 + 4 * 256                        # more than the launch holds
out = np.zeros(nw, np.uint64)
g._ck(g.lib.psamd_debug_wave_trace(g.h, out.ctypes.data_as(C.c_void_p), nw))
t = out[out > 0].astype(np.float64) / 100.0   # us (100 MHz counter)
t = t[t > t.max() - 5000.0]                   # this launch's (older entries of slots no longer used fall out)
end = t - t.min()
dur = med["pairs"]
print("waves with an end time:", len(t), " force pass (median, us):", dur)
print("end time relative to the first wave to end, us: percentiles 1/10/25/50/75/90/99/100:",
      " ".join("%.0f" % np.percentile(end, q) for q in (1, 10, 25, 50, 75, 90, 99, 100)))
last = end.max()
print("idle wave-slots at the end: sum(last - end) / (waves x pass) = %.2f %% of the pass's wave-time" % (100.0 * (last - end).sum() / (len(end) * dur)))
for cut in (10, 20, 50, 100, 200):
    print("  waves that ended more than %3d us before the last: %5d (%.1f %%)" % (cut, (end < last - cut).sum(), 100.0 * (end < last - cut).mean()))
# structure: by launch order (workgroup b: pack workgroups first, then the balanced part XCD by XCD), by XCD, by age rank
idx = np.nonzero(out > 0)[0]
tt = out[idx].astype(np.float64) / 100.0
keep = tt > tt.max() - 5000.0
idx, tt = idx[keep], tt[keep]
e = tt - tt.min()
blk = idx // 4
nb = blk.max() + 1
print("workgroups in the launch:", nb)
for lo in range(0, nb, 128):
    m = (blk >= lo) & (blk < lo + 128)
    if m.any():
        print("  workgroups %4d-%4d: end us min %5.0f p50 %5.0f max %5.0f" % (lo, lo + 127, e[m].min(), np.median(e[m]), e[m].max()))
for x in range(8):
    m = (blk % 8) == x
    print("  XCD %d: p10 %5.0f p50 %5.0f p90 %5.0f max %5.0f" % (x, np.percentile(e[m], 10), np.median(e[m]), np.percentile(e[m], 90), e[m].max()))
