"""Diagnostic: per-wave timeline of the pair kernel (single GPU; argv kept for old notes).
Build first with PSAMD_EXTRA_FLAGS=-DPSAMD_WAVE_TRACE python particlesystem_amd/build.py --force"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import particlesystem_amd as ps
world, rank = int(sys.argv[1]), int(sys.argv[2])
n = 1 << 20
g = ps.ParticleSystem(ps.default_config())
xyz = g.uniform_cloud(n, 2026)
age = np.random.default_rng(2026).uniform(15 / 7, 7.5, n).astype(np.float32)
g.fill_particles(xyz, age=age, fert_age=np.full(n, 1e6, np.float32))
g.snapshot_save()
for _ in range(3):
    g.snapshot_restore(); g.init_iframe(); g.build_grid()
    g.calc_forces_pairs(); g.calc_forces_apply()
g.synchronize()
t = g.wave_trace()
t = t[t[:, 1] > 0]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0   # us
dur = (t[:, 1] - t[:, 0]) / 100.0
hw = t[:, 2] & 0xffffffff
xcc = (t[:, 2] >> 32) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3
where = (xcc.astype(np.int64) << 12) | (se.astype(np.int64) << 8) | (sh.astype(np.int64) << 6) | (cu.astype(np.int64) << 2) | simd.astype(np.int64)
print("waves", len(t), "kernel span us %.1f" % ((t[:, 1].max() - t0) / 100.0))
print("start us: min %.1f p50 %.1f p99 %.1f max %.1f" % (start.min(), np.median(start), np.percentile(start, 99), start.max()))
print("dur us: min %.1f p50 %.1f p99 %.1f max %.1f" % (dur.min(), np.median(dur), np.percentile(dur, 99), dur.max()))
u, cnt = np.unique(where, return_counts=True)
print("distinct SIMDs used", len(u), "waves per SIMD: min %d p50 %d max %d" % (cnt.min(), np.median(cnt), cnt.max()))
print("distinct CUs", len(np.unique(where >> 2)), "XCCs", np.unique(xcc))
end = (t[:, 1] - t0) / 100.0
for x in np.unique(xcc):
    m = xcc == x
    print("XCC %d: waves %d  busy wave-us %.0f  last end %.1f us  p50 dur %.1f" % (x, m.sum(), dur[m].sum(), end[m].max(), np.median(dur[m])))
work = dur
u2, inv = np.unique(where, return_inverse=True)
per_simd = np.bincount(inv, weights=work)
last_end = np.zeros(len(u2)); np.maximum.at(last_end, inv, end)
print("work per SIMD wave-us: min %.0f p50 %.0f max %.0f; last end per SIMD: min %.0f p50 %.0f max %.0f" % (
    per_simd.min(), np.median(per_simd), per_simd.max(), last_end.min(), np.median(last_end), last_end.max()))
# round 4: where the kernel's idle issue slots are -- the ramp at the start, the tail at the end
span = (t[:, 1].max() - t0) / 100.0
for name, v in (("start", start), ("end", end)):
    print(name, "us percentiles 1/10/50/90/99/100:", " ".join("%.0f" % np.percentile(v, q) for q in (1, 10, 50, 90, 99, 100)))
# resident waves over time (how many waves are alive at time x), in 20 slices of the span
edges = np.linspace(0, span, 21)
alive = [(np.minimum(end, b) - np.maximum(start, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
print("mean resident waves per slice of the span:", " ".join("%.0f" % a for a in alive))
print("SIMD-slots idle at the end: sum over SIMDs of (span - last end) = %.0f wave-us of %.0f (%.1f %%)" % (
    (span - last_end).sum(), span * len(u2), 100 * (span - last_end).sum() / (span * len(u2))))
# long and short waves
order = np.argsort(dur)
print("shortest 5 durations", dur[order[:5]], "longest 5", dur[order[-5:]])
# end time against dispatch order: block b of the balanced part is the (b // 256)-th oldest workgroup on its CU
tr = g.wave_trace()
nblk = len(tr) // 4
blk = np.repeat(np.arange(nblk), 4)[:len(tr)]
ok = tr[:, 1] > 0
t00 = tr[ok, 0].min()
e = (tr[:, 1] - t00) / 100.0
s0 = (tr[:, 0] - t00) / 100.0
live_blocks = np.unique(blk[ok])
print("blocks with a trace:", len(live_blocks), "first", live_blocks[:3], "last", live_blocks[-3:])
nb_total = live_blocks.max() + 1
nmb = nb_total - 1792 if nb_total > 1792 else 0
print("pack workgroups (first %d blocks): waves %d, end p50 %.0f max %.0f" % (nmb, (ok & (blk < nmb)).sum(), np.median(e[ok & (blk < nmb)]) if (ok & (blk < nmb)).any() else 0, e[ok & (blk < nmb)].max() if (ok & (blk < nmb)).any() else 0))
for a in range(7):
    m = ok & (blk >= nmb + 256 * a) & (blk < nmb + 256 * (a + 1))
    if m.any():
        print("age rank %d: waves %d  start p50 %.0f  end p10 %.0f p50 %.0f p90 %.0f max %.0f" % (a, m.sum(), np.median(s0[m]), np.percentile(e[m], 10), np.median(e[m]), np.percentile(e[m], 90), e[m].max()))
