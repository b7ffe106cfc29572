"""BASELINE.json configs[3] and configs[4] on the GPU (configs[0] and [2]: test_gpu_parity,
test_gpu_fullsize; configs[1], all-pairs: test_gpu_allpairs).  At these sizes the oracle cannot
run whole steps in seconds, so: size-independent properties (sorted partition, determinism,
slab invariance) plus windows of the sorted order against the oracle's pair pass, bit for bit."""
import hashlib

import numpy as np
import pytest

import oracle_py as O
import particlesystem_amd as ps
from particlesystem_amd.slab import merge_owned, step_local
from util import oracle_cfg_from

pytestmark = pytest.mark.gpu


def digest(p, qi, q):
    h = hashlib.sha256()
    h.update(p.tobytes()); h.update(qi.tobytes()); h.update(q.tobytes())
    return h.hexdigest()


def windows_match_oracle(g, o, windows, min_free=500):
    """force4 of the GPU's pair pass vs the oracle's on [lo, hi) windows of the sorted order"""
    total = o.sorted_count()
    f = np.zeros((total, 4), np.float32)
    for lo, hi in windows:
        o.calc_pairs(lo, hi, f)
        got = g.download_force4(lo, hi - lo).view(np.uint32)
        want = f[lo:hi].view(np.uint32)
        assert np.array_equal(got[:, 3], want[:, 3]), "collision flags differ at %d" % lo
        keep = want[:, 3] == 0                       # a flagged particle's force is never looked at
        assert keep.sum() >= min_free and np.array_equal(got[keep, :3], want[keep, :3]), lo


@pytest.mark.timeout(1500)
def test_config3_n22_two_slabs_on_one_gpu():
    """configs[3]: N = 2^22 (1024 per cell in the default 16^3 box).  One context vs two slabs
    of the same system: identical union after a whole step; windows of the pair pass against
    the oracle (each particle there sums ~27600 terms)."""
    n = 1 << 22
    over = dict(max_particles_num=n)
    g = ps.ParticleSystem(ps.default_config(**over))
    xyz = g.uniform_cloud(n, 11)
    rng = np.random.default_rng(11)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = np.full(n, 1e6, np.float32)
    ids = g.fill_particles(xyz, age=age, fert_age=fert)
    g.snapshot_save()
    g.init_iframe(); g.build_grid()
    cg = g.download_cellgrid()
    assert cg[:, 0].sum() == n and cg[:, 0].max() <= g.sizes.max_per_cell
    g.calc_forces_pairs()
    o = O.System(oracle_cfg_from(g.cfg))
    assert np.array_equal(o.fill(xyz, age=age, fert_age=fert), ids)
    o.init_iframe(); o.build_grid()
    assert np.array_equal(cg, o.cellgrid)
    total = o.sorted_count()
    windows_match_oracle(g, o, [(0, 1500), (total // 2 + 333, total // 2 + 1833), (total - 1500, total)], min_free=50)
    o.close()
    g.calc_forces_apply()
    whole = (g.download_particles(), *g.download_queues())
    c1 = g.counters
    g.snapshot_restore(); g.step(1)
    assert digest(*whole) == digest(g.download_particles(), *g.download_queues())          # deterministic
    g.close()
    halves = [ps.ParticleSystem(ps.default_config(rank=r, world=2, **over)) for r in range(2)]
    for h in halves:
        h.fill_particles(xyz, age=age, fert_age=fert)
    step_local(halves)
    plans = [h.slab_plan() for h in halves]
    qs = [h.download_queues() for h in halves]
    union = (merge_owned([h.download_particles() for h in halves], plans), merge_owned([q[0] for q in qs], plans, "records"),
             merge_owned([q[1] for q in qs], plans))
    assert digest(*union) == digest(*whole)
    assert sum(h.counters["relocations"] for h in halves) == c1["relocations"] > 0
    for h in halves:
        h.close()


@pytest.mark.timeout(1500)
def test_config3_literally_n22_all_pairs_eight_slabs_snapshot_all_gather():
    """configs[3] as BASELINE words it: N = 2^22 sharded over EIGHT slabs with an all-gather of the positions once
    per step (PSAMD_FLAG_ALL_PAIRS: every rank contributes the snapshot of its own cells, walks the gathered buffer).
    One whole step; the union of the eight slabs must be the one-context step byte for byte (particles, QUEUE_INFO,
    queues), and -- no CPU re-does 1.8e13 pairs -- three windows of 200 particles of the one-context pair pass are
    held against an fp64 direct sum over all 2^22 bodies (1e-5 relative, BASELINE's bar)."""
    n = 1 << 22
    over = dict(max_particles_num=n, flags=ps.FLAG_ALL_PAIRS)
    one = ps.ParticleSystem(ps.default_config(**over))
    xyz = one.uniform_cloud(n, 33)
    rng = np.random.default_rng(33)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = np.full(n, 1e6, np.float32)
    ids = one.fill_particles(xyz, age=age, fert_age=fert)
    one.init_iframe(); one.build_grid()
    cg = one.download_cellgrid()
    assert cg[:, 0].sum() == n and cg[:, 0].max() <= one.sizes.max_per_cell
    order = np.concatenate([row[1:1 + row[0]] for row in cg])              # slot id at every place of the sorted order
    one.calc_forces_pairs()
    where = np.empty(one.sizes.container_size, np.int64)
    where[ids] = np.arange(n)
    import torch
    pos = torch.from_numpy(xyz.astype(np.float64)).cuda()
    W = 6000
    worst = 0.0
    for lo in (0, n // 2 + 777, n - W):
        f = one.download_force4(lo, W)
        free = np.nonzero(f[:, 3].view(np.int32) == 0)[0][:200]              # a flagged particle's force is never looked at
        assert len(free) == 200, "window at %d holds only %d particles the force pass visits" % (lo, len(free))
        idx = torch.from_numpy(where[order[lo + free]]).cuda()
        exact = torch.empty((200, 3), dtype=torch.float64, device="cuda")
        for a in range(0, 200, 20):                                          # 20 particles x 2^22 bodies x 3 doubles at a time
            d = pos[None, :, :] - pos[idx[a:a + 20], None, :]
            r2 = (d * d).sum(2) + 0.2
            sc = 60.0 / (r2 * torch.sqrt(r2))
            sc[torch.arange(len(idx[a:a + 20]), device="cuda"), idx[a:a + 20]] = 0.0     # not itself (ps.cpp:1258)
            exact[a:a + 20] = (d * sc[:, :, None]).sum(1)
            del d, r2, sc
        ex = exact.cpu().numpy()
        rel = np.linalg.norm(f[free, :3].astype(np.float64) - ex, axis=1) / np.linalg.norm(ex, axis=1)
        worst = max(worst, float(rel.max()))
    del pos
    torch.cuda.empty_cache()
    print("configs[3], all-pairs N=2^22 vs fp64 direct sum (3 x 200 particles): max relative deviation %.3g" % worst)
    assert worst < 1e-5
    one.calc_forces_apply()
    whole = digest(one.download_particles(), *one.download_queues())
    c1 = one.counters
    one.close()
    world = 8
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=world, **over)) for r in range(world)]
    for h in ranks:
        h.fill_particles(xyz, age=age, fert_age=fert)
    assert ranks[0].msg_bytes(ps.MSG_ALLG_OUT) > 0 and ranks[0].msg_bytes(ps.MSG_ALLG_IN) == world * ranks[0].msg_bytes(ps.MSG_ALLG_OUT)
    step_local(ranks)                                                        # (the snapshot all-gather is one of its exchanges)
    plans = [h.slab_plan() for h in ranks]
    union_p = merge_owned([h.download_particles() for h in ranks], plans)
    qs = [h.download_queues() for h in ranks]
    union = digest(union_p, merge_owned([q[0] for q in qs], plans, "records"), merge_owned([q[1] for q in qs], plans))
    assert union == whole
    assert sum(h.counters["integrated"] for h in ranks) == c1["integrated"] > 0
    for h in ranks:
        h.close()


@pytest.mark.timeout(1500)
def test_config4_n24_in_40_cubed_cells():
    """configs[4]: N = 2^24, grid scaled to the reference's density (40^3 cells, 262 per cell):
    sorted partition, windows of the pair pass vs the oracle, determinism of a whole step."""
    n = 1 << 24
    over = dict(max_particles_num=n, chunk_factor=10)
    g = ps.ParticleSystem(ps.default_config(**over))
    assert g.sizes.num_cells == 64000
    xyz = g.uniform_cloud(n, 12)
    rng = np.random.default_rng(12)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = np.full(n, 1e6, np.float32)
    ids = g.fill_particles(xyz, age=age, fert_age=fert)
    g.snapshot_save()
    g.init_iframe(); g.build_grid()
    cg = g.download_cellgrid()
    counts = cg[:, 0]
    assert counts.sum() == n and counts.max() <= g.sizes.max_per_cell
    for c in range(0, g.sizes.num_cells, 997):
        row = cg[c, 1:1 + counts[c]]
        assert (np.diff(row) > 0).all()
    g.calc_forces_pairs()
    o = O.System(oracle_cfg_from(g.cfg))
    assert np.array_equal(o.fill(xyz, age=age, fert_age=fert), ids)
    o.init_iframe(); o.build_grid()
    assert np.array_equal(cg, o.cellgrid)
    total = o.sorted_count()
    windows_match_oracle(g, o, [(0, 4000), (total // 2 + 777, total // 2 + 4777), (total - 4000, total)])
    o.close()
    g.calc_forces_apply()
    a = digest(g.download_particles(), *g.download_queues())
    relocations = g.counters["relocations"]
    assert relocations > 0
    g.snapshot_restore(); g.step(1)
    assert a == digest(g.download_particles(), *g.download_queues())
    g.close()
    # ... and on the partition the config names: eight slabs of five cell layers each (here all on this GPU, the messages
    # copied rank to rank), every rank holding only its own segments -- slots, particles, queues (ps.cpp:431-487: a
    # subtask's 27 segments) -- with halo, force and transfer messages between ring neighbours.  The union of the eight
    # after one step is the single context's state, byte for byte.
    W = 8
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=W, **over)) for r in range(W)]
    plans = [h.slab_plan() for h in ranks]
    assert [(pl.cut_lo, pl.cut_hi) for pl in plans] == [(5 * r, 5 * r + 5) for r in range(W)]
    for h in ranks:
        h.fill_particles(xyz, age=age, fert_age=fert)
    step_local(ranks)
    union_p = np.zeros(ranks[0].sizes.container_size, ps.P_DTYPE)
    union_qi = np.zeros(ranks[0].sizes.queue_info_size, ps.Q_DTYPE)
    union_q = np.zeros(ranks[0].sizes.container_size, np.int32)
    moved = 0
    for r, (h, pl) in enumerate(zip(ranks, plans)):
        h.synchronize()
        p = h.download_particles()
        if r == 0:
            union_p[...] = p                     # (slots nobody owns do not exist; a rank reports foreign slots as free records)
        qi, q = h.download_queues()
        if r == 0:
            union_qi[...] = qi; union_q[...] = q
        for t in range(4):
            union_p[pl.slot_lo[t]:pl.slot_hi[t]] = p[pl.slot_lo[t]:pl.slot_hi[t]]
            union_q[pl.slot_lo[t]:pl.slot_hi[t]] = q[pl.slot_lo[t]:pl.slot_hi[t]]
            union_qi[pl.rec_lo[t]:pl.rec_hi[t]] = qi[pl.rec_lo[t]:pl.rec_hi[t]]
        moved += h.counters["relocations"]
        del p, qi, q
        h.close()
    assert digest(union_p, union_qi, union_q) == a
    assert moved == relocations
