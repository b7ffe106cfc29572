"""The slab-partitioned (multi-GPU) path on ONE GPU: `world` contexts in one process, each
holding only its slab -- the segments of its cell layers with their slots, particles and
free-slot queues -- stepped through the four psamd_slab_* stage calls with the halo / force /
transfer messages copied rank to rank (particlesystem_amd.slab.step_local; the transport is
the only thing that differs from an RCCL run).  The union of the ranks' states must equal the
single-system oracle byte for byte, every step, including the particles that change owner
and the slot ids the neighbours' queues hand out to them."""
import numpy as np
import pytest

import oracle_py as O
import particlesystem_amd as ps
from particlesystem_amd.slab import merge_owned, step_local
from util import assert_same_particles, cloud, explosion_rng, g2_cloud, oracle_cfg_from

pytestmark = pytest.mark.gpu

COUNTERS = ("deaths_age", "deaths_collision", "survives", "integrated", "relocations",
            "relocations_lost", "births", "births_failed", "cell_overflow_kills")


def make_world(world, xyz, age, fert, flags=0, cuts=None, **over):
    ranks = []
    for r in range(world):
        kw = dict(rank=r, world=world, flags=flags, **over)
        if cuts is not None:
            kw["cuts"] = cuts
        ranks.append(ps.ParticleSystem(ps.default_config(**kw)))
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    ids_o = o.fill(xyz, age=age, fert_age=fert)
    ids = np.stack([g.fill_particles(xyz, age=age, fert_age=fert) for g in ranks])
    # every particle was placed by exactly one rank, in the slot the oracle gave it
    assert ((ids >= 0).sum(0) == 1).all()
    assert np.array_equal(ids.max(0), ids_o)
    return ranks, o


def compare_world(ranks, o, what):
    plans = [g.slab_plan() for g in ranks]
    p = merge_owned([g.download_particles() for g in ranks], plans)
    assert_same_particles(p, o.particles, what)
    qs = [g.download_queues() for g in ranks]
    qi = merge_owned([q[0] for q in qs], plans, "records")
    q = merge_owned([q[1] for q in qs], plans)
    assert qi.tobytes() == o.queue_info.tobytes(), what + ": QUEUE_INFO differs"
    assert np.array_equal(q, o.queue), what + ": queue array differs"
    cs = [g.counters for g in ranks]
    for k in COUNTERS:
        assert sum(c[k] for c in cs) == o.counters[k], (what, k, [c[k] for c in cs], o.counters[k])
    return cs


def changed_owner(ranks):
    """particles that arrived from a neighbour so far = relocations + births whose slot the
    neighbour's queue handed out: visible as xfer records sent"""
    return sum(int(g.msg_download(ps.MSG_XFER_OUT + k)[0]) for g in ranks for k in (0, 1))


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_slab_world_matches_oracle_every_step(world):
    n = 60000
    xyz = cloud(n, 300 + world)
    rng = np.random.default_rng(world)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = (1e6 + np.arange(n)).astype(np.float32)
    ranks, o = make_world(world, xyz, age, fert)
    moved = 0
    for step in range(12):
        step_local(ranks)
        o.step(1)
        moved += changed_owner(ranks)
        compare_world(ranks, o, "world %d step %d" % (world, step + 1))
    assert moved > 0 and o.counters["relocations"] > 0 and o.counters["deaths_collision"] > 0
    for g in ranks:
        g.close()


def test_slab_wraps_around_the_periodic_box():
    """fast particles: |v| up to 90 => every particle moves a whole cell per step (MAX_DX clamp),
    so the top and bottom layers trade particles through the wrap, rank world-1 <-> rank 0."""
    n, world = 30000, 4
    xyz = cloud(n, 41)
    rng = np.random.default_rng(41)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=world)) for r in range(world)]
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    v = rng.uniform(-90, 90, (n, 3)).astype(np.float32)
    ids_o = o.fill(xyz, age=age, fert_age=np.float32(1e6))
    pp = o.particles
    pp["vx"][ids_o], pp["vy"][ids_o], pp["vz"][ids_o] = v.T
    for g in ranks:
        g.fill_particles(xyz, age=age, fert_age=np.float32(1e6), vxyz=v)
    for step in range(6):
        step_local(ranks); o.step(1)
        compare_world(ranks, o, "wrap step %d" % (step + 1))
    sent_down_by_rank0 = int(ranks[0].msg_download(ps.MSG_XFER_OUT + 0)[0])
    sent_up_by_last = int(ranks[-1].msg_download(ps.MSG_XFER_OUT + 1)[0])
    assert sent_down_by_rank0 > 0 and sent_up_by_last > 0
    for g in ranks:
        g.close()


def test_slab_births_cross_ranks():
    """explosions on: a child is born into the parent's NEW segment, which may be a
    neighbour's; the counter-based RNG is keyed by the parent's id, so the child is the same
    whoever places it."""
    n, world, seed = 40000, 4, 777
    xyz = cloud(n, 51)
    rng = np.random.default_rng(51)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = rng.uniform(2.5, 8.0, n).astype(np.float32)
    ranks, o = make_world(world, xyz, age, fert, flags=ps.FLAG_EXPLOSIONS, seed=seed)
    o.set_rng(explosion_rng(seed))
    for step in range(10):
        step_local(ranks); o.step(1)
        compare_world(ranks, o, "births step %d" % (step + 1))
    assert o.counters["births"] > 100
    for g in ranks:
        g.close()


@pytest.mark.parametrize("cuts", [[0, 7, 16], [0, 9, 16], [0, 3, 16], [0, 4, 7, 11, 16], [0, 2, 5, 8, 10, 13, 16]])
def test_slab_uneven_and_group_aligned_cuts(cuts):
    """cuts at 7|: aligned with a segment group (a halo layer travels DOWN as well, nothing is
    lent); 9|, 3|: the other parity; mixed."""
    world = len(cuts) - 1
    n = 30000
    xyz = cloud(n, 61)
    rng = np.random.default_rng(61)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    ranks, o = make_world(world, xyz, age, np.float32(1e6), cuts=cuts)
    for step in range(6):
        step_local(ranks); o.step(1)
        compare_world(ranks, o, "cuts %r step %d" % (cuts, step + 1))
    for g in ranks:
        g.close()


def test_slab_interior_pass_while_the_halo_travels():
    """the optional split of the pair stage -- interior cells first, the rest after the halo --
    gives the same state (4 slabs of 4 layers: two interior layers each)"""
    n = 60000
    xyz = cloud(n, 66)
    rng = np.random.default_rng(66)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    ranks, o = make_world(4, xyz, age, np.float32(1e6))
    for step in range(6):
        step_local(ranks, overlap_interior=True); o.step(1)
        compare_world(ranks, o, "interior pass, step %d" % (step + 1))
    for g in ranks:
        g.close()


def test_slab_g2_cloud_100_steps():
    """BASELINE config 0's cloud, 100 steps on three slabs: the reference's own life-cycle
    counts come out of the union."""
    xyz = g2_cloud()
    fert = (1e6 + np.arange(len(xyz))).astype(np.float32)
    ranks, o = make_world(3, xyz, np.float32(40 * 0.01), fert, dt=0.01)
    for step in range(100):
        step_local(ranks)
    o.step(100)
    cs = compare_world(ranks, o, "G2 on 3 slabs after 100 steps")
    assert sum(g.live_count() for g in ranks) == 2724 and sum(c["relocations"] for c in cs) == 1537
    for g in ranks:
        g.close()


def test_slab_other_grid_and_one_pass_mode():
    """chunk_dim 5 (three-layer inner groups) and an EPS2 that forces the generic one-pass
    pair kernel (collision flags come from the force walk, remote ids and ages are used)."""
    over = {"chunk_factor": 3, "chunk_dim": 5, "max_particles_num": 20000}
    n = 15000
    xyz = cloud(n, 71, 34.9)              # the odd grid is not centred: x in [-35, 40), y and z in (-40, 35]
    rng = np.random.default_rng(71)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    for eps2 in (0.2, 1e-20):
        ranks, o = make_world(3, xyz, age, np.float32(1e6), eps2=eps2, **over)
        for step in range(4):
            step_local(ranks); o.step(1)
            cs = compare_world(ranks, o, "15^3 grid eps2=%g step %d" % (eps2, step + 1))
        # 12 slots per cell, 4.4 particles on average: some cells overflow on ranks 1 and 2; the
        # reference frees the killed slots into queue record 0 (ps.cpp:1523-1526), which rank 0
        # holds: they get there in the status record
        assert cs[1]["cell_overflow_kills"] + cs[2]["cell_overflow_kills"] > 0
        for g in ranks:
            g.close()


def test_world_one_slab_calls_are_the_plain_step():
    xyz = cloud(20000, 81)
    g = ps.ParticleSystem(ps.default_config())
    o = O.System(oracle_cfg_from(g.cfg))
    g.fill_particles(xyz, age=np.float32(3.0), fert_age=np.float32(1e6))
    o.fill(xyz, age=np.float32(3.0), fert_age=np.float32(1e6))
    for step in range(3):
        step_local([g]); o.step(1)
        assert_same_particles(g.download_particles(), o.particles, "world 1 step %d" % (step + 1))
    assert all(g.msg_bytes(k) == 0 for k in range(10))
    with pytest.raises(ps.PsamdError):
        g.slab_pairs()                      # out of order
    g.close()


def test_slab_context_refuses_the_plain_stage_calls_and_bad_worlds():
    g = ps.ParticleSystem(ps.default_config(rank=0, world=2))
    with pytest.raises(ps.PsamdError):
        g.step(1)
    g.close()
    with pytest.raises(ps.PsamdError):
        ps.ParticleSystem(ps.default_config(rank=0, world=16))      # 16 layers: one per rank is not enough


def test_transfer_messages_grow_with_the_traffic():
    """A cloud whose density rises towards its tail drifts across the cut between rank 0 and rank 1: the records that change
    owner per step go from a handful to many hundreds.  The transfer messages start with room for 64 (xfer_cap) and a
    step would be refused at ~70; instead every rank reports its traffic in its status record, all ranks apply the same
    rule to the same numbers, and the messages grow on every rank in the same step, two steps ahead of the need -- the run
    goes on, byte for byte the one-context run, and nobody negotiated anything.  When the dense tail has crossed, the thin
    front is at the next cut: the messages shrink again (the same rule, the same step on every rank), and grow with the next
    ramp."""
    world, n, steps = 4, 30000, 24
    rng = np.random.default_rng(511)
    xyz = np.empty((n, 3), np.float32)
    xyz[:, 0:2] = rng.uniform(-39.9, 39.9, (n, 2))
    xyz[:, 2] = 20.02 + 18.9 * rng.uniform(0, 1, n) ** (1.0 / 3.0)          # cell layers i3 = 0..3 (rank 0), thin at the cut (z = 20)
    v = np.zeros((n, 3), np.float32)
    v[:, 2] = -10.0                                                         # half a cell width per step towards the cut (i3 grows as z falls)
    over = dict(max_particles_num=1 << 18, collision_radius=0.0)
    fert = (1e6 + np.arange(n)).astype(np.float32)
    kw = dict(age=np.float32(3.0), fert_age=fert, vxyz=v, w=np.float32(1e-3))   # (light particles: gravity leaves the drift alone)
    one = ps.ParticleSystem(ps.default_config(**over))
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=world, xfer_cap=64, **over)) for r in range(world)]
    for g in [one] + ranks:
        g.fill_particles(xyz, **kw)
    plans = [g.slab_plan() for g in ranks]
    first = ranks[0].msg_bytes(ps.MSG_XFER_OUT)
    sizes, sent = [], []
    for k in range(steps):
        one.step(1)
        step_local(ranks)
        sizes.append([g.msg_bytes(ps.MSG_XFER_OUT) for g in ranks])
        sent.append(int(ranks[0].msg_download(ps.MSG_XFER_OUT + 1, 64)[0]))
        assert len(set(sizes[-1])) == 1, "the ranks disagree on the transfer messages' size in step %d: %r" % (k + 1, sizes[-1])
        union = merge_owned([g.download_particles() for g in ranks], plans)
        assert_same_particles(union, one.download_particles(), "drifting cloud, step %d" % (k + 1))
    for g in ranks:
        g.synchronize()                                                     # (no sticky error anywhere)
    print("records rank 0 sent up per step:", sent, "message bytes:", [s_[0] for s_ in sizes])
    per_step = [s_[0] for s_ in sizes]
    assert max(sent) > 3 * 64 and max(per_step) > 4 * first and max(per_step) <= ranks[0].slab_buffers().xfer_bytes_max
    assert min(per_step) >= first                                           # never below what the contexts were created with
    assert any(b < a for a, b in zip(per_step, per_step[1:])), "the messages never shrank again: %r" % per_step
    for g in [one] + ranks:
        g.close()


def test_slab_message_overflow_is_loud():
    """halo_cap_cell too small for what a boundary LAYER holds (8 per cell on average, the cloud has 15) => a sticky
    error on the ranks involved, not a silent truncation.  (One crowded cell alone is served: the room is pooled over
    the layer, test_slab_halo_room_is_pooled_over_a_layer.)"""
    n = 60000
    xyz = cloud(n, 91)
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=2, halo_cap_cell=8)) for r in range(2)]
    for g in ranks:
        g.fill_particles(xyz, age=np.float32(3.0), fert_age=np.float32(1e6))
    with pytest.raises(ps.PsamdError):
        step_local(ranks)
        for g in ranks:
            g.synchronize()
    for g in ranks:
        g.close()


def test_slab_halo_room_is_pooled_over_a_layer():
    """halo_cap_cell is the room PER CELL ON AVERAGE over a cell layer: a clump of 300 particles in one boundary
    cell of a layer that holds 1 250 is served by a message with room for 16 per cell (4 096 a layer) -- until round 4
    every cell was held to the figure on its own and this cloud was refused.  Byte-equal to the oracle, with the
    clump's bodies crossing the cut as halo, as lent cells' force records and as particles that change owner."""
    n = 20000
    xyz = cloud(n, 92)
    rng = np.random.default_rng(92)
    m = 300
    for k, z in enumerate((-0.5, 0.5, 19.5, -19.5)):         # clumps in the cell layers on both sides of the cuts of a 4-rank world (z = 0, +-20)
        xyz[k * m:(k + 1) * m] = (rng.uniform(-1.8, 1.8, (m, 3)) + np.array([7.5 + 10 * k, -12.5, z])).astype(np.float32)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = (1e6 + np.arange(n)).astype(np.float32)
    ranks, o = make_world(4, xyz, age, fert, halo_cap_cell=16)
    o.init_iframe(); o.build_grid()
    assert int(o.cellgrid[:, 0].max()) > 100                  # one cell far above the average the message is sized for
    for step in range(6):
        step_local(ranks)
        o.step(1)
        compare_world(ranks, o, "pooled halo, step %d" % (step + 1))
    assert o.counters["deaths_collision"] > 100 and changed_owner(ranks) >= 0
    for g in ranks:
        g.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_slab_chunk_list_capacity_rule(world):
    """The chunk lists' capacity rule (ps.cpp:1502-1508; one GPU: test_gpu_parity's
    test_chunk_list_capacity_rule) needs the rank of a particle among ALL slots of its chunk, and
    a chunk's 27 segments are spread over up to three ranks (4 ranks: chunk layer 1 lives on ranks
    0 and 1; 8 ranks: on 1, 2 and 3).  Every rank reports its part of every chunk per segment type in
    the status record; with all records in, each ranks its own particles behind what precedes them in
    slot order.  Same cloud as the one-GPU test, byte-equal to the oracle every step."""
    over = {"max_particles_num": 4096}
    cfg = ps.default_config(**over)
    G, cs = cfg.chunk_factor * cfg.chunk_dim, cfg.cell_size
    rng = np.random.default_rng(5)
    pts = []
    for i3 in range(4, 8):
        for i1 in range(4, 8):
            for i2 in range(4, 8):
                for _ in range({0: 4, 1: 6, 2: 7, 3: 9}[sum(v in (4, 7) for v in (i1, i2, i3))]):
                    u = rng.uniform(0.05, 0.95, 3)
                    pts.append(((i2 - G / 2 + u[0]) * cs, -(i1 - G / 2 + u[1]) * cs, -(i3 - G / 2 + u[2]) * cs))
    rest = cloud(1500, 23)                 # ordinary traffic elsewhere
    idx = np.floor(rest.astype(np.float64) * [1, -1, -1] / cs).astype(int) + G // 2
    rest = rest[~((idx >= 2) & (idx <= 9)).all(1)]
    xyz = np.concatenate([np.array(pts, np.float32), rest])
    age = np.random.default_rng(6).uniform(2.2, 7.0, len(xyz)).astype(np.float32)
    fert = (1e6 + np.arange(len(xyz))).astype(np.float32)
    ranks, o = make_world(world, xyz, age, fert, **over)
    for k in range(4):
        step_local(ranks); o.step(1)
        if k == 0:
            assert o.chunkgrid[:, 0].max() > o.d.max_per_chunk, "the scenario must pass the chunk list's capacity"
        compare_world(ranks, o, "chunk capacity on %d slabs, step %d" % (world, k + 1))
    assert o.counters["cell_overflow_kills"] > 100 and o.counters["integrated"] > 0
    for g in ranks:
        g.close()


def test_slab_failure_is_collective():
    """An error raised AFTER the build stage (here: a transfer message with room for 4 records) is
    known to the rank that raised it and, through the message header, to its ring neighbours -- not
    to the ranks further away, which would wait in the next exchange for a peer that has stopped.
    So a slab fails only on what the all-gathered status records show: the sticky bit goes out with
    the next step's record and EVERY rank gets the error as that step's verdict (from the slab_finish after it, or
    from psamd_synchronize)."""
    world = 4
    n = 60000
    xyz = cloud(n, 302)
    v = np.zeros((n, 3), np.float32)
    v[:, 2] = 90.0                          # everything moves a layer per step: far more than 4 records per face
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=world, xfer_cap=4)) for r in range(world)]
    for g in ranks:
        g.fill_particles(xyz, age=np.float32(3.0), fert_age=np.float32(1e6), vxyz=v)
    step_local(ranks)                       # the overflow happens in slab_apply of this step: nobody fails yet
    failed = 0
    for g in ranks:                         # the next step: every rank's record carries the bit; all fail together
        g.slab_build()
    every = np.concatenate([g.msg_download(ps.MSG_STATUS_OUT) for g in ranks])
    from particlesystem_amd.slab import routes
    for r, g in enumerate(ranks):
        g.msg_upload(ps.MSG_STATUS_IN, every)
        for ph, out_slot, peer, in_slot in routes(r, world):
            if ph == "halo" and g.msg_bytes(out_slot):
                ranks[peer].msg_upload(in_slot, g.msg_download(out_slot))
    for g in ranks:
        g.slab_pairs()
    for r, g in enumerate(ranks):
        for ph, out_slot, peer, in_slot in routes(r, world):
            if ph == "force" and g.msg_bytes(out_slot):
                ranks[peer].msg_upload(in_slot, g.msg_download(out_slot))
    for g in ranks:
        g.slab_apply()
    for r, g in enumerate(ranks):
        for ph, out_slot, peer, in_slot in routes(r, world):
            if ph == "xfer" and g.msg_bytes(out_slot):
                ranks[peer].msg_upload(in_slot, g.msg_download(out_slot))
    for g in ranks:
        with pytest.raises(ps.PsamdError):
            g.slab_finish()                 # (a call reports the steps before the one it enqueues: run-ahead) ...
            g.synchronize()                 # ... and this, whatever is outstanding: the step's own verdict
        failed += 1
    assert failed == world
    for g in ranks:
        g.close()


def test_slab_two_layer_jump_across_a_cut():
    """MAX_DX = CELL_SIZE moves a particle one cell layer -- or two: one ulp below a cell face,
    moved by exactly +CELL_SIZE, the rounded sum lands ON the far face (4.9999995 + 5 = 10.0 in
    fp32).  Across a slab cut that is a transfer to the neighbour two layers in; it used to be
    sent the wrong way round the ring (found by scripts/fuzz_parity.py as a device fault)."""
    n = 4000
    rng = np.random.default_rng(171)
    cs = np.float32(5.0)
    xyz = np.zeros((n, 3), np.float32)
    xyz[:, 0] = rng.uniform(-39, 39, n)
    xyz[:, 1] = rng.uniform(-39, 39, n)
    u3 = np.nextafter(cs, np.float32(0))                       # one ulp below the face between layers 8 and 9
    xyz[:, 2] = -u3
    xyz[n // 2:, 2] = -np.nextafter(np.float32(-5.0), np.float32(-10))   # and one ulp below the face -5 (layer 6), moving up too
    v = np.zeros((n, 3), np.float32)
    v[:, 2] = -300.0                                           # -z is up the layers; clamped to one CELL_SIZE per step
    age = np.full(n, 3.0, np.float32)
    fert = (1e6 + np.arange(n)).astype(np.float32)
    over = dict(collision_radius=0.0, cuts=[0, 9, 16])
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=2, **over)) for r in range(2)]
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    ids = o.fill(xyz, age=age, fert_age=fert)
    p = o.particles
    p["vx"][ids], p["vy"][ids], p["vz"][ids] = v.T
    for g in ranks:
        g.fill_particles(xyz, age=age, fert_age=fert, vxyz=v)
    GG = 16 * 16
    layer0 = {int(t): int(c) // GG for t, c in zip(p["fertility_age"][ids], p["cell"][ids])}
    for step in range(3):
        step_local(ranks); o.step(1)
        compare_world(ranks, o, "two-layer jump step %d" % (step + 1))
        if step == 0:
            live = o.particles["cell"] >= 0
            jumps = [int(c) // GG - layer0[int(t)] for t, c in zip(o.particles["fertility_age"][live], o.particles["cell"][live])]
            assert jumps.count(2) >= n // 2, "the scenario must contain two-layer jumps (8 -> 10 across the cut at 9)"
    for g in ranks:
        g.close()


def test_slab_two_layer_jump_over_a_single_layer_rank():
    """A 15^3 grid (3-cell chunks) cut every two layers: rank 3 of 7 holds ONE layer of state (layer 7,
    the interior layer of chunk layer 2).  Particles in layer 6, a hair below the face z = 0, move by
    exactly one CELL_SIZE: the rounded sum lands ON the face 5 -- layer 8, two layers up, which belongs
    to rank 4.  The record flies over rank 3: it travels in the hop-two outbox, rank 2 -> rank 4
    (psamd_slab_buffers.xfer2_*), and takes its slot from rank 4's queue in the reference's serial
    order (ps.cpp:1335-1374 relocates to any segment).  This used to be a refusal (ERR_SLAB_MISMATCH)."""
    n = 1600                                            # (800 jumpers: the hop-two messages have room for 1024 records)
    rng = np.random.default_rng(181)
    xyz = np.zeros((n, 3), np.float32)
    xyz[:, 0] = rng.uniform(-34, 39, n)                 # (an odd grid is not centred: i = floor(c / 5) + 7, c in [-35, 40))
    xyz[:, 1] = rng.uniform(-39, 34, n)
    xyz[:, 2] = np.float32(1e-10)                       # -z = -1e-10: layer 6 (of 0..14), a hair below the face
    xyz[n // 2:, 2] = rng.uniform(0.5, 4.5, n - n // 2)     # and ordinary particles of layer 6, one layer per step
    v = np.zeros((n, 3), np.float32)
    v[:, 2] = -300.0                                    # -z grows: up the layers; clamped to one CELL_SIZE per step
    age = np.full(n, 3.0, np.float32)
    fert = (1e6 + np.arange(n)).astype(np.float32)
    over = dict(collision_radius=0.0, chunk_factor=5, chunk_dim=3, max_particles_num=60000, cuts=[0, 2, 4, 6, 8, 10, 12, 15])
    world = 7
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=world, **over)) for r in range(world)]
    assert ranks[3].slab_plan().state_hi - ranks[3].slab_plan().state_lo == 1
    assert ranks[2].msg_bytes(ps.MSG_XFER2_OUT + 1) > 0
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    ids = o.fill(xyz, age=age, fert_age=fert)
    p = o.particles
    p["vx"][ids], p["vy"][ids], p["vz"][ids] = v.T
    for g in ranks:
        g.fill_particles(xyz, age=age, fert_age=fert, vxyz=v)
    GG = 15 * 15
    layer0 = {int(t): int(c) // GG for t, c in zip(p["fertility_age"][ids], p["cell"][ids])}
    flew = 0
    for step in range(4):
        step_local(ranks); o.step(1)
        compare_world(ranks, o, "jump over a single-layer rank, step %d" % (step + 1))
        flew += int(ranks[2].msg_download(ps.MSG_XFER2_OUT + 1)[0])
        if step == 0:
            live = o.particles["cell"] >= 0
            jumps = [int(c) // GG - layer0[int(t)] for t, c in zip(o.particles["fertility_age"][live], o.particles["cell"][live])]
            assert jumps.count(2) >= n // 2 - 5, "the scenario must contain two-layer jumps (6 -> 8 over rank 3's layer 7)"
    assert flew >= n // 2 - 5, "the records must have gone two ranks up in one hop"
    for g in ranks:
        g.close()


@pytest.mark.parametrize("world,grid", [(8, {}), (6, dict(chunk_factor=3, chunk_dim=4))])
def test_slab_far_relocation_of_a_particle_that_is_no_number(world, grid):
    """A particle whose velocity is not a number (what a child born with the direction (0, 0, 0) gets: 0/0,
    ps.cpp:1306-1333) has, a step later, a position that is not one, and the reference files it under one fixed
    cell wherever it was: (0,0,0) in the default 16^3 grid, (4,4,4) for G = 12.  On slabs its record must reach the
    rank that holds that cell's segment, possibly across the ring: it travels in the far outbox
    (psamd_slab_buffers.far_*), all-gathered in the transfer phase when births are on, and takes its slot from that
    rank's queue in the reference's serial order.  A kid and an adult start in the middle of the box, among a
    crowd; every byte of the merged ranks against the oracle.  (Until round 3: a refusal, ERR_FOREIGN_CELL.)"""
    rng = np.random.default_rng(191)
    over = dict(grid)
    G = over.get("chunk_factor", 4) * over.get("chunk_dim", 4)
    L = 0.5 * G * 5.0 * 0.99
    n = 6000
    xyz = rng.uniform(-L, L, (n, 3)).astype(np.float32)
    # a crowd around the cell the two will be filed under, so that the adult meets somebody there
    home = 0 if G == 16 else 4
    lo = -0.5 * G * 5.0
    c0 = np.array([lo + 5.0 * home, -(lo + 5.0 * home) - 10.0, -(lo + 5.0 * home) - 10.0], np.float32)      # x up, y and z down
    xyz[:600] = (c0 + rng.uniform(0.2, 9.8, (600, 3))).astype(np.float32)
    age = rng.uniform(2.0, 9.0, n).astype(np.float32)
    v = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    # the two start in layers of ranks at least two away on the ring from the one that holds the home cell
    # (16^3, 8 ranks: layers 7 and 8, ranks 3 and 4, home on rank 0; 12^3, 6 ranks: layers 10 and 0, ranks 5 and 0, home on rank 2)
    z_a, z_b = (1.5, -3.5) if G == 16 else (-20.0, -27.0)
    xyz[1000] = (1.0, 2.0, z_a); age[1000] = 0.05; v[1000] = np.nan
    xyz[1001] = (-7.0, 4.0, z_b); age[1001] = 4.0; v[1001] = np.nan
    fert = (1e6 + np.arange(n)).astype(np.float32)
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=world, flags=ps.FLAG_EXPLOSIONS, **over)) for r in range(world)]
    assert ranks[0].msg_bytes(ps.MSG_FAR_OUT) > 0 and ranks[0].msg_bytes(ps.MSG_FAR_IN) == world * ranks[0].msg_bytes(ps.MSG_FAR_OUT)
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    ids = o.fill(xyz, age=age, fert_age=fert)
    p = o.particles
    p["vx"][ids], p["vy"][ids], p["vz"][ids] = v.T
    for g in ranks:
        g.fill_particles(xyz, age=age, fert_age=fert, vxyz=v)
    far = 0
    for step in range(4):
        step_local(ranks); o.step(1)
        compare_world(ranks, o, "far relocation, world %d, step %d" % (world, step + 1))
        far += sum(int(g.msg_download(ps.MSG_FAR_OUT)[0]) for g in ranks)
    live = o.particles["cell"] >= 0
    assert np.isnan(o.particles["x"][live]).sum() >= 1
    assert far >= 1, "a record must have travelled in the far outbox"       # (the other may start next to the home rank: the state cuts follow the segments)
    for g in ranks:
        g.close()


def test_slab_far_outbox_left_out_is_loud():
    """A caller that exchanges the neighbour messages but not the far outboxes (an older transport) would lose the
    records in them: every rank checks that every rank's far outbox arrived this step."""
    world = 4
    rng = np.random.default_rng(193)
    xyz = rng.uniform(-39, 39, (2000, 3)).astype(np.float32)
    age = rng.uniform(2.0, 9.0, 2000).astype(np.float32)
    fert = (1e6 + np.arange(2000)).astype(np.float32)
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=world, flags=ps.FLAG_EXPLOSIONS)) for r in range(world)]
    for g in ranks:
        g.fill_particles(xyz, age=age, fert_age=fert)
    step_local(ranks)                                          # with the far all-gather: fine
    from particlesystem_amd import slab
    far_out = slab.FAR_OUT
    slab.FAR_OUT = 99                                          # (no such message: step_local skips the all-gather)
    try:
        with pytest.raises(ps.PsamdError, match="does not match"):
            for _ in range(2):                                 # (raised in finish, reported collectively with the next step's status)
                step_local(ranks)
                for g in ranks:
                    g.synchronize()
    finally:
        slab.FAR_OUT = far_out
    for g in ranks:
        g.close()
