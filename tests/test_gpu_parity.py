"""GPU parity: the HIP path behind the C ABI against the CPU oracle, bit for bit.

Everything here is integer/index work or fp32 arithmetic reproduced operation by
operation, so the bar is exact equality of the reference-layout buffers
(P_DATA_TYPE records, free-slot queues, cell/chunk lists, T_DATA snapshot) --
no tolerance anywhere in this file.  The fast-math mode has its own test file.
"""
import numpy as np
import pytest

import oracle_py as O
import particlesystem_amd as ps
from util import assert_same_particles, cloud, explosion_rng, g2_cloud, oracle_cfg_from

pytestmark = pytest.mark.gpu


def make_pair(xyz, age, fert, flags=0, w=None, vxyz=None, **over):
    cfg = ps.default_config(flags=flags, **over)
    g = ps.ParticleSystem(cfg)
    o = O.System(oracle_cfg_from(cfg))
    ids_g = g.fill_particles(xyz, age=age, fert_age=fert, w=w, vxyz=vxyz)
    ids_o = o.fill(xyz, age=age, fert_age=fert, w=w)
    if vxyz is not None:
        p = o.particles
        p["vx"][ids_o], p["vy"][ids_o], p["vz"][ids_o] = np.asarray(vxyz, np.float32).T
    assert np.array_equal(ids_g, ids_o)
    return g, o


def compare_all(g, o, what):
    assert_same_particles(g.download_particles(), o.particles, what)
    qi, q = g.download_queues()
    assert qi.tobytes() == o.queue_info.tobytes(), what + ": QUEUE_INFO differs"
    assert np.array_equal(q, o.queue), what + ": queue array differs"
    cg, co = g.counters, o.counters
    for k in ("deaths_age", "deaths_collision", "survives", "integrated", "relocations",
              "relocations_lost", "births", "births_failed", "cell_overflow_kills"):
        assert cg[k] == co[k], (what, k, cg[k], co[k])


def test_geometry_tables_match_oracle():
    import ctypes as C
    for over in ({}, {"chunk_factor": 2, "chunk_dim": 3, "max_particles_num": 1000},
                 {"chunk_factor": 3, "chunk_dim": 5, "max_particles_num": 5000},
                 {"chunk_factor": 1, "chunk_dim": 3, "max_particles_num": 100}):
        g = ps.ParticleSystem(ps.default_config(**over))
        oc = oracle_cfg_from(g.cfg)
        od = O.derive(oc)
        s = g.sizes
        assert (s.grid_dim, s.num_cells, s.num_chunks, s.max_per_cell, s.max_per_chunk, s.container_size,
                s.queue_info_size) == (od.grid_dim, od.num_cells, od.num_chunks, od.max_per_cell,
                                       od.max_per_chunk, od.container_size, od.queue_info_size)
        tab = g.cell_table()
        out3 = (C.c_int * 3)()
        for c in range(od.num_cells):
            O.lib().pso_get_cell_info(C.byref(od), C.byref(oc), c, out3)
            assert list(out3) == list(tab[c]), (over, c)
        pk = np.zeros(27, dtype=O.PAIR_DTYPE)
        mine = g.pkgdistrib()
        for ch in range(od.num_chunks):
            O.lib().pso_set_pkg_segments(C.byref(oc), ch, pk.ctypes.data)
            assert np.array_equal(np.stack([pk["c"], pk["p"]], 1).ravel(), mine[ch]), (over, ch)
        # initial queues = q_start_fast
        so = O.System(oc)
        qi, q = g.download_queues()
        assert qi.tobytes() == so.queue_info.tobytes() and np.array_equal(q, so.queue)
        assert_same_particles(g.download_particles(), so.particles, "fresh container")
        so.close()
        g.close()


def test_two_body_known_answers_on_gpu():
    """the reference's own numbers for the (-4,0,0)/(4,0,0) pair."""
    g, o = make_pair([[-4, 0, 0], [4, 0, 0]], age=np.float32(2.0), fert=1e6)
    g.step(1)
    p = g.download_particles()
    a = p[2738592]
    assert np.float32(a["ax"]) == np.float32(0.933122575)
    assert np.float32(a["vx"]) == np.float32(0.046656128)
    assert np.float32(a["x"]) == np.float32(-3.99883366)
    assert np.float32(a["age"]) == np.float32(2.04999995)
    o.step(1)
    compare_all(g, o, "two-body step 1")
    g.step(1)
    o.step(1)
    compare_all(g, o, "two-body step 2")


def test_stage_buffers_after_build_grid():
    xyz = cloud(20000, 3)
    g, o = make_pair(xyz, age=2.0, fert=1e6)
    g.init_iframe(); g.build_grid()
    o.init_iframe(); o.build_grid()
    assert np.array_equal(g.gridmax(), o.gridmax)
    assert np.array_equal(g.download_cellgrid(), o.cellgrid)
    assert np.array_equal(g.download_chunkgrid(), o.chunkgrid)
    assert g.download_tdata().tobytes() == o.tdata.tobytes()
    g.calc_forces(); o.calc_forces()
    compare_all(g, o, "stage-by-stage step")


@pytest.mark.parametrize("dt,steps", [(0.01, (1, 10, 100)), (0.05, (1, 10, 100))])
def test_g2_cloud_full_state(dt, steps):
    """BASELINE config 0: N=4096, 100 steps, whole container compared after 1/10/100."""
    xyz = g2_cloud()
    fert = (1e6 + np.arange(len(xyz))).astype(np.float32)
    g, o = make_pair(xyz, age=np.float32(40 * dt), fert=fert, dt=dt)
    done = 0
    for s in steps:
        g.step(s - done); o.step(s - done); done = s
        compare_all(g, o, "G2 dt=%g after %d steps" % (dt, s))
    if dt == 0.01:
        assert g.live_count() == 2724 and g.counters["relocations"] == 1537
    else:
        assert g.live_count() == 1716 and g.counters["relocations"] == 11375


@pytest.mark.parametrize("n,seed", [(1 << 16, 5), (1 << 18, 6)])
def test_dense_cloud_steps(n, seed):
    """16 and 64 particles per cell: long serial fp32 sums, many collisions and moves."""
    xyz = cloud(n, seed)
    rng = np.random.default_rng(seed)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)   # [MIN_ADULT_AGE, MAX_ADULT_AGE)
    g, o = make_pair(xyz, age=age, fert=1e6)
    for k in range(2):
        g.step(1); o.step(1)
        compare_all(g, o, "N=%d step %d" % (n, k + 1))


def test_ages_kids_and_elders():
    """kids exert/feel no force and never collide; age > PARTICLE_LIFE dies."""
    n = 30000
    xyz = cloud(n, 9)
    rng = np.random.default_rng(9)
    age = rng.choice(np.array([0.0, 1.4999999, 1.5, 1.5000001, 3.0, 14.96, 15.0, 15.000001, 16.0],
                              np.float32), n)
    g, o = make_pair(xyz, age=age, fert=1e6)
    for k in range(3):
        g.step(1); o.step(1)
        compare_all(g, o, "ages step %d" % (k + 1))
    assert g.counters["deaths_age"] > 0


def test_wrap_and_clamps():
    """fast particles leave the box and re-enter on the other side; dx and v clamp."""
    n = 5000
    rng = np.random.default_rng(13)
    xyz = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    xyz[: n // 2] = np.sign(xyz[: n // 2]) * rng.uniform(39.0, 39.999, (n // 2, 3)).astype(np.float32)
    v = rng.uniform(-300, 300, (n, 3)).astype(np.float32)
    g, o = make_pair(xyz, age=3.0, fert=1e6, vxyz=v)
    for k in range(4):
        g.step(1); o.step(1)
        compare_all(g, o, "wrap step %d" % (k + 1))


def test_cell_overflow_rule():
    """more than MAX_PARTICLES_PER_CELL in one cell: the highest slots are killed."""
    rng = np.random.default_rng(21)
    xyz = rng.uniform(0.1, 4.9, (700, 3)).astype(np.float32)   # all in one cell; cap is 514
    more = cloud(2000, 22)
    g, o = make_pair(np.concatenate([xyz, more]), age=3.0, fert=1e6)
    g.step(1); o.step(1)
    compare_all(g, o, "overflow step 1")
    assert g.counters["cell_overflow_kills"] == o.counters["cell_overflow_kills"] > 0
    g.step(1); o.step(1)
    compare_all(g, o, "overflow step 2")


def test_chunk_list_capacity_rule():
    """More particles in one chunk than its list holds (MAX_PARTICLES_PER_CHUNK = 64 cells x 4 here;
    only possible while cells overflow, the count includes the ones the overflow rule kills):
    build_grid stores the first 256 ids in slot order and calc_forces walks the stored list
    (ps.cpp:1502-1508), so the tail is neither aged nor collided nor moved that step."""
    over = {"max_particles_num": 4096}
    cfg = ps.default_config(**over)
    G, cs = cfg.chunk_factor * cfg.chunk_dim, cfg.cell_size
    rng = np.random.default_rng(5)
    pts = []
    for i3 in range(4, 8):                 # chunk (1,1,1): 4 particles in its interior cells, more towards its corners
        for i1 in range(4, 8):
            for i2 in range(4, 8):
                k = {0: 4, 1: 6, 2: 7, 3: 9}[sum(v in (4, 7) for v in (i1, i2, i3))]
                for _ in range(k):
                    u = rng.uniform(0.05, 0.95, 3)
                    pts.append(((i2 - G / 2 + u[0]) * cs, -(i1 - G / 2 + u[1]) * cs, -(i3 - G / 2 + u[2]) * cs))
    rest = cloud(1500, 23)                 # ordinary traffic elsewhere (the segments around the chunk are full)
    idx = np.floor(rest.astype(np.float64) * [1, -1, -1] / cs).astype(int) + G // 2
    rest = rest[~((idx >= 2) & (idx <= 9)).all(1)]
    xyz = np.concatenate([np.array(pts, np.float32), rest])
    age = np.random.default_rng(6).uniform(2.2, 7.0, len(xyz)).astype(np.float32)
    g, o = make_pair(xyz, age=age, fert=1e6, **over)
    for k in range(4):
        g.step(1); o.step(1)
        if k == 0:
            assert o.chunkgrid[:, 0].max() > o.d.max_per_chunk, "the scenario must pass the chunk list's capacity"
            assert g.gridmax()[0] == o.d.max_per_chunk
        compare_all(g, o, "chunk capacity step %d" % (k + 1))
    assert o.counters["cell_overflow_kills"] > 100 and o.counters["integrated"] > 0


def test_cell_with_more_ids_than_the_lds_ranking_holds():
    """4105 particles in ONE cell of an 8-cell interior segment (capacity 8 x 514 = 4112 slots): more
    than the 4096 ids k_sort_cells ranks in LDS.  The 514 lowest ids stay, the rest is the overflow
    the reference kills (ps.cpp:1517-1526) -- found by bisection on the id value instead."""
    rng = np.random.default_rng(141)
    blob = rng.uniform(5.1, 9.9, (4105, 3)).astype(np.float32)        # cell (i1, i2, i3) = (6, 9, 6): inner cell of chunk (1, 2, 1)
    blob[:, 1] *= -1; blob[:, 2] *= -1
    rest = cloud(3000, 142)
    rest = rest[(np.abs(rest) > 20.0).any(axis=1)]
    xyz = np.concatenate([blob, rest])
    age = rng.uniform(2.2, 7.0, len(xyz)).astype(np.float32)
    g, o = make_pair(xyz, age=age, fert=1e6)
    g.init_iframe(); g.build_grid()
    assert g.download_cellgrid()[:, 0].max() == g.sizes.max_per_cell
    g.calc_forces()
    o.step(1)
    compare_all(g, o, "big cell step 1")
    assert o.counters["cell_overflow_kills"] == 4105 - g.sizes.max_per_cell
    g.step(1); o.step(1)
    compare_all(g, o, "big cell step 2")


def test_small_grids_and_empty():
    for over, n in (({"chunk_factor": 1, "chunk_dim": 3, "max_particles_num": 200}, 150),
                    ({"chunk_factor": 2, "chunk_dim": 3, "max_particles_num": 2000}, 1500),
                    ({"chunk_factor": 2, "chunk_dim": 5, "max_particles_num": 3000, "cell_size": 2.5}, 0)):
        cfg = ps.default_config(**over)
        G = cfg.chunk_factor * cfg.chunk_dim
        # odd grids are not centred: i = floor(+-c/CELL_SIZE) + G/2 with integer G/2 (app.cu:126-128)
        lo, hi = -(G // 2) * cfg.cell_size, (G - G // 2) * cfg.cell_size
        rng = np.random.default_rng(31)
        xyz = np.zeros((n, 3), np.float32)
        xyz[:, 0] = rng.uniform(lo, hi, n) * 0.999
        xyz[:, 1] = -rng.uniform(lo, hi, n) * 0.999
        xyz[:, 2] = -rng.uniform(lo, hi, n) * 0.999
        g, o = make_pair(xyz, age=3.0, fert=1e6, **over)
        for k in range(3):
            g.step(1); o.step(1)
            compare_all(g, o, "%r step %d" % (over, k + 1))


def test_explosions_counter_rng():
    """births with the counter-based RNG: exact against the oracle fed the same draws."""
    n = 20000
    xyz = cloud(n, 41)
    rng = np.random.default_rng(41)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = rng.uniform(2.5, 9.0, n).astype(np.float32)
    seed = 0xC0FFEE
    g, o = make_pair(xyz, age=age, fert=fert, flags=ps.FLAG_EXPLOSIONS, seed=seed)
    o.set_rng(explosion_rng(seed))
    for k in range(6):
        g.step(1); o.step(1)
        compare_all(g, o, "explosions step %d" % (k + 1))
    assert g.counters["births"] > 100


def test_upload_download_roundtrip():
    xyz = cloud(3000, 51)
    g, o = make_pair(xyz, age=3.0, fert=1e6)
    o.step(3)
    g2 = ps.ParticleSystem(ps.default_config())
    g2.upload_particles(o.particles.copy())
    g2.upload_queues(o.queue_info.copy(), o.queue.copy())
    assert_same_particles(g2.download_particles(), o.particles, "roundtrip")
    g2.step(2); o.step(2)
    assert_same_particles(g2.download_particles(), o.particles, "continued from uploaded state")
    qi, q = g2.download_queues()
    assert qi.tobytes() == o.queue_info.tobytes() and np.array_equal(q, o.queue)


def test_errors_are_statuses():
    g = ps.ParticleSystem(ps.default_config())
    with pytest.raises(ps.PsamdError) as e:
        g.fill_particles([[100.0, 0, 0]])
    assert e.value.status == 5          # PSAMD_ERR_OUTSIDE_BOX (ps.cpp:954-957)
    with pytest.raises(ps.PsamdError) as e:
        g.build_grid()
    assert e.value.status == 8          # PSAMD_ERR_STATE
    with pytest.raises(ps.PsamdError) as e:
        ps.ParticleSystem(ps.default_config(chunk_dim=2))
    assert e.value.status == 1
    # uploads are validated: id must equal the slot, live particles must be inside the box
    p = g.download_particles(0, 4)
    bad = p.copy(); bad["id"][2] = 7
    with pytest.raises(ps.PsamdError) as e:
        g.upload_particles(bad)
    assert e.value.status == 1
    bad = p.copy(); bad["cell"][1] = 5; bad["x"][1] = 1e6
    with pytest.raises(ps.PsamdError) as e:
        g.upload_particles(bad)
    assert e.value.status == 1
    # ... and inside the cell the record claims (the reference derives the cell from the position)
    bad = p.copy(); bad["cell"][1] = 5; bad["x"][1] = 0.0; bad["y"][1] = 0.0; bad["z"][1] = 0.0
    with pytest.raises(ps.PsamdError) as e:
        g.upload_particles(bad)
    assert e.value.status == 1
    good = p.copy(); good["cell"][1] = 8 * 256 + 8 * 16 + 8; good["x"][1] = 0.5; good["y"][1] = -0.5; good["z"][1] = -0.5
    good["w"][1] = 60.0
    g.upload_particles(good)            # cell 2184 = (i3, i1, i2) = (8, 8, 8) holds (0.5, -0.5, -0.5)
    # a free record is stored as a reset one, whatever else it carries
    junk = p.copy(); junk["x"][3] = 7.0; junk["age"][3] = 3.0
    g.upload_particles(junk)
    back = g.download_particles(0, 4)
    assert back["x"][3] == 0.0 and back["age"][3] == 0.0
    g.upload_particles(p)               # a valid upload afterwards is accepted


def test_large_grid_windowed_histogram():
    """BASELINE config 4's grid (40^3 = 64000 cells, more than one LDS histogram holds): the
    windowed hist/scatter kernels and the 1000-chunk scan, against the oracle."""
    over = {"chunk_factor": 10, "chunk_dim": 4, "max_particles_num": 200000}
    n = 150000
    xyz = cloud(n, 71, 100.0)
    rng = np.random.default_rng(71)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    g, o = make_pair(xyz, age=age, fert=1e6, **over)
    assert g.sizes.num_cells == 64000
    for k in range(3):
        g.step(1); o.step(1)
        compare_all(g, o, "40^3 grid step %d" % (k + 1))


def test_snapshot_restore_is_exact():
    xyz = cloud(20000, 81)
    g, o = make_pair(xyz, age=3.0, fert=1e6)
    g.step(2); o.step(2)
    g.snapshot_save()
    g.step(3)
    g.snapshot_restore()
    compare_all_state_only = assert_same_particles
    compare_all_state_only(g.download_particles(), o.particles, "restored")
    qi, q = g.download_queues()
    assert qi.tobytes() == o.queue_info.tobytes() and np.array_equal(q, o.queue)
    g.step(1); o.step(1)
    assert_same_particles(g.download_particles(), o.particles, "step after restore")


@pytest.mark.parametrize("eps2", [1e-20, 0.01, 0.37, 1.0, 3.0])
def test_other_softening_lengths(eps2):
    """EPS2 decides which arithmetic path the pair kernel takes: 1e-20 is outside the range
    the lean sqrt/rcp were checked on (generic compiler forms are used), the others pick a
    different fp32-add threshold (validated on the device when the context is created)."""
    n = 40000
    xyz = cloud(n, 101)
    rng = np.random.default_rng(101)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    g, o = make_pair(xyz, age=age, fert=1e6, eps2=eps2)
    for k in range(2):
        g.step(1); o.step(1)
        compare_all(g, o, "eps2=%g step %d" % (eps2, k + 1))


def test_kid_at_the_position_of_an_adult_with_vanishing_softening():
    """bodyBodyInteraction returns ai unchanged for a kid neighbour (app_common.cu:240-243).  The
    GPU encodes a kid as a body of mass 0 and multiplies -- fine while 1/sqrt(eps2^3) is finite.
    With EPS2 = 1e-20 (generic arithmetic) a kid at EXACTLY an adult's position would give
    0 * inf = NaN there; the generic walk skips such bodies instead, like the reference."""
    n = 6000
    xyz = cloud(n, 131)
    rng = np.random.default_rng(131)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    twins = xyz[:200].copy()                      # 200 kids, each exactly on top of an adult
    xyz = np.concatenate([xyz, twins])
    age = np.concatenate([age, np.full(200, 0.1, np.float32)])
    g, o = make_pair(xyz, age=age, fert=1e6, eps2=1e-20)
    for k in range(2):
        g.step(1); o.step(1)
        compare_all(g, o, "coincident kid step %d" % (k + 1))
    p = g.download_particles()
    live = p["cell"] >= 0
    assert np.isfinite(p["ax"][live]).all() and np.isfinite(p["x"][live]).all()


def test_big_segments_replay_in_global_memory():
    """chunk_dim = 8: interior segments hold 216 cells x 514 slots, more than the LDS queue
    window, so their free-slot queues are replayed in global memory."""
    over = {"chunk_factor": 2, "chunk_dim": 8}
    n = 60000
    xyz = cloud(n, 111)
    rng = np.random.default_rng(111)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    v = rng.uniform(-60, 60, (n, 3)).astype(np.float32)
    g, o = make_pair(xyz, age=age, fert=1e6, vxyz=v, **over)
    assert max(g.sizes.seg_size_t) > 6144
    for k in range(3):
        g.step(1); o.step(1)
        compare_all(g, o, "chunk_dim 8 step %d" % (k + 1))
    assert g.counters["relocations"] > 1000


@pytest.mark.parametrize("size", ["mid", "small", "big"])
def test_long_operation_lists(size):
    """Fast particles fill the corner segment at the box centre (its 8 cells are [-5,5)^3) and move
    one cell along every axis: 7/8 of them leave it in one step, while more, one cell further out,
    move INTO it.  Small: 3600 + 2000 particles, more queue operations on one record than a
    workgroup used to sort in LDS (4096; now 8192).  Big (twice the container, 8208-slot corner
    segments): 7200 + 4000, more than the in-LDS replay holds at all -- the radix sort of all keys and
    the streamed closed-form replay (k_replay) run; and with the queue nearly empty (third step) its
    serial walk.  Mid: 2000 + 1400, a list between 2048 and 4096 operations -- the long-list instance in the
    first step, and in the next, on that step's hint, the 4096-operation instance for every queue."""
    big = size == "big"
    rng = np.random.default_rng(121)
    nb, no = (7200, 4800) if big else (3600, 2400) if size == "small" else (2000, 1400)
    blob = rng.uniform(-4.9, 4.9, (nb, 3)).astype(np.float32)
    outer = rng.uniform(-9.9, -0.1, (no, 3)).astype(np.float32)
    outer = outer[~(outer > -5.0).all(axis=1)]                        # not the centre cell itself
    rest = cloud(5000, 122)
    rest = rest[(np.abs(rest) > 10.0).any(axis=1)]
    xyz = np.concatenate([blob, outer, rest])
    v = np.zeros_like(xyz)
    v[:len(blob) + len(outer)] = 300.0                                # clamped to one cell per step
    over = {"max_particles_num": 1 << 21} if big else {}
    g, o = make_pair(xyz, age=3.0, fert=1e6, vxyz=v, collision_radius=0.0, **over)
    for k in range(3):
        g.step(1); o.step(1)
        compare_all(g, o, "long list step %d" % (k + 1))
    assert g.counters["relocations"] > (9000 if big else 4500 if size == "small" else 2500)
    assert g.counters["max_ops_one_queue"] > (8192 if big else 4096 if size == "small" else 2048)      # big: the sorted path really ran
    if size == "mid":
        assert g.counters["max_ops_one_queue"] <= 4096


def test_long_free_run_with_births_and_collapse():
    """A free-running cloud dense enough to collapse at its surface: cells hit the list
    capacity (overflow kills into queue record 0), segments fill up (relocations and births
    that find no free slot), long operation lists -- every step compared in full."""
    n = 1 << 17
    xyz = cloud(n, 131)
    rng = np.random.default_rng(131)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = rng.uniform(2.5, 12.0, n).astype(np.float32)
    seed = 20261003
    # a tight container (list capacity 40 for a mean of 32 particles per cell, 320 slots per
    # 8-cell segment) overflows cells at once and fills segments as the cloud collapses
    g, o = make_pair(xyz, age=age, fert=fert, flags=ps.FLAG_EXPLOSIONS, seed=seed, max_particles_num=80000)
    o.set_rng(explosion_rng(seed))
    for k in range(10):
        g.step(1); o.step(1)
        compare_all(g, o, "free run step %d" % (k + 1))
    c = g.counters
    print("free run counters:", {k: v for k, v in c.items() if v})
    assert c["cell_overflow_kills"] > 0 and c["births"] > 0 and c["relocations"] > 0
    assert c["relocations_lost"] + c["births_failed"] > 0


def test_timing_levels_report_without_stalling_steps():
    """psamd_set_timing: level 1 brackets the pair pass, apply and the life cycle only; level 2
    every stage; intervals are accumulated over steps and the life-cycle one is read late."""
    xyz = cloud(20000, 91)
    g = ps.ParticleSystem(ps.default_config())
    g.fill_particles(xyz, age=np.float32(3.0), fert_age=np.full(len(xyz), 1e6, np.float32))
    g.step(1)
    g.set_timing(True)
    g.step(3)
    t, n = g.timing()
    assert n == 3 and t["pairs"] > 0 and t["apply"] > 0 and t["lifecycle"] > 0
    assert t["hist"] == 0 and t["sort_cells"] == 0
    g.set_timing(True, every_stage=True)
    g.step(2)
    t, n = g.timing()
    assert n == 2 and all(v > 0 for v in t.values()), t
    g.set_timing(False)
    g.step(1)
    assert g.timing()[1] == 0
    # every fourth step only (what bench.py's timed region does): steps 0, 4, 8 of ten carry the events
    g.set_timing(True, period=4)
    g.step(10)
    t, n = g.timing()
    assert n == 3 and t["pairs"] > 0 and t["lifecycle"] > 0
    per_step = t["pairs"] / n
    g.set_timing(True, period=1)
    g.step(4)
    t1, n1 = g.timing()
    assert n1 == 4 and 0.3 * per_step < t1["pairs"] / n1 < 3.0 * per_step
    g.close()


def test_two_pass_halo_overflow_and_kids():
    """Two-pass pair stage: (a) a cell whose neighbours crowd more bodies against its faces
    than its halo list holds falls back to scanning the whole stencil; (b) kids sit among
    colliding adults (no force, no collision, still integrated).  Both byte-exact."""
    rng = np.random.default_rng(123)
    # cell (8,8,8) of the 16^3 grid spans [0,5)^3 in (x, -y, -z); crowd its 26 neighbours' near sides
    centre = np.array([2.5, -2.5, -2.5], np.float32)
    pts = []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if dx == dy == dz == 0:
                    continue
                lo = np.array([2.5 + 5 * dx, -2.5 + 5 * dy, -2.5 + 5 * dz]) - 2.5
                p = rng.uniform(0.0, 5.0, (60, 3))
                for ax, d in enumerate((dx, dy, dz)):
                    if d == -1: p[:, ax] = rng.uniform(4.75, 4.999, 60)     # hug the face towards the centre cell
                    if d == +1: p[:, ax] = rng.uniform(0.001, 0.25, 60)
                pts.append(lo + p)
    crowd = np.concatenate(pts).astype(np.float32)                 # 26 * 60 = 1560 bodies within 0.25 of the cell
    inside = (centre + rng.uniform(-2.45, 2.45, (300, 3))).astype(np.float32)
    rest = cloud(20000, 124)
    xyz = np.concatenate([crowd, inside, rest])
    age = rng.choice(np.array([0.5, 1.4, 3.0, 3.0, 3.0, 6.0], np.float32), len(xyz))
    g, o = make_pair(xyz, age=age, fert=1e6)
    for k in range(3):
        g.step(1); o.step(1)
        compare_all(g, o, "halo overflow / kids step %d" % (k + 1))
    assert g.counters["deaths_collision"] > 0
    # the getter of the force pass's per-cell counts: never more than the cell holds
    g.init_iframe(); g.build_grid(); g.calc_forces_pairs()
    fc, cg = g.download_force_counts(), g.download_cellgrid()[:, 0]
    assert (fc <= cg).all() and 0 < fc.sum() < cg.sum()
    g.calc_forces_apply()
    g.close(); o.close()


def test_single_walk_pair_stage_still_matches(monkeypatch):
    """PSAMD_ONE_PASS keeps the lean arithmetic but settles collisions inside the force walk
    (what a configuration whose collision radius is large against the cell falls back to)."""
    monkeypatch.setenv("PSAMD_ONE_PASS", "1")
    xyz = cloud(30000, 131)
    rng = np.random.default_rng(131)
    age = rng.choice(np.array([0.5, 3.0, 3.0, 6.0, 15.5], np.float32), len(xyz))
    g, o = make_pair(xyz, age=age, fert=1e6)
    monkeypatch.delenv("PSAMD_ONE_PASS")
    for k in range(3):
        g.step(1); o.step(1)
        compare_all(g, o, "single-walk step %d" % (k + 1))
    g.init_iframe(); g.build_grid(); g.calc_forces_pairs()
    assert np.array_equal(g.download_force_counts(), g.download_cellgrid()[:, 0])   # every particle's sum is evaluated
    g.calc_forces_apply()
    g.close(); o.close()


def test_many_queue_records_grid_40():
    """A 40^3 grid (10 x 10 x 10 chunks of 4^3 cells): 64 000 cells and 9261 queue records -- more than
    the prefix arrays of the force pass's plan and the life cycle's per-workgroup scan of the queue census
    hold in LDS, so the global-memory forms of both run (k_plan_force's fallback, k_ops_scan +
    k_ops_scatter<false>).  Fast particles, three steps, every byte against the oracle."""
    n = 60000
    rng = np.random.default_rng(151)
    xyz = rng.uniform(-99.0, 99.0, (n, 3)).astype(np.float32)
    v = rng.uniform(-120, 120, (n, 3)).astype(np.float32)
    age = rng.uniform(2.2, 7.0, n).astype(np.float32)
    g, o = make_pair(xyz, age=age, fert=1e6, vxyz=v, chunk_factor=10)
    for k in range(3):
        g.step(1); o.step(1)
        compare_all(g, o, "40^3 grid step %d" % (k + 1))
    assert g.counters["relocations"] > 10000


@pytest.mark.parametrize("steps,radius", [(4, 0.4), (3, 2.0)])
def test_particles_whose_position_is_not_a_number(steps, radius):
    """A child born with the direction (0, 0, 0) gets the velocity 0/0 (ps.cpp:1306-1333) and, a step later, a
    position that is not a number; the reference files it under cell 0 from then on.  While it is a kid it is
    skipped by the force loop and the collision test (here: mass 0 in the snapshot, position replaced); once it
    is an adult, `dist > COLLISION_RADIUS` does not fail for it and every adult of its stencil that scans it
    collides with it -- and it, scanning, with every adult body of ITS stencil.  Two such particles (a kid and an
    adult) start among a dense crowd in the corner cells (0..1)^3; every byte against the oracle.  (Radius 2.0: the
    one-pass pair stage, where flags and forces come from the same walk.)"""
    rng = np.random.default_rng(171)
    n = 1500
    corner = np.stack([rng.uniform(-39.9, -30.1, n), rng.uniform(30.1, 39.9, n), rng.uniform(30.1, 39.9, n)], axis=1).astype(np.float32)
    rest = cloud(2000, 172)
    xyz = np.concatenate([corner, rest]).astype(np.float32)
    m = len(xyz)
    age = rng.uniform(2.0, 9.0, m).astype(np.float32)
    v = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    # the kid (its velocity is not a number from birth) and the adult, both well inside the box so that the first
    # step files them under cell 0 from somewhere else
    xyz[0] = (1.0, 2.0, 3.0); age[0] = 0.05; v[0] = np.nan
    xyz[1] = (-7.0, 4.0, -9.0); age[1] = 4.0; v[1] = np.nan
    g, o = make_pair(xyz, age=age, fert=1e6, vxyz=v, collision_radius=radius)
    for k in range(steps):
        g.step(1); o.step(1)
        compare_all(g, o, "not-a-number particles, step %d" % (k + 1))
    p = o.particles
    assert np.isnan(p["x"][p["cell"] >= 0]).sum() >= 1 and (p["cell"] == 0).sum() >= 1
    assert o.counters["deaths_collision"] + o.counters["survives"] > 100          # the crowd met the adult
    # the reference's buffers out of this context and into a fresh one: such a particle is valid state, where it is filed
    part, (qi, q) = g.download_particles(), g.download_queues()
    g2 = ps.ParticleSystem(g.cfg)
    g2.upload_particles(part); g2.upload_queues(qi, q)
    g2.step(1); o.step(1)
    assert_same_particles(g2.download_particles(), o.particles, "after the upload into a fresh context")
    g2.close()
