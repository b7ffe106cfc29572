"""The CPU oracle against the committed golden vectors (tests/golden/*.npz), which
were produced by the reference's own L4 code (tests/golden/make_golden.py).
Bit-exact everywhere: integers by ==, floats by their bit patterns."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_py as O

IP = C.POINTER(C.c_int)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.fixture(scope="module")
def cfg():
    c = O.default_config()
    return c, O.derive(c)


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_constants(golden_dir, cfg):
    c, d = cfg
    g = load(golden_dir, "ref_constants.npz")
    layout, ints, reals = g["layout"], g["ints"], g["reals"]
    assert layout[0] == O.P_DTYPE.itemsize == 72
    names = ["id", "cell", "chunk", "seg_type", "seg_tid", "seg_fault", "is_parent",
             "w", "age", "fertility_age", "x", "vx", "ax"]
    for k, n in enumerate(names):
        assert layout[1 + k] == O.P_DTYPE.fields[n][1], n
    assert (layout[14], layout[15], layout[16]) == (24, 24, 8)
    mine = [c.max_particles_num, c.x_factor, c.chunk_factor, c.chunk_dim, d.grid_dim, d.num_cells,
            d.num_chunks, d.cells_per_chunk, d.max_per_cell, d.max_per_chunk, d.max_neib_particles,
            *d.seg_cells, *d.seg_count, *d.seg_size_t, *d.seg_size, d.container_size, d.queue_info_size]
    assert list(ints[:len(mine)]) == mine
    mine_r = [c.cell_size, c.eps2, c.collision_radius, c.particle_weight, c.dt, d.particle_life,
              d.kid_age, d.min_fertility_age, d.max_fertility_age, d.min_adult_age, d.max_adult_age,
              d.max_dx, c.max_v, c.explosion_speed]
    assert np.array_equal(np.array(mine_r).view(np.uint64), reals[:len(mine_r)].view(np.uint64))


def test_tables(golden_dir, cfg):
    c, d = cfg
    L = O.lib()
    g = load(golden_dir, "ref_tables.npz")
    out3 = (C.c_int * 3)()
    out27 = (C.c_int * 27)()
    for cell in range(d.num_cells):
        L.pso_get_cell_info(C.byref(d), C.byref(c), cell, out3)
        assert list(out3) == list(g["cell_info"][cell]), cell
        n = L.pso_fill_cells(C.byref(d), cell, out27)
        assert n == g["neib_n"][cell]
        assert list(out27)[:n] == list(g["neib"][cell][:n]), cell
    pk = np.zeros(27, dtype=O.PAIR_DTYPE)
    for ch in range(d.num_chunks):
        L.pso_set_pkg_segments(C.byref(c), ch, pk.ctypes.data)
        assert np.array_equal(np.stack([pk["c"], pk["p"]], 1).ravel(), g["pkg"][ch])
    for t, tid, want in g["cont_rloc"]:
        assert L.pso_get_cont_rloc(C.byref(d), int(t), int(tid)) == want
    for t, tid, want in g["info_rloc"]:
        assert L.pso_get_info_rloc(C.byref(d), int(t), int(tid)) == want
    out2 = (C.c_int * 2)()
    for i, want in zip(g["ids"], g["id_info"]):
        L.pso_get_id_info(C.byref(d), int(i), out2)
        assert list(out2) == list(want)


def _blank(n):
    p = np.zeros(n, dtype=O.P_DTYPE)
    for f in ("cell", "chunk", "seg_type", "seg_tid"):
        p[f] = -1
    return p


def test_set_pos(golden_dir, cfg):
    c, d = cfg
    L = O.lib()
    g = load(golden_dir, "ref_setpos.npz")
    pos = g["pos"]
    n = len(pos)
    px, pi = _blank(n), _blank(n)
    px["seg_type"], px["seg_tid"] = g["in_seg_type"], g["in_seg_tid"]
    for k in range(n):
        L.pso_set_pos_x(C.byref(c), C.byref(d), px[k:k + 1].ctypes.data, *map(float, pos[k]))
        L.pso_set_pos_i(C.byref(c), C.byref(d), pi[k:k + 1].ctypes.data, *map(float, pos[k]))
    for pref, arr in (("x_p_", px), ("i_p_", pi)):
        for f in ("cell", "chunk", "seg_type", "seg_tid", "seg_fault"):
            assert np.array_equal(arr[f], g[pref + f]), pref + f
        for f in ("x", "y", "z"):
            assert np.array_equal(bits(arr[f]), bits(g[pref + f])), pref + f
    # the SURVEY.md 8(c) G5 example survives the round trip
    k = 640
    assert tuple(pos[k]) == (41.0, np.float32(-3.0), np.float32(0.1))
    assert (px["x"][k], px["cell"][k], px["chunk"][k]) == (-39.0, 1920, 24)


def _pairs(g):
    n = len(g["bi_id"])
    bi = _blank(n)
    bj = np.zeros(n, dtype=O.T_DTYPE)
    for f in ("id", "age", "x", "y", "z"):
        bi[f] = g["bi_" + f]
    bi["w"] = 60.0
    for f in ("id", "age", "w", "x", "y", "z"):
        bj[f] = g["bj_" + f]
    return bi, bj


def test_pair_kernels(golden_dir, cfg):
    c, d = cfg
    L = O.lib()
    g = load(golden_dir, "ref_pairs.npz")
    bi, bj = _pairs(g)
    acc = g["acc_in"].copy()
    flags = np.zeros(len(bi), np.int32)
    for k in range(len(bi)):
        L.pso_body_body_interaction(C.byref(c), C.byref(d), bi[k:k + 1].ctypes.data,
                                    bj[k:k + 1].ctypes.data, acc[k:k + 1].ctypes.data)
        flags[k] = L.pso_body_body_collision(C.byref(c), C.byref(d), bi[k:k + 1].ctypes.data,
                                             bj[k:k + 1].ctypes.data)
    assert np.array_equal(bits(acc), bits(g["acc_out"]))
    assert np.array_equal(flags, g["flags"])
    assert set(np.unique(flags)) == {0, 1, 2}


def test_serial_accumulation_order(golden_dir, cfg):
    """~6.9k fp32 terms added one by one: only the reference's order reproduces the bits."""
    c, d = cfg
    L = O.lib()
    g = load(golden_dir, "ref_accumulate.npz")
    for k in range(int(g["n"])):
        me = _blank(1)
        me["id"] = g["me%d" % k][0]
        me["x"], me["y"], me["z"], me["age"] = g["mepos%d" % k]
        nb = np.zeros(len(g["nbid%d" % k]), dtype=O.T_DTYPE)
        nb["id"] = g["nbid%d" % k]
        for a, f in enumerate(("x", "y", "z", "w", "age")):
            nb[f] = g["nb%d" % k][:, a]
        acc = np.zeros(3, np.float32)
        for j in range(len(nb)):
            if nb["id"][j] != me["id"][0]:
                L.pso_body_body_interaction(C.byref(c), C.byref(d), me.ctypes.data,
                                            nb[j:j + 1].ctypes.data, acc.ctypes.data)
        assert np.array_equal(bits(acc), bits(g["acc%d" % k])), k


def test_queue_script(golden_dir, cfg):
    c, d = cfg
    L = O.lib()
    g = load(golden_dir, "ref_queue.npz")
    s = O.System()
    qi, q = s.queue_info, s.queue
    got = np.zeros(len(g["script"]), np.int32)
    for k, (op, t, tid, x) in enumerate(g["script"]):
        if op == 1:
            L.pso_q_insert(qi.ctypes.data, q.ctypes.data, C.byref(d), int(t), int(tid), int(x))
        else:
            got[k] = L.pso_q_remove(qi.ctypes.data, q.ctypes.data, C.byref(d), int(t), int(tid))
    assert np.array_equal(got, g["results"])
    assert (got == -1).sum() > 0  # the underflow branch was exercised
    fin = np.stack([qi[n] for n in O.Q_DTYPE.names], 1)[g["touched"]]
    assert np.array_equal(fin, g["final_info"])
    h = int(np.bitwise_xor.reduce((q.astype(np.int64) + 1) *
                                  (np.arange(len(q), dtype=np.int64) * 2654435761 % (1 << 31))))
    assert h == int(g["final_queue_hash"][0])
    s.close()
