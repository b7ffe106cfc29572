"""Live comparison of the CPU oracle with the reference's own L4 code on fresh random
inputs (container only: needs oracle/_ref/libref_l4.so, which is built from
/root/reference and is absent on the GPU box)."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.skipif(not O.have_ref(), reason="reference L4 library not built here")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _blank(n):
    p = np.zeros(n, dtype=O.P_DTYPE)
    for f in ("cell", "chunk", "seg_type", "seg_tid"):
        p[f] = -1
    return p


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_pairs_random(seed):
    rng = np.random.default_rng(seed)
    c = O.default_config()
    d = O.derive(c)
    L, R = O.lib(), O.ref()
    n = 20000
    bi, bj = _blank(n), np.zeros(n, dtype=O.T_DTYPE)
    bi["id"] = rng.integers(0, d.container_size, n)
    bj["id"] = rng.integers(0, d.container_size, n)
    for f in ("x", "y", "z"):
        bi[f] = rng.uniform(-40, 40, n).astype(np.float32)
        bj[f] = (bi[f] + rng.normal(scale=rng.choice([0.2, 3.0, 10.0]), size=n)).astype(np.float32)
    bi["age"] = rng.uniform(0, 16, n).astype(np.float32)
    bj["age"] = rng.uniform(0, 16, n).astype(np.float32)
    bj["w"] = rng.uniform(0, 100, n).astype(np.float32)
    a_ref = rng.normal(size=(n, 3)).astype(np.float32)
    a_me = a_ref.copy()
    f_ref = np.zeros(n, np.int32)
    R.ref_body_body_interaction(n, bi.ctypes.data, bj.ctypes.data, a_ref.ctypes.data)
    R.ref_body_body_collision(n, bi.ctypes.data, bj.ctypes.data, f_ref.ctypes.data)
    f_me = np.zeros(n, np.int32)
    for k in range(n):
        L.pso_body_body_interaction(C.byref(c), C.byref(d), bi[k:k + 1].ctypes.data,
                                    bj[k:k + 1].ctypes.data, a_me[k:k + 1].ctypes.data)
        f_me[k] = L.pso_body_body_collision(C.byref(c), C.byref(d), bi[k:k + 1].ctypes.data,
                                            bj[k:k + 1].ctypes.data)
    assert np.array_equal(bits(a_me), bits(a_ref))
    assert np.array_equal(f_me, f_ref)


def test_set_pos_random():
    rng = np.random.default_rng(7)
    c = O.default_config()
    d = O.derive(c)
    L, R = O.lib(), O.ref()
    n = 20000
    pos = rng.uniform(-130, 130, (n, 3)).astype(np.float32)
    a, b = _blank(n), _blank(n)
    a["seg_type"] = b["seg_type"] = rng.choice([-1, 1, 2, 4, 8], n)
    a["seg_tid"] = b["seg_tid"] = rng.integers(-1, 64, n)
    for k in range(n):
        R.ref_set_pos_x(a[k:k + 1].ctypes.data, *map(float, pos[k]))
        L.pso_set_pos_x(C.byref(c), C.byref(d), b[k:k + 1].ctypes.data, *map(float, pos[k]))
    assert a.tobytes() == b.tobytes()


def test_set_pos_of_what_is_no_number():
    """A position that is not a number (a child born with the direction (0, 0, 0) has one a step later): the
    reference's set_pos_t converts floor(NaN) + G/2 to int -- INT_MIN on this host's cvttsd2si -- and its wrap loop
    walks that to cell index 0 on each axis of the default 16^3 grid (2^31 is a multiple of 16).  The oracle, the
    reference's compiled code and, through them, the kernels' conversion (k_apply) agree; so do +-inf and values far
    outside the box."""
    c = O.default_config()
    d = O.derive(c)
    L, R = O.lib(), O.ref()
    nan, inf = float("nan"), float("inf")
    cases = [(nan, nan, nan), (nan, 1.0, -2.0), (3.0, nan, 7.0), (1.0, 2.0, nan), (inf, 0.0, 0.0), (0.0, -inf, 0.0), (1e30, -1e30, 5.0)]
    a, b = _blank(len(cases)), _blank(len(cases))
    a["seg_type"] = b["seg_type"] = 1
    a["seg_tid"] = b["seg_tid"] = 0
    for k, pos in enumerate(cases):
        R.ref_set_pos_x(a[k:k + 1].ctypes.data, *pos)
        L.pso_set_pos_x(C.byref(c), C.byref(d), b[k:k + 1].ctypes.data, *pos)
    for f in ("cell", "chunk", "seg_type", "seg_tid", "seg_fault"):
        assert np.array_equal(a[f], b[f]), (f, a[f], b[f])
    assert np.array_equal(bits(np.stack([a["x"], a["y"], a["z"]])), bits(np.stack([b["x"], b["y"], b["z"]])))
    assert a["cell"][0] == 0 and (a["cell"] >= 0).all() and (a["cell"] < 4096).all()


def test_neighbour_gather_matches_reference():
    """fill_cells + fill_particles over a real cell grid == the oracle's gather order."""
    rng = np.random.default_rng(11)
    s = O.System()
    xyz = rng.uniform(-40, 40, (20000, 3)).astype(np.float32)
    s.fill(xyz, age=2.0, fert_age=1e6)
    s.init_iframe()
    s.build_grid()
    R = O.ref()
    cg = s.cellgrid
    out27 = (C.c_int * 27)()
    buf = np.zeros(s.d.max_neib_particles, np.int32)
    for cell in rng.integers(0, s.d.num_cells, 200):
        n = R.ref_fill_particles(int(cell), cg.ctypes.data, buf.ctypes.data, len(buf))
        m = O.lib().pso_fill_cells(C.byref(s.d), int(cell), out27)
        mine = np.concatenate([cg[out27[i], 1:1 + cg[out27[i], 0]] for i in range(m)] or [np.zeros(0, np.int32)])
        assert n == len(mine)
        assert np.array_equal(buf[:n], mine)
    s.close()


def test_particle_state_helpers():
    R = O.ref()
    L = O.lib()
    rng = np.random.default_rng(5)
    raw = rng.integers(0, 255, 72 * 4, dtype=np.uint8)
    a = raw.copy().view(O.P_DTYPE)
    b = raw.copy().view(O.P_DTYPE)
    R.ref_reset_particle(a[0:1].ctypes.data)
    L.pso_reset_particle(b[0:1].ctypes.data)
    R.ref_survive_particle(a[1:2].ctypes.data)
    L.pso_survive_particle(b[1:2].ctypes.data)
    assert a.tobytes() == b.tobytes()
