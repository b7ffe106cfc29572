"""Host logic of the product without a GPU: psamd_describe (sizes, cell -> segment table,
chunk package table, initial free-slot queues) against the oracle over many
configurations, including randomly drawn ones."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

import oracle_py as O
import particlesystem_amd as ps
from util import oracle_cfg_from


def check(cfg):
    sizes, table, pkg, qi, q = ps.describe(cfg)
    oc = oracle_cfg_from(cfg)
    od = O.derive(oc)
    assert (sizes.grid_dim, sizes.num_cells, sizes.num_chunks, sizes.cells_per_chunk, sizes.max_per_cell,
            sizes.max_per_chunk, sizes.container_size, sizes.queue_info_size) == \
        (od.grid_dim, od.num_cells, od.num_chunks, od.cells_per_chunk, od.max_per_cell, od.max_per_chunk,
         od.container_size, od.queue_info_size)
    assert list(sizes.seg_count) == list(od.seg_count) and list(sizes.seg_size_t) == list(od.seg_size_t)
    assert sizes.n_cellgrid == od.num_cells * (1 + od.max_per_cell)
    L = O.lib()
    out3 = (C.c_int * 3)()
    for c in range(od.num_cells):
        L.pso_get_cell_info(C.byref(od), C.byref(oc), c, out3)
        assert tuple(out3) == tuple(table[c]), (c,)
    pk = np.zeros(27, dtype=O.PAIR_DTYPE)
    for ch in range(od.num_chunks):
        L.pso_set_pkg_segments(C.byref(oc), ch, pk.ctypes.data)
        assert np.array_equal(np.stack([pk["c"], pk["p"]], 1).ravel(), pkg[ch]), ch
    s = O.System(oc)
    assert qi.tobytes() == s.queue_info.tobytes()
    assert np.array_equal(q, s.queue)
    s.close()


@pytest.mark.parametrize("over", [
    {},                                                            # the reference's shipped constants
    {"chunk_factor": 2, "chunk_dim": 3, "max_particles_num": 1000},
    {"chunk_factor": 1, "chunk_dim": 3, "max_particles_num": 50},
    {"chunk_factor": 3, "chunk_dim": 5, "max_particles_num": 7777, "x_factor": 3},
    {"chunk_factor": 6, "chunk_dim": 4, "max_particles_num": 200000},
    {"chunk_factor": 2, "chunk_dim": 8, "max_particles_num": 100000},
])
def test_describe_matches_oracle(over):
    check(ps.default_config(**over))


@settings(max_examples=25, deadline=None)
@given(f=st.integers(1, 5), d=st.integers(3, 7), n=st.integers(1, 300000), x=st.integers(1, 3))
def test_describe_random_configs(f, d, n, x):
    check(ps.default_config(chunk_factor=f, chunk_dim=d, max_particles_num=n, x_factor=x))


def test_describe_rejects_bad_configs():
    lib = ps.load()
    for over in ({"chunk_dim": 2}, {"chunk_factor": 0}, {"max_particles_num": 0}, {"cell_size": 0.0}, {"dt": -1.0}):
        cfg = ps.default_config(**over)
        assert lib.psamd_describe(C.byref(cfg), None, None, None, None, None) == 1
