"""Stage sequences as hipGraphs (psamd_set_graphs): the same kernels, one submission per stage.  What must hold:
every byte equals the oracle's with graphs on -- also with births, whose RNG is keyed by the step's number
(ps.cpp:1306-1333: the number lives in device memory, a replayed graph carries no step-dependent argument), across a
snapshot restore (which rewinds that number), and on steps that carry timing events (those run as plain launches)."""
import numpy as np
import pytest

import particlesystem_amd as ps
from util import O, assert_same_particles, explosion_rng, g2_cloud, oracle_cfg_from

pytestmark = pytest.mark.gpu


def start(n_extra=0, seed=11, **over):
    xyz = g2_cloud()
    rng = np.random.default_rng(seed)
    age = rng.uniform(2.0, 9.0, len(xyz)).astype(np.float32)
    fert = rng.uniform(3.0, 12.0, len(xyz)).astype(np.float32)
    v = rng.uniform(-20, 20, xyz.shape).astype(np.float32)
    g = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_EXPLOSIONS, seed=seed, **over))
    o = O.System(oracle_cfg_from(g.cfg))
    o.set_rng(explosion_rng(seed))
    ids = o.fill(xyz, age=age, fert_age=fert)
    p = o.particles
    p["vx"][ids], p["vy"][ids], p["vz"][ids] = v.T
    g.fill_particles(xyz, age=age, fert_age=fert, vxyz=v)
    return g, o


def same(g, o, what):
    assert_same_particles(g.download_particles(), o.particles, what)
    qi, q = g.download_queues()
    assert qi.tobytes() == o.queue_info.tobytes() and np.array_equal(q, o.queue), what + ": queues differ"


def test_step_as_one_graph_equals_the_oracle_with_births():
    g, o = start()
    g.set_graphs(True)
    for k in range(1, 31):
        g.step(1); o.step(1)
        if k % 5 == 0:
            same(g, o, "step %d" % k)
    assert o.counters["births"] > 300 and o.counters["relocations"] > 1000
    replays, captures = g.graph_stats()
    assert replays == 30 and 1 <= captures <= 4, (replays, captures)      # (a shape is captured when it is first met)
    g.close(); o.close()


def test_graphs_across_snapshot_restore_and_timing_steps():
    g, o = start(seed=12)
    g.set_graphs(True)
    g.step(4); o.step(4)
    g.snapshot_save()
    g.step(6); o.step(6)
    same(g, o, "ten steps")
    want = g.download_particles()
    g.snapshot_restore()                 # rewinds the step's number on the device too: the same births come again
    g.set_timing(True, period=2)         # every other step carries events and runs as plain launches
    g.step(6)
    assert g.download_particles().tobytes() == want.tobytes()
    tim, launches = g.timing()
    assert launches == 3 and tim["pairs"] > 0
    g.set_timing(False)
    g.set_graphs(False)                  # and off again: plain launches from here on
    g.step(3); o.step(3)
    same(g, o, "thirteen steps")
    g.close(); o.close()


def test_a_stage_call_between_graph_steps_keeps_the_numbers_in_step():
    """the single stage calls (the reference's task 3 / 8 / 6 one by one) never replay a graph; mixed with psamd_step
    the step's number and the scalar records' sequence stay consistent"""
    g, o = start(seed=13)
    g.set_graphs(True)
    for k in range(6):
        if k % 2:
            g.init_iframe(); g.build_grid(); g.calc_forces()
        else:
            g.step(1)
        o.step(1)
    same(g, o, "six steps")
    g.close(); o.close()
