"""The oracle's deferred life cycle (pso_apply_collect + pso_replay_ops) is the serial
calc_forces: same particles, same queues, same counters, every step.  It exists so that the
queue operations of one step can be collected on several slab-holding systems and replayed by
each queue's owner (tests/oracle_slab.py, the CPU stand-in of the multi-GPU path)."""
import numpy as np

import oracle_py as O
from util import cloud, explosion_rng, g2_cloud


def same(a, b, what):
    assert a.particles.tobytes() == b.particles.tobytes(), what + ": particles"
    assert a.queue_info.tobytes() == b.queue_info.tobytes() and np.array_equal(a.queue, b.queue), what + ": queues"
    ca, cb = a.counters, b.counters
    assert ca == cb, (what, ca, cb)


def deferred_step(o):
    o.init_iframe(); o.build_grid()
    n = o.sorted_count()
    f = np.zeros((n + 8, 4), np.float32)
    o.calc_pairs(0, n, f)
    ops = o.apply_collect(f)
    rng = np.random.default_rng(len(ops))
    o.replay_ops(ops[rng.permutation(len(ops))])      # any order in: the replay sorts by (queue, key)
    o.advance_step()
    return ops


def test_collect_plus_replay_is_calc_forces():
    for dt in (0.01, 0.05):
        xyz = g2_cloud()
        fert = (1e6 + np.arange(len(xyz))).astype(np.float32)
        a, b = O.System(dt=dt), O.System(dt=dt)
        for o in (a, b):
            o.fill(xyz, age=np.float32(40 * dt), fert_age=fert)
        kinds = set()
        for step in range(40):
            a.step(1)
            kinds |= set(np.unique(deferred_step(b)["kind"]))
            same(a, b, "dt=%g step %d" % (dt, step + 1))
        assert kinds >= {0, 1} and a.counters["relocations"] > 0 and a.counters["deaths_collision"] > 0


def test_deferred_births_and_dense_cloud():
    n = 30000
    xyz = cloud(n, 5)
    rng = np.random.default_rng(5)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = rng.uniform(2.5, 8.0, n).astype(np.float32)
    a, b = O.System(), O.System()
    for o in (a, b):
        o.fill(xyz, age=age, fert_age=fert)
        o.set_rng(explosion_rng(99))
    for step in range(6):
        a.step(1)
        ops = deferred_step(b)
        same(a, b, "births step %d" % (step + 1))
    assert a.counters["births"] > 100 and 2 in set(np.unique(ops["kind"])) | {2}
