"""The N>1 path on CPU: (1) the cut of calc_forces into a pair pass and an apply pass
is exact, (2) a world_size-2 gloo run of particlesystem_amd.sharded.step_sharded --
the orchestration bench.py uses with RCCL -- reproduces the serial reference.  The
compute stand-in here is the oracle (the product itself has no CPU path)."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as O
from particlesystem_amd.sharded import shard_bounds, step_sharded
from util import g2_cloud

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def state_hash(o):
    h = hashlib.sha256()
    h.update(o.particles.tobytes())
    h.update(o.queue_info.tobytes())
    h.update(o.queue.tobytes())
    return h.hexdigest()


def make(dt=0.05):
    xyz = g2_cloud()
    o = O.System(dt=dt)
    o.fill(xyz, age=np.float32(40 * dt), fert_age=(1e6 + np.arange(len(xyz))).astype(np.float32))
    return o


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 4096, 1048576, 1048577):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi, share = shard_bounds(total, r, world)
                assert hi - lo <= share and world * share >= total
                got += list(range(lo, hi)) if total < 5000 else []
            if total < 5000:
                assert got == list(range(total))


def test_pair_pass_plus_apply_is_calc_forces():
    a, b = make(), make()
    for step in range(12):
        a.step(1)
        b.init_iframe(); b.build_grid()
        n = b.sorted_count()
        f = np.zeros((n + 8, 4), np.float32)
        cut = n // 3                      # any split of the pair pass gives the same forces
        b.calc_pairs(cut, n, f)
        b.calc_pairs(0, cut, f)
        b.apply_forces(f)
        assert state_hash(a) == state_hash(b), step
    assert a.counters["relocations"] > 0 and a.counters["deaths_collision"] > 0


def test_threaded_pair_pass_matches_serial():
    o = make()
    o.step(3)
    o.init_iframe(); o.build_grid()
    n = o.sorted_count()
    want = np.zeros((n + 8, 4), np.float32)
    got = np.zeros_like(want)
    o.calc_pairs(0, n, want)
    o.calc_pairs_threads(0, n, got, 5)
    assert want.tobytes() == got.tobytes()
    part = np.zeros_like(want)
    o.calc_pairs_threads(100, n - 77, part, 3)          # a sub-range leaves the rest untouched
    assert part[100:n - 77].tobytes() == want[100:n - 77].tobytes()
    assert not part[:100].any() and not part[n - 77:].any()


class OracleRank:
    """Stage interface of ParticleSystem on top of the oracle, for one rank."""

    def __init__(self, o, force, rank, world):
        self.o, self.force, self.rank, self.world = o, force, rank, world

    def init_iframe(self):
        self.o.init_iframe()

    def build_grid(self):
        self.o.build_grid()

    def force_shard(self):
        return shard_bounds(self.o.sorted_count(), self.rank, self.world)

    def calc_forces_pairs(self):
        lo, hi, _ = self.force_shard()
        self.o.calc_pairs(lo, hi, self.force.numpy())

    def calc_forces_apply(self):
        self.o.apply_forces(self.force.numpy())


def _worker():
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = make()
    force = torch.zeros((4096 + 64, 4), dtype=torch.float32)
    sysr = OracleRank(o, force, rank, world)
    for _ in range(int(os.environ["PS_STEPS"])):
        step_sharded(sysr, force, dist, rank, world)
    print("HASH %d %s" % (rank, state_hash(o)), flush=True)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world_size_2_gloo_matches_serial():
    steps = 8
    ref = make()
    ref.step(steps)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", PS_STEPS=str(steps),
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=280)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    hashes = [line.split()[2] for out in outs for line in out.splitlines() if line.startswith("HASH")]
    assert len(hashes) == 2 and hashes[0] == hashes[1] == state_hash(ref)


if __name__ == "__main__" and "--worker" in sys.argv:
    _worker()
