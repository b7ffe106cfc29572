"""Two PROCESSES, one rank each, sharing the box's one GPU: the sharded pair pass (with its
split-task legs) runs on the HIP path in both, the float4 shards are exchanged with a real
collective (gloo, through host memory -- RCCL refuses two ranks on one device), integrate and
life cycle are replicated, and both ranks must end bit-identical to the oracle.  This is
bench.py's multi-GPU loop (particlesystem_amd.sharded.step_sharded) with only the transport
swapped."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, STEPS, SEED = 60000, 6, 77


def inputs():
    rng = np.random.default_rng(SEED)
    xyz = rng.uniform(-39.9, 39.9, (N, 3)).astype(np.float32)
    age = rng.uniform(15 / 7, 7.5, N).astype(np.float32)
    fert = (1e6 + np.arange(N)).astype(np.float32)
    return xyz, age, fert


class HostStagedRank:
    """Stage interface of ParticleSystem whose force4 shard travels through a CPU tensor."""

    def __init__(self, g, force):
        self.g, self.force = g, force

    def init_iframe(self):
        self.g.init_iframe()

    def build_grid(self):
        self.g.build_grid()

    def force_shard(self):
        self.lo, self.hi, self.share = self.g.force_shard()
        return self.lo, self.hi, self.share

    def calc_forces_pairs(self):
        self.g.calc_forces_pairs()
        if self.hi > self.lo:
            self.force.numpy()[self.lo:self.hi] = self.g.download_force4(self.lo, self.hi - self.lo)

    def calc_forces_apply(self):
        world = int(os.environ["WORLD_SIZE"])
        self.g.upload_force4(self.force.numpy()[: world * self.share], 0)
        self.g.calc_forces_apply()


def _worker():
    import torch
    import torch.distributed as dist
    import particlesystem_amd as ps
    from particlesystem_amd.sharded import step_sharded
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_driver import digest
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = ps.ParticleSystem(ps.default_config(device=0, rank=rank, world=world))
    xyz, age, fert = inputs()
    g.fill_particles(xyz, age=age, fert_age=fert)
    force = torch.zeros((g.sizes.container_size + world, 4), dtype=torch.float32)
    r = HostStagedRank(g, force)
    for _ in range(STEPS):
        step_sharded(r, force, dist, rank, world)
    print("DIGEST %d %016x" % (rank, digest(g.download_particles())), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    g.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_processes_share_the_pair_pass_and_match_the_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_host_driver import digest
    from util import O
    xyz, age, fert = inputs()
    o = O.System(O.default_config())
    o.fill(xyz, age=age, fert_age=fert)
    o.step(STEPS)
    want = "%016x" % digest(o.particles)
    assert o.counters["relocations"] > 0 and o.counters["deaths_collision"] > 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=560)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    got = [line.split()[2] for out in outs for line in out.splitlines() if line.startswith("DIGEST")]
    assert got == [want, want], (got, want, outs)


if __name__ == "__main__" and "--worker" in sys.argv:
    _worker()
