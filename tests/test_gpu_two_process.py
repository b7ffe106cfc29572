"""Two PROCESSES, one rank each, sharing the box's one GPU: each holds its slab of the system
on the HIP path, the halo / force / transfer messages travel through a real process group
(gloo, staged through host memory -- RCCL refuses two ranks on one device) with
particlesystem_amd.slab.HostRing, i.e. bench.py's multi-GPU loop with only the transport
swapped.  After 12 steps the union of the two ranks must be the oracle's state, byte for byte."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, STEPS, SEED = 60000, 12, 77


def inputs():
    rng = np.random.default_rng(SEED)
    xyz = rng.uniform(-39.9, 39.9, (N, 3)).astype(np.float32)
    age = rng.uniform(15 / 7, 7.5, N).astype(np.float32)
    fert = (1e6 + np.arange(N)).astype(np.float32)
    return xyz, age, fert


def _worker():
    import torch.distributed as dist
    import particlesystem_amd as ps
    from particlesystem_amd.slab import HostRing
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = ps.ParticleSystem(ps.default_config(device=0, rank=rank, world=world))
    xyz, age, fert = inputs()
    g.fill_particles(xyz, age=age, fert_age=fert)
    ring = HostRing(g, dist, rank, world)
    sent = 0
    for _ in range(STEPS):
        ring.step()
        sent += sum(int(g.msg_download(ps.MSG_XFER_OUT + k, 64)[0]) for k in (0, 1))
    qi, q = g.download_queues()
    np.savez(os.environ["PS_OUT"] + ".%d.npz" % rank, p=g.download_particles(), qi=qi, q=q, sent=sent,
             reloc=g.counters["relocations"])
    dist.barrier()
    dist.destroy_process_group()
    g.close()


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_two_processes_hold_one_slab_each_and_match_the_oracle(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import particlesystem_amd as ps
    from particlesystem_amd.slab import merge_owned
    from util import O, assert_same_particles
    xyz, age, fert = inputs()
    o = O.System(O.default_config())
    o.fill(xyz, age=age, fert_age=fert)
    o.step(STEPS)
    assert o.counters["relocations"] > 0 and o.counters["deaths_collision"] > 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", WORLD_SIZE="2", PS_OUT=str(tmp_path / "rank"),
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=560)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    got = [np.load(str(tmp_path / "rank") + ".%d.npz" % r) for r in range(2)]
    plans = [ps.slab_plan(ps.default_config(rank=r, world=2)) for r in range(2)]
    assert_same_particles(merge_owned([g["p"] for g in got], plans), o.particles, "two processes, %d steps" % STEPS)
    assert merge_owned([g["qi"] for g in got], plans, "records").tobytes() == o.queue_info.tobytes()
    assert np.array_equal(merge_owned([g["q"] for g in got], plans), o.queue)
    assert sum(int(g["sent"]) for g in got) > 0                       # particles changed owner
    assert sum(int(g["reloc"]) for g in got) == o.counters["relocations"]


if __name__ == "__main__" and "--worker" in sys.argv:
    _worker()
