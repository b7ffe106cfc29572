"""The N>1 path on CPU.  (1) The slab plan: every cell layer, slot and queue record has exactly
one owner, neighbours hold what a rank reads.  (2) The orchestration bench.py uses on GPUs
(particlesystem_amd.slab: message routes, ring order, the four stage calls) driven with a host
stand-in per rank that holds only its slab (tests/oracle_slab.py): in one process, and as a
world_size-2 and -3 gloo job -- the union of the ranks must be the serial reference state."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_py as O
import particlesystem_amd as ps
from oracle_slab import OracleSlabRank
from particlesystem_amd.slab import HostRing, merge_owned, routes, step_local
from util import cloud, g2_cloud

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def plans_of(world, **over):
    return [ps.slab_plan(ps.default_config(rank=r, world=world, **over)) for r in range(world)]


@pytest.mark.parametrize("over", [{}, {"chunk_factor": 10}, {"chunk_factor": 3, "chunk_dim": 5}, {"chunk_factor": 6, "chunk_dim": 3}])
def test_plans_partition_layers_slots_and_records(over):
    cfg = ps.default_config(**over)
    sizes, table, _, _, _ = ps.describe(cfg)
    G, GG = sizes.grid_dim, sizes.grid_dim ** 2
    seg_base = np.concatenate([[0], np.cumsum(list(sizes.seg_size))])
    info_base = np.concatenate([[0], np.cumsum(list(sizes.seg_count))])
    for world in (1, 2, 3, 4, 5, 8):
        if G < 2 * world:
            with pytest.raises(ps.PsamdError):
                plans_of(world, **over)
            continue
        pl = plans_of(world, **over)
        assert pl[0].cut_lo == 0 and pl[-1].cut_hi == G and pl[0].state_lo == 0 and pl[-1].state_hi == G
        for a, b in zip(pl, pl[1:]):
            assert a.cut_hi == b.cut_lo and a.state_hi == b.state_lo
            # what a sends up is what b holds from below, and the other way round
            assert (a.send_up_lo, a.send_up_hi) == (b.below_lo, b.below_hi)
            assert (b.send_down_lo, b.send_down_hi) == (a.above_lo, a.above_hi)
            assert (a.lentout_lo, a.lentout_hi) == (b.lentin_lo, b.lentin_hi)
            for t in range(4):
                assert a.slot_hi[t] == b.slot_lo[t] and a.rec_hi[t] == b.rec_lo[t]
        for t in range(4):
            assert pl[0].slot_lo[t] == seg_base[t] and pl[-1].slot_hi[t] == seg_base[t + 1]
            assert pl[0].rec_lo[t] == info_base[t] and pl[-1].rec_hi[t] == info_base[t + 1]
        for p in pl:
            assert p.cut_hi - p.cut_lo >= (2 if world > 1 else 1)
            # a rank reads its compute layers and one more on each side: all of them held
            held = set(range(p.state_lo, p.state_hi)) | set(range(p.below_lo, p.below_hi)) | set(range(p.above_lo, p.above_hi))
            assert set(range(max(0, p.cut_lo - 1), min(G, p.cut_hi + 1))) <= held
            # every cell of a state layer has its segment (slots + queue record) on this rank
            for i3 in range(p.state_lo, p.state_hi):
                for c in range(i3 * GG, (i3 + 1) * GG, 7):
                    _, st, tid = table[c]
                    t = {1: 0, 2: 1, 4: 2, 8: 3}[int(st)]
                    rec = info_base[t] + tid
                    assert p.rec_lo[t] <= rec < p.rec_hi[t], (world, p.rank, c)
                    slot0 = seg_base[t] + tid * sizes.seg_size_t[t]
                    assert p.slot_lo[t] <= slot0 < p.slot_hi[t]
        if world > 1:
            assert [p.up_rank for p in pl] == [(r + 1) % world for r in range(world)]


def test_balanced_cuts_and_ring_routes():
    pl = plans_of(8)
    assert [(p.cut_lo, p.cut_hi) for p in pl] == [(2 * r, 2 * r + 2) for r in range(8)]
    pl = plans_of(4)
    assert [(p.cut_lo, p.cut_hi) for p in pl] == [(0, 4), (4, 8), (8, 12), (12, 16)]
    # the outer layers see 18 of 27 cells: a 5-rank cut of 16 layers gives the ends one layer more
    assert [p.cut_hi - p.cut_lo for p in plans_of(5)] in ([4, 3, 3, 3, 3], [3, 3, 3, 3, 4], [4, 3, 3, 3, 3][::-1], [4, 3, 2, 3, 4], [3, 3, 3, 3, 4])
    # every message sent has exactly one receiver slot, and the ring closes
    for world in (2, 3, 8):
        got = {}
        for r in range(world):
            for ph, out_slot, peer, in_slot in routes(r, world):
                assert (ph, peer, in_slot) not in got
                got[(ph, peer, in_slot)] = (r, out_slot)
        # ring neighbours, and from four ranks on the hop-two routes (rank +-2; 0 bytes unless a rank's state is one layer)
        assert sum(1 for k in got if k[0] == "xfer") == (2 if world < 4 else 4) * world


def make_ranks(world, xyz, age, fert, **over):
    ranks = [OracleSlabRank(ps.default_config(rank=r, world=world, **over)) for r in range(world)]
    ref = O.System(**over)
    ids_ref = ref.fill(xyz, age=age, fert_age=fert)
    ids = np.stack([s.fill_particles(xyz, age, fert) for s in ranks])
    assert ((ids >= 0).sum(0) == 1).all() and np.array_equal(ids.max(0), ids_ref)
    return ranks, ref


def union_equals(ranks, ref, what):
    plans = [s.plan for s in ranks]
    p = merge_owned([s.download_particles() for s in ranks], plans)
    for f in p.dtype.names:
        assert np.array_equal(p[f].view(np.uint32 if p[f].dtype.kind == "f" else p[f].dtype),
                              ref.particles[f].view(np.uint32 if p[f].dtype.kind == "f" else p[f].dtype)), (what, f)
    qs = [s.download_queues() for s in ranks]
    assert merge_owned([q[0] for q in qs], plans, "records").tobytes() == ref.queue_info.tobytes(), what
    assert np.array_equal(merge_owned([q[1] for q in qs], plans), ref.queue), what
    for k in ("relocations", "deaths_collision", "survives", "integrated", "relocations_lost"):
        assert sum(s.counters[k] for s in ranks) == ref.counters[k], (what, k)


@pytest.mark.parametrize("world", [2, 3, 4])
def test_stand_in_slabs_reproduce_the_serial_reference(world):
    dt = 0.05
    xyz = g2_cloud()
    fert = (1e6 + np.arange(len(xyz))).astype(np.float32)
    ranks, ref = make_ranks(world, xyz, np.float32(40 * dt), fert, dt=dt)
    for step in range(12):
        step_local(ranks)
        ref.step(1)
        union_equals(ranks, ref, "world %d step %d" % (world, step + 1))
    assert sum(s.sent for s in ranks) > 0 and ref.counters["relocations"] > 0


def state_hash(p, qi, q):
    h = hashlib.sha256()
    for f in p.dtype.names:
        h.update(np.ascontiguousarray(p[f]).tobytes())
    h.update(qi.tobytes()); h.update(np.ascontiguousarray(q).tobytes())
    return h.hexdigest()


N_GLOO, STEPS_GLOO = 20000, 10


def gloo_inputs():
    xyz = cloud(N_GLOO, 17)
    rng = np.random.default_rng(17)
    return xyz, rng.uniform(15 / 7, 7.5, N_GLOO).astype(np.float32), (1e6 + np.arange(N_GLOO)).astype(np.float32)


def allpairs_inputs():
    xyz = g2_cloud()
    n = len(xyz)
    rng = np.random.default_rng(29)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    age[::19] = 0.5                                   # kids: feel and exert nothing
    return xyz, age, (1e6 + np.arange(n)).astype(np.float32)


STEPS_ALLPAIRS = 4


def _worker():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    allp = os.environ.get("PS_ALLPAIRS") == "1"
    xyz, age, fert = allpairs_inputs() if allp else gloo_inputs()
    s = OracleSlabRank(ps.default_config(rank=rank, world=world), all_pairs=allp)
    s.fill_particles(xyz, age, fert)
    ring = HostRing(s, dist, rank, world)
    for _ in range(STEPS_ALLPAIRS if allp else STEPS_GLOO):
        ring.step()
    out = os.environ["PS_OUT"] + ".%d.npz" % rank
    qi, q = s.download_queues()
    np.savez(out, p=s.download_particles(), qi=qi, q=q, sent=s.sent, reloc=s.counters["relocations"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_gloo_job_matches_serial(tmp_path, world):
    """bench.py's multi-GPU loop (slab.HostRing: batched isend/irecv between ring neighbours)
    over gloo, one process per rank, 10 steps, particles changing owner every step.  World 2: both
    neighbours are the same peer (two messages each way in one batch, matched by order); world 3:
    a middle rank with two different peers, and the periodic wrap between the first and last rank."""
    xyz, age, fert = gloo_inputs()
    ref = O.System()
    ref.fill(xyz, age=age, fert_age=fert)
    ref.step(STEPS_GLOO)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29531 + world), WORLD_SIZE=str(world), PS_OUT=str(tmp_path / "rank"),
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=560)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    got = [np.load(str(tmp_path / "rank") + ".%d.npz" % r) for r in range(world)]
    plans = plans_of(world)
    p = merge_owned([g["p"] for g in got], plans)
    qi = merge_owned([g["qi"] for g in got], plans, "records")
    q = merge_owned([g["q"] for g in got], plans)
    assert state_hash(p, qi, q) == state_hash(ref.particles, ref.queue_info, ref.queue)
    assert sum(int(g["sent"]) for g in got) > 0 and sum(int(g["reloc"]) for g in got) == ref.counters["relocations"] > 0


@pytest.mark.timeout(600)
def test_gloo_job_all_pairs_snapshot_gather(tmp_path):
    """All-pairs forces across ranks: every rank contributes the snapshot of its
    own cells to an ALL-GATHER between slab_build and slab_pairs (slab.HostRing.gather_snapshot, the
    same call sequence DeviceRing makes with RCCL's ncclAllGather) and adds the far field of all
    ranks' cells to its particles.  A world_size-2 gloo job of stand-in ranks must end in the state a
    world of ONE stand-in reaches -- which sees the same bodies in the same global cell order."""
    xyz, age, fert = allpairs_inputs()
    one = OracleSlabRank(ps.default_config(rank=0, world=1), all_pairs=True)
    one.fill_particles(xyz, age, fert)
    cut = OracleSlabRank(ps.default_config(rank=0, world=1))
    cut.fill_particles(xyz, age, fert)
    for _ in range(STEPS_ALLPAIRS):
        step_local([one]); step_local([cut])
    assert one.download_particles().tobytes() != cut.download_particles().tobytes(), "the far field must matter"
    world = 2
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE=str(world), PS_OUT=str(tmp_path / "rank"), PS_ALLPAIRS="1",
               PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]))
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker"], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=560)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0, out
    got = [np.load(str(tmp_path / "rank") + ".%d.npz" % r) for r in range(world)]
    plans = plans_of(world)
    p = merge_owned([g["p"] for g in got], plans)
    qi = merge_owned([g["qi"] for g in got], plans, "records")
    q = merge_owned([g["q"] for g in got], plans)
    qi1, q1 = one.download_queues()
    assert state_hash(p, qi, q) == state_hash(one.download_particles(), qi1, q1)


if __name__ == "__main__" and "--worker" in sys.argv:
    _worker()


@pytest.mark.timeout(300)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts the ranks itself
    (fresh processes with RANK / WORLD_SIZE / MASTER_* set), relays rank 0's single JSON line and exits 0."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--steps", "7", "--warmup", "3"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["launch_check"] and d["steps"] == 7 and d["warmup"] == 3
    assert d["host"].startswith("C++ ranks")       # the default: every Python rank started its C++ program (host/ps_ring_rccl)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("backend", ["ring", "gloo"])
def test_bench_under_the_drivers_launcher(backend):
    """As the driver starts it for N > 1: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N.  With the
    default backend each rank starts its C++ program, the programs meet through the id file (nonce from MASTER_PORT and
    the launcher's pid) and rank 0's record comes back as the one line; --backend gloo: the torch.distributed ranks."""
    import json
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "3", "--launch-check", "--steps", "5", "--warmup", "2",
                        "--backend", backend], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["launch_check"] and d["steps"] == 5 and d["warmup"] == 2
    assert ("host" in d) == (backend == "ring")


@pytest.mark.timeout(120)
def test_bench_launcher_ends_the_job_when_a_rank_dies():
    """rank 1 exits before the rendezvous: rank 0 would wait for it for ever; the launcher notices,
    ends rank 0 and exits with the dead rank's status."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PSAMD_BENCH_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=100)
    assert p.returncode == 3, (p.returncode, p.stderr[-1000:])
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
