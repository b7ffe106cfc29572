"""host/ps_driver: the reference's driver loop (DoParallelProcess, ps.cpp:1843-1928) in C++
on the C ABI.  CPU: it builds with plain g++ against include/psamd.h and its host-only mode
agrees with the library's geometry.  GPU: ten iterations of the golden cloud end in the
oracle's state, with and without the reference's per-stage fetch-back."""
import os
import re
import subprocess

import numpy as np
import pytest

import particlesystem_amd as ps
from particlesystem_amd import _build as psbuild
from util import GOLDEN, O, g2_cloud

CLOUD = os.path.join(GOLDEN, "g2_cloud_n4096_seed12345.f32")


def digest(particles):
    """state_digest of host/ps_driver.cpp on a P_DATA_TYPE array."""
    w = np.ascontiguousarray(particles).view(np.uint32).reshape(-1, 18).copy()
    w[:, 5] &= 0x0000FFFF
    flat = w.reshape(-1).astype(np.uint64)
    k = np.arange(flat.size, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return int((flat * (k % np.uint64(65521) + np.uint64(1))).sum(dtype=np.uint64))


def run_driver(*args):
    exe = psbuild.build_driver()
    out = subprocess.run([exe, *args], check=True, capture_output=True, text=True, timeout=600).stdout
    return out


def test_driver_builds_and_describes_the_default_geometry():
    out = run_driver("--describe")
    s = ps.describe(ps.default_config())[0]
    m = re.search(r"grid (\d+)\^3 cells, (\d+) chunks, container (\d+) slots, (\d+) queue records, cell list (\d+), chunk list (\d+)", out)
    assert m, out
    assert [int(v) for v in m.groups()] == [s.grid_dim, s.num_chunks, s.container_size, s.queue_info_size,
                                            s.max_per_cell, s.max_per_chunk]


@pytest.mark.gpu
@pytest.mark.parametrize("fetch_back", [False, True])
def test_driver_ends_in_the_oracle_state(fetch_back):
    xyz = g2_cloud()
    dt = 0.01
    o = O.System(dt=dt)
    o.fill(xyz, age=np.float32(40 * dt), fert_age=(1e6 + np.arange(len(xyz))).astype(np.float32))
    o.step(10)
    args = ["--cloud", CLOUD, "--iters", "10", "--dt", str(dt)] + (["--fetch-back"] if fetch_back else [])
    out = run_driver(*args)
    m = re.search(r"state-hash ([0-9a-f]{16}) live (\d+)", out)
    assert m, out
    assert int(m.group(2)) == int((o.particles["cell"] >= 0).sum())
    assert int(m.group(1), 16) == digest(o.particles), out
    assert out.count(">>>>>>>>>>> Execution time of iteration (sec):") == 10    # the reference's per-iteration print


G1 = ["--graphs", "1"]       # every stage's kernels as one hipGraph (off by default: profiles/r4_ab_graphs.txt)
RING_CASES = [
    (2, []), (3, ["--overlap-interior", "--side-stream", "1"] + G1), (4, ["--births", "--side-stream", "2"] + G1), (8, G1),
    (2, ["--all-pairs"] + G1), (4, ["--all-pairs"]), (8, ["--all-pairs", "--n", "30000"]),
    (3, ["--side-stream", "2"]),                               # the round-4 form: every transfer on a second stream, events in between
    (4, ["--births", "--overlap-interior", "--wait", "0"]), (8, ["--births", "--side-stream", "1"]),
]


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,extra", RING_CASES, ids=["w%d%s" % (w, "".join(a for a in e if not a.isdigit())) for w, e in RING_CASES])
def test_cpp_ring_moves_every_message_with_rccl(world, extra):
    """host/ps_ring_rccl --loopback: all slabs in one C++ process on this GPU, a communicator of one
    rank, every halo / force / transfer message an ncclSend to self matched by an ncclRecv from self
    -- on the compute stream between the stage kernels by default, on a second HIP stream ordered by events with
    --side-stream 1 / 2 (where every stage may run as one captured hipGraph) --, the status records -- and, with --all-pairs, the snapshot blocks: SURVEY 8(e)'s
    all-gather of positions once per step -- by ncclAllGather.  The program itself requires the
    union of the slabs to equal the single-context run byte for byte (which the parity tests tie to
    the oracle) and exits non-zero otherwise.  Stage loop: ps.cpp:1843-1928; what a rank subscribes
    to: ps.cpp:380-487."""
    exe = psbuild.build_ring()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "--loopback", "--world", str(world), "--n", "60000", "--iters", "8"] + extra, env=env,
                       capture_output=True, text=True, timeout=560)
    print(p.stdout)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    m = re.search(r"ring-rccl ok: 0 of \d+ records differ from the single context after 8 steps \((\d+) relocations, (\d+) births", p.stdout)
    assert m and int(m.group(1)) > 0, p.stdout
    assert (int(m.group(2)) > 0) == ("--births" in extra), p.stdout
    assert float(re.search(r"([0-9.]+) MB through RCCL", p.stdout).group(1)) > 1.0
    replays = int(re.search(r"(\d+) graph replays", p.stdout).group(1))
    if "--graphs" not in extra:
        assert replays == 0
    else:
        assert replays >= world * 8 * 3, p.stdout        # four stage sequences per rank and step, less the captures


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_cpp_ring_bench_record_in_loopback():
    """The benchmark protocol of the C++ host (what bench.py --gpus N relays from rank 0), all four slabs in one
    process: settle, warm up, time K restored steps between barriers, census of the frame -- one JSON record."""
    import json
    exe = psbuild.build_ring()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "--loopback", "--world", "4", "--bench", "--n", "131072", "--steps", "6", "--warmup", "2",
                        "--settle-seconds", "0.05", "--timing-period", "2", "--graphs", "1", "--side-stream", "2"], env=env, capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    recs = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{") and '"psamd_ring"' in l]
    assert len(recs) == 1, p.stdout
    r = recs[0]
    assert r["world"] == 4 and r["steps"] == 6 and r["updates"] == 6 * 131072 and r["elapsed_s"] > 0
    assert r["graphs"] is True and r["graph_replays"] > 0 and r["side_stream"] is True
    assert r["pairs_rank0"] > 0 and r["kernel_us"]["pairs"] > 0 and r["timed_launches"] == 3
    assert r["message_bytes_rank0"]["halo_up"] > 0 and r["particles_with_a_force_term"] > 0
