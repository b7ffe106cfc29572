"""`python bench.py --gpus 2` end to end on the GPU box: bench.py starts its own two ranks
(torch.distributed.run), each holds one slab of a small cloud on the (shared) GPU, the messages
travel through the process group -- gloo here, because RCCL refuses two ranks on one device;
with the default backend the same code path hands the device buffers to RCCL -- and rank 0
prints the one JSON line the driver reads."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_two_ranks_one_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--n", "131072",
                        "--steps", "4", "--warmup", "1", "--settle-seconds", "0.05"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "strong"
    assert d["value"] > 0 and d["config"]["updates_in_timed_region"] == 4 * 131072
    assert "slabs" in d["config"]["parallelism"] and d["cpu_baseline"] is None
    assert d["roofline"]["frac"] > 0 and d["config"]["message_bytes_rank0"]["halo_up"] > 0


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("ranks,all_pairs", [(4, False), (2, True)])
def test_bench_more_ranks_and_all_pairs(ranks, all_pairs):
    """Four slabs (interior ranks with two neighbours each, the status all-gather over four records), and the
    all-pairs mode across two ranks (the snapshot all-gather through the process group): the line comes out,
    every particle was advanced in every timed step."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    n = 32768 if all_pairs else 131072
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--backend", "gloo", "--n", str(n),
           "--steps", "3", "--warmup", "1", "--settle-seconds", "0.05"] + (["--all-pairs"] if all_pairs else [])
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=860)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["steps"] == 3
    assert d["value"] > 0 and d["config"]["updates_in_timed_region"] == 3 * n
    assert ("all-pairs" in d["metric"]) == all_pairs


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_line_from_the_cpp_ranks_record():
    """What rank 0's C++ program prints (here: four slabs in one process, --loopback) becomes the driver's line."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    import particlesystem_amd as ps
    exe = ps._build.build_ring()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "--loopback", "--world", "4", "--bench", "--n", "131072", "--steps", "4", "--warmup", "1", "--settle-seconds", "0.05",
                        "--timing-period", "2", "--graphs", "1"], env=env, capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{") and '"psamd_ring"' in l][-1])
    args = argparse.Namespace(all_pairs=False, fast_math=False, evolve=False, n=131072)
    d = bench.line_from_ring_record(args, rec)
    assert d["n_gpus"] == 4 and d["steps"] == 4 and d["scaling"] == "strong" and d["higher_is_better"] is True
    assert d["value"] == pytest.approx(4 * 131072 / rec["elapsed_s"]) and d["ms_per_step"] == pytest.approx(1e3 * rec["elapsed_s"] / 4)
    assert d["config"]["updates_in_timed_region"] == 4 * 131072 and "C++ ranks" in d["config"]["host"] and "hipGraph" in d["config"]["host"]
    assert 0 < d["roofline"]["frac"] < 1 and d["roofline"]["bound"] == "valu" and d["roofline_streaming"]["bound"] == "hbm"
    json.dumps(d)


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_ranks_fall_back_together_when_the_cpp_ranks_fail():
    """Every C++ rank fails at once (a test hook breaks their command line): the Python ranks hear of it through the files
    beside the id file and run the step over torch.distributed instead (gloo here: two ranks share this GPU); the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PSAMD_BENCH_RING_BREAK="1", PSAMD_BENCH_FALLBACK_BACKEND="gloo")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "131072", "--steps", "3", "--warmup", "1",
                        "--settle-seconds", "0.05"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["updates_in_timed_region"] == 3 * 131072
    assert "FALLBACK" in d["config"]["host"] and "falling back" in p.stderr


def _bench(args, env_extra=None, timeout=560):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_one_gpu_through_the_cpp_host_agrees_with_the_python_host():
    """The whole scaling curve is driven by ONE host: `bench.py --gpus 1` goes through host/ps_ring_rccl --world 1 by
    default.  Its figure must be the Python host's (psamd_step through ctypes) within 2 % -- same kernels, and neither
    host is on the step's critical path; measured 0.6-0.9 % apart on one box, profiles/r5_ab_host.txt, of which 0.35 %
    are the C++ host's own stage events on every eighth step -- and its record must be the full one-GPU line."""
    common = ["--gpus", "1", "--steps", "150", "--warmup", "5", "--no-side-runs", "--no-cpu"]
    ring, _ = _bench(common)                                   # (--host ring is the default)
    py, _ = _bench(common + ["--host", "python"])
    assert "C++ ranks" in ring["config"]["host"] and "ring_failed" not in ring and "Python" in py["config"]["host"]
    assert ring["n_gpus"] == py["n_gpus"] == 1 and ring["steps"] == py["steps"] == 150
    assert ring["config"]["updates_in_timed_region"] == py["config"]["updates_in_timed_region"] == 150 * (1 << 20)
    assert ring["config"]["particles_with_a_force_term"] == py["config"]["particles_with_a_force_term"]
    print("one GPU: C++ host %.4f ms per step, Python host %.4f" % (ring["ms_per_step"], py["ms_per_step"]))
    assert abs(ring["ms_per_step"] / py["ms_per_step"] - 1.0) < 0.02
    assert ring["rccl_ranks"] == 1 and ring["roofline"]["traffic"] == py["roofline"]["traffic"]
    for d in (ring, py):
        assert set(d["kernel_us_per_step"]) >= {"pairs", "apply", "lifecycle", "collide"} and d["kernel_us_per_step_max"]["pairs"] >= d["kernel_us_per_step"]["pairs"]
        assert 0.15 < d["roofline"]["frac"] < 0.5


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_the_drivers_form_of_the_bench_runs_through_the_cpp_host():
    """`bench.py --steps 20 --warmup 5` -- what the driver runs at round end -- must be served by the C++ host and not by
    the fallback: a timed region whose length is no multiple of the timing period leaves stage events unrecorded, and
    asking those for their time once left a HIP error behind that failed the next launch (the ranks fell back, quietly)."""
    for steps in ("20", "50"):
        d, _ = _bench(["--gpus", "1", "--steps", steps, "--warmup", "5", "--no-side-runs", "--no-cpu"])
        assert "C++ ranks" in d["config"]["host"] and "FALLBACK" not in d["config"]["host"] and "ring_failed" not in d, d["config"]["host"]
        assert d["steps"] == int(steps) and d["rccl_ranks"] == 1
        assert d["sustained"] and abs(d["sustained"]["ms_per_step"] / d["ms_per_step"] - 1.0) < 0.05


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_cpp_host_record_explains_itself_on_eight_slabs():
    """What a multi-GPU run must say about itself (here eight slabs in one process, --loopback): how many ranks RCCL
    joined, every rank's own stage times, how long the compute stream waited for each phase's messages (minimum and
    maximum over the ranks), the bytes per phase -- and the message sizes were checked before the first step."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    import particlesystem_amd as ps
    exe = ps._build.build_ring()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "--loopback", "--world", "8", "--bench", "--n", "262144", "--steps", "6", "--warmup", "1", "--settle-seconds", "0.05",
                        "--timing-period", "2"], env=env, capture_output=True, text=True, timeout=560)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith("{") and '"psamd_ring"' in l][-1])
    assert rec["rccl_ranks"] == 1 and rec["world"] == 8                    # (a loopback communicator has one rank; --gpus 8 must say 8 here)
    assert set(rec["stage_ms_per_rank"]) == {"build", "pairs", "apply", "finish"}
    assert all(len(v) == 8 and min(v) > 0 for v in rec["stage_ms_per_rank"].values())
    assert set(rec["wait_ms"]) == {"halo", "force", "xfer"} and all(w["max"] >= w["min"] >= 0 for w in rec["wait_ms"].values())
    assert rec["bytes_per_phase_rank0"]["halo"] > 0 and rec["bytes_per_phase_rank0"]["xfer"] > 0 and rec["bytes_per_phase_rank0"]["gathers"] > 0
    assert rec["kernel_us_max"]["pairs"] >= rec["kernel_us_median"]["pairs"] > 0
    d = bench.line_from_ring_record(argparse.Namespace(all_pairs=False, fast_math=False, evolve=False, n=262144), rec)
    assert d["rccl_ranks"] == 1 and d["stage_ms_per_rank"] == rec["stage_ms_per_rank"] and d["wait_for_messages_ms"] == rec["wait_ms"]
    json.dumps(d)


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_two_cpp_ranks_on_two_gpus_over_rccl():
    """The path `bench.py --gpus N` takes on a node: one C++ rank per GPU, the id-file rendezvous, ncclCommInitRank
    across processes, the non-loopback exchange.  Needs two devices (the one-GPU boxes of this pool skip it)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here: the multi-process RCCL path needs two")
    d, err = _bench(["--gpus", "2", "--n", "262144", "--steps", "8", "--warmup", "2", "--settle-seconds", "0.1"])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and "C++ ranks" in d["config"]["host"] and "ring_failed" not in d
    assert d["config"]["updates_in_timed_region"] == 8 * 262144
    assert all(len(v) == 2 for v in d["stage_ms_per_rank"].values())


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_cpp_ranks_that_disagree_on_message_sizes_do_not_start():
    """A rank created with another halo_cap_cell than its neighbours would post sends and receives of other sizes than
    they expect, and RCCL would sit in the transfer until the watchdog: the size tables of all ranks are compared before
    the first step (in loopback directly; across processes they are all-gathered) and nobody starts."""
    import particlesystem_amd as ps
    exe = ps._build.build_ring()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([exe, "--loopback", "--world", "3", "--n", "20000", "--iters", "2", "--halo-cap-cell", "64", "--test-size-mismatch"],
                       env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode != 0 and "message sizes disagree" in p.stderr, (p.returncode, p.stderr[-1000:])
    p = subprocess.run([exe, "--loopback", "--world", "3", "--n", "20000", "--iters", "2", "--halo-cap-cell", "64"], env=env, capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stderr[-1000:]
