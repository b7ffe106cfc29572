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
