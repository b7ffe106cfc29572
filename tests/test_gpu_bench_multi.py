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
