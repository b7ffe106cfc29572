"""The slab messages through RCCL itself on the one GPU of a test box (see rccl_loopback_worker.py:
a world-1 nccl group whose rank sends to itself, three slab contexts in the process)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_three_slabs_exchange_their_messages_through_rccl():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(HERE, "rccl_loopback_worker.py")], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=560)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-3000:])
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(line)
    assert line["ok"] and line["rccl_messages"] >= 8 * 8 and line["changed_owner"] > 0
