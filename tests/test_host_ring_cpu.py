"""host/ps_ring_rccl.cpp, the part of its multi-process path that runs without a GPU: the rendezvous
of the communicator id through a file.  (The non-loopback path -- one process per GPU -- is the only
multi-process code of the product that has never run on hardware here: this pool has one GPU per
box.  Its message routes are the ones --loopback exercises on the GPU, tests/test_host_driver.py.)"""
import os
import struct
import subprocess
import time

import particlesystem_amd as ps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_id_file_rendezvous_ignores_a_stale_file(tmp_path):
    """A file left behind by an earlier job (another nonce) must not be taken for this job's id: ranks
    that start before rank 0 wait past it; all ranks end up with the id rank 0 wrote."""
    exe = ps._build.build_ring()
    idf = str(tmp_path / "id")
    with open(idf, "wb") as f:                       # a stale record: right magic, another job
        f.write(struct.pack("<QQ", 0x70735f72696e6731, 41) + b"\x55" * 128)
    args = [exe, "--world", "3", "--id-file", idf, "--job", "42", "--id-only"]
    late = [subprocess.Popen(args + ["--rank", str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in (1, 2)]
    time.sleep(1.0)
    assert all(p.poll() is None for p in late), "a rank accepted the stale id file"
    first = subprocess.run(args + ["--rank", "0"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    outs = [first.stdout] + [p.communicate(timeout=60)[0] for p in late]
    assert first.returncode == 0 and all(p.returncode == 0 for p in late), outs
    ids = {o.split("communicator id ")[1].split()[0] for o in outs}
    assert len(ids) == 1, outs


def test_usage_errors_are_reported():
    exe = ps._build.build_ring()
    p = subprocess.run([exe, "--world", "2", "--rank", "1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=30)
    assert p.returncode == 2 and "usage" in p.stdout      # more than one rank needs --id-file (or --loopback)
