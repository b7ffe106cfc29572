"""Host-side cost of what a rank does per step besides computing: a batch of RCCL sends and
receives posted from Python (torch.distributed.batch_isend_irecv), the status all-gather, and the
four stage calls' enqueue time.  One GPU: a world of one rank sending to itself.  Prints microseconds."""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import particlesystem_amd as ps  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29655")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
st = torch.cuda.Stream()
a = [torch.zeros(1 << 20, dtype=torch.int32, device="cuda") for _ in range(4)]
ops = [dist.P2POp(dist.isend, a[0], 0), dist.P2POp(dist.isend, a[1], 0), dist.P2POp(dist.irecv, a[2], 0), dist.P2POp(dist.irecv, a[3], 0)]
so, si = torch.zeros(4096, dtype=torch.int32, device="cuda"), torch.zeros(4096, dtype=torch.int32, device="cuda")


def timed(f, n=200):
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    host = (time.perf_counter() - t) / n
    torch.cuda.synchronize()
    return 1e6 * host


def batch():
    with torch.cuda.stream(st):
        for w in dist.batch_isend_irecv(ops):
            w.wait()


def gather():
    with torch.cuda.stream(st):
        dist.all_gather_into_tensor(si, so, async_op=True).wait()


print("batch_isend_irecv (2 sends + 2 receives of 4 MB, posted and stream-waited): %.1f us of host time" % timed(batch))
print("all_gather_into_tensor (16 KB, async + stream wait): %.1f us of host time" % timed(gather))

g = ps.ParticleSystem(ps.default_config(rank=0, world=1))
xyz = g.uniform_cloud(1 << 17, 3)
g.fill_particles(xyz, age=np.float32(3.0), fert_age=np.float32(1e6))
g.snapshot_save()
g.set_stream(st.cuda_stream)


def stages():
    g.snapshot_restore()
    g.slab_build(); g.slab_pairs(); g.slab_apply(); g.slab_finish()


print("snapshot_restore + the four stage calls, N = 2^17 (host time incl. the step's one read-back): %.1f us" % timed(stages, 100))
dist.destroy_process_group()
