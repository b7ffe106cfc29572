"""RCCL on the real message buffers, with the one GPU a test box has: a world-1 process group
(backend nccl = RCCL) whose only rank sends to ITSELF.  Three slab contexts (ranks 0..2 of a
three-rank plan) live in this process; every message of a step is an RCCL send from the sending context's device
buffer and an RCCL receive into the receiving context's device buffer, posted in one
batch_isend_irecv per phase exactly as DeviceRing posts them (same tensors over the same device
pointers, same stream discipline: posted under the contexts' stream, waited for stream-side).
The status records go through all_gather_into_tensor.  What this cannot show is two RANKS
agreeing on the order of their operations; tests/test_slab_cpu.py does that over gloo.
Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
for d in (HERE, os.path.join(HERE, ".."), os.path.join(HERE, "..", "oracle")):
    sys.path.insert(0, os.path.abspath(d))
import oracle_py as O                      # noqa: E402  (the checker)
import particlesystem_amd as ps            # noqa: E402
from particlesystem_amd.slab import DeviceRing, STATUS_IN, STATUS_OUT, merge_owned, routes   # noqa: E402
from util import cloud, oracle_cfg_from    # noqa: E402

WORLD = 3


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    stream = torch.cuda.Stream()
    n = 60000
    xyz = cloud(n, 77)
    rng = np.random.default_rng(77)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    fert = (1e6 + np.arange(n)).astype(np.float32)
    ranks = [ps.ParticleSystem(ps.default_config(rank=r, world=WORLD)) for r in range(WORLD)]
    for g in ranks:
        g.fill_particles(xyz, age=age, fert_age=fert)
    o = O.System(oracle_cfg_from(ranks[0].cfg))
    o.fill(xyz, age=age, fert_age=fert)
    rings = [DeviceRing(g, dist, r, WORLD, stream) for r, g in enumerate(ranks)]

    def exchange(phase):
        sends, recvs = [], []
        for r in range(WORLD):
            for ph, out_slot, peer, in_slot in routes(r, WORLD):
                if ph == phase and ranks[r].msg_bytes(out_slot):
                    sends.append(dist.P2POp(dist.isend, rings[r].t[out_slot], 0))
                    recvs.append(dist.P2POp(dist.irecv, rings[peer].t[in_slot], 0))   # k-th receive from self = k-th send to self
        with torch.cuda.stream(stream):
            for w in dist.batch_isend_irecv(sends + recvs):
                w.wait()
        return len(sends)

    sent = 0
    moved = 0
    for step in range(8):
        with torch.cuda.stream(stream):
            for g in ranks:
                g.slab_build()
            sent += exchange("halo")
            # status: every context's record into every context's gathered block (world-1 all-gather = RCCL copy)
            for r in range(WORLD):
                for q in range(WORLD):
                    nb = rings[r].t[STATUS_OUT].numel()
                    dist.all_gather_into_tensor(rings[q].t[STATUS_IN][r * nb:(r + 1) * nb], rings[r].t[STATUS_OUT])
            for g in ranks:
                g.slab_pairs()
            sent += exchange("force")
            for g in ranks:
                g.slab_apply()
            sent += exchange("xfer")
            for g in ranks:
                g.slab_finish()
        o.step(1)
        moved += sum(int(g.msg_download(ps.MSG_XFER_OUT + k)[0]) for g in ranks for k in (0, 1))
        plans = [g.slab_plan() for g in ranks]
        p = merge_owned([g.download_particles() for g in ranks], plans)
        if p.tobytes() != o.particles.tobytes():
            print(json.dumps({"ok": False, "step": step + 1}))
            return 1
    dist.destroy_process_group()
    print(json.dumps({"ok": True, "steps": 8, "rccl_messages": sent, "changed_owner": moved,
                      "relocations": int(o.counters["relocations"])}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
