"""Stepping a slab-partitioned system: the message exchange around the four stage calls.

Every rank holds one slab of the system (include/psamd.h, "slab partition": the segments of
its cell layers, their slots, particles and free-slot queues) and talks to its two neighbours
only.  One step is

    slab_build   ->  halo:  snapshot of boundary layers to the rank below / above
    slab_pairs   ->  force: (ax, ay, az, flag) of lent layers back to the rank below
    slab_apply   ->  xfer:  particles whose new segment a neighbour owns (ring: periodic box)
    slab_finish

with fixed-size messages, so the transport never negotiates a length.  This module is the
host-side plumbing only; it computes nothing.  Three transports:

  * ``DeviceRing``  torch.distributed P2P (RCCL over xGMI) directly on the contexts' device
    buffers; the transfers run on RCCL's stream: the status records land before the first
    pair-stage call, the halo travels beside the interior pass if that is asked for (overlap_interior).
  * ``HostRing``    torch.distributed P2P (gloo) through host copies of the messages: for tests,
    and for several processes that share one GPU.
  * ``step_local``  all ranks live in one process (tests): messages are copied rank to rank.

A "rank" is anything with the stage calls and ``msg_bytes / msg_download / msg_upload``
(``particlesystem_amd.ParticleSystem``; the CPU tests drive a host stand-in the same way).
"""
import numpy as np

# message slots, as in psamd_slab_msg_download: out/in x below/above; the status record is all-gathered
HALO_OUT, HALO_IN, FORCE_OUT, FORCE_IN, XFER_OUT, XFER_IN, STATUS_OUT, STATUS_IN = 0, 2, 4, 5, 6, 8, 10, 11
ALLG_OUT, ALLG_IN = 12, 13          # all-pairs forces only: every rank's snapshot block, all-gathered between build and pairs
XFER2_OUT, XFER2_IN = 14, 16        # particles changing owner to a rank TWO away (worlds where a single-layer rank can be flown over)
FAR_OUT, FAR_IN = 18, 19            # ... to any other rank: a small outbox, all-gathered in the transfer phase (births on, world >= 4)
BELOW, ABOVE = 0, 1


def routes(rank, world, periodic_ring=True):
    """(phase, out slot of this rank, peer rank, in slot of the peer) for every message this
    rank may send.  Which of them exist is decided by the sizes (0 bytes = no such message)."""
    r = []
    if rank > 0:
        r.append(("halo", HALO_OUT + BELOW, rank - 1, HALO_IN + ABOVE))
        r.append(("force", FORCE_OUT, rank - 1, FORCE_IN))
    if rank + 1 < world:
        r.append(("halo", HALO_OUT + ABOVE, rank + 1, HALO_IN + BELOW))
    if world > 1:
        r.append(("xfer", XFER_OUT + BELOW, (rank - 1) % world, XFER_IN + ABOVE))
        r.append(("xfer", XFER_OUT + ABOVE, (rank + 1) % world, XFER_IN + BELOW))
    if world >= 4:      # (0 bytes unless the plan has a rank whose whole state is one layer)
        r.append(("xfer", XFER2_OUT + BELOW, (rank - 2) % world, XFER2_IN + ABOVE))
        r.append(("xfer", XFER2_OUT + ABOVE, (rank + 2) % world, XFER2_IN + BELOW))
    return r


def _bytes(sysr, slot):
    """size of a message slot; ranks that predate a slot (stand-ins) simply do not have it"""
    try:
        return sysr.msg_bytes(slot)
    except (KeyError, IndexError):
        return 0


def step_local(ranks, overlap_interior=False):
    """One step of a whole system whose ranks all live in this process (tests)."""
    world = len(ranks)

    def deliver(phase):
        for r, sysr in enumerate(ranks):
            for ph, out_slot, peer, in_slot in routes(r, world):
                n = _bytes(sysr, out_slot)
                if ph != phase or n == 0:
                    continue
                assert ranks[peer].msg_bytes(in_slot) == n, (phase, r, peer, n, ranks[peer].msg_bytes(in_slot))
                ranks[peer].msg_upload(in_slot, sysr.msg_download(out_slot))

    def gather(out_slot, in_slot):                                                # the "all-gathers"
        if world > 1 and _bytes(ranks[0], out_slot):
            every = np.concatenate([s.msg_download(out_slot) for s in ranks])
            for s in ranks:
                s.msg_upload(in_slot, every)

    for s in ranks:
        s.slab_build()
    gather(STATUS_OUT, STATUS_IN)             # before the first pair-stage call: its chunk census needs every rank's record
    if overlap_interior:
        for s in ranks:
            s.slab_pairs_interior()   # (in a real run: while the halo travels)
    deliver("halo")
    gather(ALLG_OUT, ALLG_IN)
    for s in ranks:
        s.slab_pairs()
    deliver("force")
    for s in ranks:
        s.slab_apply()
    deliver("xfer")
    gather(FAR_OUT, FAR_IN)
    for s in ranks:
        s.slab_finish()


class _Ring:
    """Pairs every message this rank sends with the one it receives in the same phase."""

    def __init__(self, sysr, dist, rank, world):
        self.s, self.dist, self.rank, self.world = sysr, dist, rank, world
        self.sends = {}      # phase -> [(out slot, peer)]
        self.recvs = {}      # phase -> [(in slot, peer)]
        for ph, out_slot, peer, _ in routes(rank, world):
            if _bytes(sysr, out_slot):
                self.sends.setdefault(ph, []).append((out_slot, peer))
        # what the neighbours (and, hop two, their neighbours) send here: their routes, seen from this side
        for peer in sorted({(rank - 1) % world, (rank + 1) % world, (rank - 2) % world, (rank + 2) % world} - {rank}):
            for ph, _, dst, in_slot in routes(peer, world):
                if dst == rank and _bytes(sysr, in_slot):
                    self.recvs.setdefault(ph, []).append((in_slot, peer))
        # both sides must post the operations between one pair of ranks in the same order
        # (RCCL matches sends and receives of a pair by order, there are no tags): order every
        # list by (peer, direction of travel)
        for d in (self.sends, self.recvs):
            for ph in d:
                d[ph].sort(key=lambda e: (e[1], e[0] & 1) if d is self.sends else (e[1], 1 - (e[0] & 1)))

    def _ops(self, phase, tensor_of):
        P2POp = self.dist.P2POp
        ops = []
        for slot, peer in self.sends.get(phase, []):
            ops.append(P2POp(self.dist.isend, tensor_of(slot), peer))
        for slot, peer in self.recvs.get(phase, []):
            ops.append(P2POp(self.dist.irecv, tensor_of(slot), peer))
        return ops


class HostRing(_Ring):
    """Messages travel as CPU tensors (gloo, or any backend that takes host memory)."""

    def exchange(self, phase):
        import torch
        out = {slot: torch.from_numpy(self.s.msg_download(slot)) for slot, _ in self.sends.get(phase, [])}
        inn = {slot: torch.empty(self.s.msg_bytes(slot) // 4, dtype=torch.int32) for slot, _ in self.recvs.get(phase, [])}
        ops = self._ops(phase, lambda slot: out[slot] if slot in out else inn[slot])
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for slot, t in inn.items():
            self.s.msg_upload(slot, t.numpy())

    def finish_status(self, work):
        pass                                    # (gather_status is synchronous here)

    def _gather(self, out_slot, in_slot):
        import torch
        n = _bytes(self.s, out_slot)
        if not n or self.world == 1:
            return
        mine = torch.from_numpy(self.s.msg_download(out_slot))
        every = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(every, mine)
        self.s.msg_upload(in_slot, torch.cat(every).numpy())

    def gather_status(self):
        self._gather(STATUS_OUT, STATUS_IN)

    def gather_snapshot(self):
        """all-pairs forces: every rank's snapshot block to every rank (synchronous here)"""
        self._gather(ALLG_OUT, ALLG_IN)

    def finish_snapshot(self, work):
        pass

    def gather_far(self):
        """the far outboxes (records for ranks beyond the neighbour messages' reach), with the transfer phase"""
        self._gather(FAR_OUT, FAR_IN)

    def finish_far(self, work):
        pass

    def step(self):
        s = self.s
        s.slab_build()
        self.exchange("halo")
        self.gather_snapshot()
        self.gather_status()
        s.slab_pairs()
        self.exchange("force")
        s.slab_apply()
        self.exchange("xfer")
        self.gather_far()
        s.slab_finish()


class _DevPtr:
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<i4", "data": (int(ptr), False), "version": 2}


class DeviceRing(_Ring):
    """RCCL send/recv straight out of / into the contexts' message buffers (no copies).
    Stage kernels and collectives are ordered by ONE stream: torch's current stream, which the
    context is switched to (psamd_set_stream)."""

    def __init__(self, sysr, dist, rank, world, torch_stream, overlap_interior=False):
        """overlap_interior: run the pair stage of the cells that need no halo as a pass of its own
        while the halo travels.  Off by default: two passes are each less well filled than one --
        measured on one MI355X (N = 2^20, half the cloud per rank) 1.68 ms for the two against
        1.40 ms for the single pass, far more than the ~50 us of halo transfer they would hide."""
        super().__init__(sysr, dist, rank, world)
        self.overlap_interior = overlap_interior
        import torch
        b = sysr.slab_buffers()
        ptrs = {HALO_OUT + 0: (b.halo_out[0], b.halo_out_bytes[0]), HALO_OUT + 1: (b.halo_out[1], b.halo_out_bytes[1]),
                HALO_IN + 0: (b.halo_in[0], b.halo_in_bytes[0]), HALO_IN + 1: (b.halo_in[1], b.halo_in_bytes[1]),
                FORCE_OUT: (b.force_out, b.force_out_bytes), FORCE_IN: (b.force_in, b.force_in_bytes),
                XFER_OUT + 0: (b.xfer_out[0], b.xfer_bytes_max), XFER_OUT + 1: (b.xfer_out[1], b.xfer_bytes_max),
                XFER_IN + 0: (b.xfer_in[0], b.xfer_bytes_max), XFER_IN + 1: (b.xfer_in[1], b.xfer_bytes_max),
                STATUS_OUT: (b.status_out, b.status_bytes), STATUS_IN: (b.status_in, b.status_bytes * world),
                ALLG_OUT: (b.allg_out, b.allg_bytes), ALLG_IN: (b.allg_in, b.allg_bytes * world),
                XFER2_OUT + 0: (b.xfer2_out[0], b.xfer2_bytes), XFER2_OUT + 1: (b.xfer2_out[1], b.xfer2_bytes),
                XFER2_IN + 0: (b.xfer2_in[0], b.xfer2_bytes), XFER2_IN + 1: (b.xfer2_in[1], b.xfer2_bytes),
                FAR_OUT: (b.far_out, b.far_bytes), FAR_IN: (b.far_in, b.far_bytes * world)}
        self.t = {slot: torch.as_tensor(_DevPtr(p, n), device="cuda") for slot, (p, n) in ptrs.items() if p and n}
        self.stream = torch_stream
        sysr.set_stream(torch_stream.cuda_stream)
        self._phase_ops = {}     # the operations of a phase never change (fixed buffers, fixed peers): built once, on first use

    def start(self, phase):
        """Post the phase's sends and receives.  RCCL runs them on its own stream once the work
        already enqueued on this context's stream is done (what was packed is complete); this
        stream is NOT held up -- whatever is enqueued next runs beside the transfer -- until
        finish() makes it wait for the arrivals."""
        import torch
        if phase == "xfer":
            # the transfer messages grow on demand (every rank in the same step): post what travels now, a prefix of the buffers
            n = self.s.msg_bytes(XFER_OUT) // 4
            key = ("xfer", n)
            if key not in self._phase_ops:
                self._phase_ops = {k: v for k, v in self._phase_ops.items() if not isinstance(k, tuple)}
                self._phase_ops[key] = self._ops(phase, lambda slot: self.t[slot][:n] if XFER_OUT <= slot < XFER_IN + 2 else self.t[slot])
            ops = self._phase_ops[key]
        else:
            if phase not in self._phase_ops:
                self._phase_ops[phase] = self._ops(phase, lambda slot: self.t[slot])
            ops = self._phase_ops[phase]
        if not ops:
            return []
        with torch.cuda.stream(self.stream):       # RCCL orders against torch's CURRENT stream: make it the context's
            return self.dist.batch_isend_irecv(ops)

    def finish(self, works):
        import torch
        with torch.cuda.stream(self.stream):
            for w in works:
                w.wait()        # a stream-side wait (the host goes on)

    def exchange(self, phase):
        self.finish(self.start(phase))

    def gather_status(self):
        """start the all-gather of the status records (needed by the first pair-stage call); returns the work or None"""
        import torch
        if self.dist is None or STATUS_OUT not in self.t or self.world == 1:
            return None
        with torch.cuda.stream(self.stream):
            return self.dist.all_gather_into_tensor(self.t[STATUS_IN], self.t[STATUS_OUT], async_op=True)

    def finish_status(self, work):
        import torch
        if work is not None:
            with torch.cuda.stream(self.stream):
                work.wait()

    def gather_snapshot(self):
        """all-pairs forces: start the all-gather of the snapshot blocks (RCCL ncclAllGather over xGMI); the
        pair stage needs it, so finish_snapshot() comes before slab_pairs"""
        import torch
        if self.dist is None or ALLG_OUT not in self.t or self.world == 1:
            return None
        with torch.cuda.stream(self.stream):
            return self.dist.all_gather_into_tensor(self.t[ALLG_IN], self.t[ALLG_OUT], async_op=True)

    finish_snapshot = finish_status

    def gather_far(self):
        """start the all-gather of the far outboxes (beside the transfer messages; slab_finish needs it)"""
        import torch
        if self.dist is None or FAR_OUT not in self.t or self.world == 1:
            return None
        with torch.cuda.stream(self.stream):
            return self.dist.all_gather_into_tensor(self.t[FAR_IN], self.t[FAR_OUT], async_op=True)

    finish_far = finish_status

    def step(self):
        import torch
        s = self.s
        with torch.cuda.stream(self.stream):
            s.slab_build()
            status = self.gather_status()      # first: the pair stage's chunk census needs every rank's record
            halo = self.start("halo")
            snap = self.gather_snapshot()      # all-pairs forces only
            self.finish_status(status)
            if self.overlap_interior:
                s.slab_pairs_interior()      # cells whose stencil lies in the own layers: no halo needed
            self.finish(halo)
            self.finish_snapshot(snap)
            s.slab_pairs()
            self.exchange("force")
            s.slab_apply()
            xfer = self.start("xfer")
            far = self.gather_far()            # births on, four or more ranks: records for ranks the messages do not reach
            self.finish(xfer)
            self.finish_far(far)
            s.slab_finish()


def merge_owned(arrays, plans, kind="slots"):
    """Assemble a whole-container array from the ranks' downloads: every slot (or QUEUE_INFO
    record, kind="records") is taken from the rank that owns it."""
    out = np.zeros(arrays[0].shape, arrays[0].dtype)    # (copies / zeros_like of a record array leave its pad bytes undefined)
    out[...] = arrays[0]
    for a, p in zip(arrays, plans):
        lo, hi = (p.slot_lo, p.slot_hi) if kind == "slots" else (p.rec_lo, p.rec_hi)
        for t in range(4):
            out[lo[t]:hi[t]] = a[lo[t]:hi[t]]
    return out
