"""CPU stand-in for ONE RANK of the slab-partitioned path, built from the oracle (test
infrastructure; the product has no CPU path).  It offers the stage calls and message slots of
particlesystem_amd.ParticleSystem, so particlesystem_amd.slab's transports can drive it on a
machine without a GPU: the world_size-2 gloo test runs bench.py's multi-GPU orchestration this
way.  It is honest about ownership: its oracle system contains ONLY the particles of the rank's
own segments; everything it knows about other layers arrived in a message.

Message formats are private to this stand-in (arrays of int32 like the product's, same slot
numbering, fixed sizes)."""
import numpy as np

import oracle_py as O
import particlesystem_amd as ps
from particlesystem_amd.slab import ALLG_IN, ALLG_OUT, FORCE_IN, FORCE_OUT, HALO_IN, HALO_OUT, XFER_IN, XFER_OUT

OP_WORDS = O.OP_DTYPE.itemsize // 4


def group_of_layer(i3, D):
    r, k = i3 % D, i3 // D
    return 2 * k if r == 0 else 2 * (k + 1) if r == D - 1 else 2 * k + 1


class OracleSlabRank:
    def __init__(self, cfg, xfer_cap=4096, all_pairs=False):
        """cfg: a particlesystem_amd Config (rank, world and the reference constants).
        all_pairs: the stand-in of PSAMD_FLAG_ALL_PAIRS across ranks -- every rank contributes the
        snapshot of its own cells to an all-gather (slots ALLG_OUT / ALLG_IN) and adds, to the cutoff
        forces of the particles it computes, the field of every cell beyond their stencil, summed per
        cell in fp32 and then over the cells in global order (numpy; what is compared is a world of N
        stand-ins against a world of one, which see the same arrays in the same order)."""
        self.cfg = cfg
        self.all_pairs = all_pairs
        self.plan = ps.slab_plan(cfg)                      # host-only geometry, no GPU involved
        ocfg = O.default_config(**{k: getattr(cfg, k) for k in
                                   ("max_particles_num", "x_factor", "chunk_factor", "chunk_dim", "cell_size", "eps2",
                                    "collision_radius", "particle_weight", "dt", "max_v", "explosion_speed", "life_steps")})
        self.o = O.System(ocfg)
        d = self.o.d
        self.G, self.GG, self.mpc = d.grid_dim, d.grid_dim * d.grid_dim, d.max_per_cell
        self.D = cfg.chunk_dim
        p = self.plan
        self.owned_rec = np.zeros(d.queue_info_size, bool)
        for t in range(4):
            self.owned_rec[p.rec_lo[t]:p.rec_hi[t]] = True
        self.xfer_cap = xfer_cap
        lay = lambda lo, hi: max(0, hi - lo) * self.GG
        halo_words = lambda cells: 16 + cells + 6 * cells * self.mpc if cells else 0
        force_words = lambda cells: 16 + 4 * cells * self.mpc if cells else 0
        xfer_words = (16 + xfer_cap * OP_WORDS) if p.world > 1 else 0
        self.sizes = {HALO_OUT + 0: halo_words(lay(p.send_down_lo, p.send_down_hi)), HALO_OUT + 1: halo_words(lay(p.send_up_lo, p.send_up_hi)),
                      HALO_IN + 0: halo_words(lay(p.below_lo, p.below_hi)), HALO_IN + 1: halo_words(lay(p.above_lo, p.above_hi)),
                      FORCE_OUT: force_words(lay(p.lentin_lo, p.lentin_hi)), FORCE_IN: force_words(lay(p.lentout_lo, p.lentout_hi)),
                      XFER_OUT + 0: xfer_words, XFER_OUT + 1: xfer_words, XFER_IN + 0: xfer_words, XFER_IN + 1: xfer_words,
                      10: 0, 11: 0,        # no status record: this stand-in neither overflows cells nor fails
                      ALLG_OUT: 0, ALLG_IN: 0}
        if all_pairs:
            plans = [ps.slab_plan(ps.default_config(**dict({k: getattr(cfg, k) for k in ("max_particles_num", "x_factor", "chunk_factor", "chunk_dim")},
                                                           rank=r, world=p.world))) for r in range(p.world)]
            self.ag_cells = max(q.state_hi - q.state_lo for q in plans) * self.GG
            self.ag_cap = int(cfg.max_particles_num)
            self.ag_words = 16 + self.ag_cells + 5 * self.ag_cap
            self.sizes[ALLG_OUT] = self.ag_words
            self.sizes[ALLG_IN] = self.ag_words * p.world
        self.msgs = {k: np.zeros(n, np.int32) for k, n in self.sizes.items()}
        self.sent = 0

    # ---- the ParticleSystem interface the transports use ----
    def msg_bytes(self, which):
        return 4 * self.sizes[which]

    def msg_download(self, which):
        return self.msgs[which].copy()

    def msg_upload(self, which, words):
        self.msgs[which][:] = np.asarray(words, np.int32)

    def slab_plan(self):
        return self.plan

    def fill_particles(self, xyz, age, fert_age):
        """only the particles of the own segments, in input order (their queues are this rank's)"""
        xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
        n = len(xyz)
        age = np.broadcast_to(np.asarray(age, np.float32), (n,))
        fert = np.broadcast_to(np.asarray(fert_age, np.float32), (n,))
        i3 = (np.floor(-xyz[:, 2].astype(np.float64) / self.cfg.cell_size) + self.G // 2).astype(int)
        mine = (i3 >= self.plan.state_lo) & (i3 < self.plan.state_hi)
        ids = np.full(n, -1, np.int32)
        ids[mine] = self.o.fill(xyz[mine], age=age[mine], fert_age=fert[mine])
        return ids

    # ---- helpers ----
    def _cells(self, lo, hi):
        return range(lo * self.GG, hi * self.GG)

    def _pack_layers(self, lo, hi, which):
        m = self.msgs[which]
        cells = list(self._cells(lo, hi))
        cg, t = self.o.cellgrid, self.o.tdata
        cap = len(cells) * self.mpc
        m[:] = 0
        m[0] = len(cells)
        body = m[16 + len(cells):].reshape(6, cap)
        at = 0
        for j, c in enumerate(cells):
            n = int(cg[c, 0])
            m[16 + j] = n
            ids = cg[c, 1:1 + n]
            rows = t[ids]
            for k, f in enumerate(("x", "y", "z", "w", "age")):
                body[k, at:at + n] = rows[f].view(np.int32)
            body[5, at:at + n] = ids
            at += n
        m[1] = at

    def _unpack_layers(self, lo, hi, which):
        m = self.msgs[which]
        cells = list(self._cells(lo, hi))
        assert m[0] == len(cells), "halo message does not match the plan"
        cap = len(cells) * self.mpc
        body = m[16 + len(cells):].reshape(6, cap)
        cg, t, p = self.o.cellgrid, self.o.tdata, self.o.particles
        at = 0
        for j, c in enumerate(cells):
            n = int(m[16 + j])
            ids = body[5, at:at + n]
            cg[c, 0] = n
            cg[c, 1:1 + n] = ids
            for k, f in enumerate(("x", "y", "z", "w", "age")):
                t[f][ids] = body[k, at:at + n].view(np.float32)
                # the pair pass reads "me" from the particle record: lend it the snapshot values
                # (cell stays -1: the record never counts as a particle of this rank)
                p[f][ids] = body[k, at:at + n].view(np.float32)
            t["id"][ids] = ids
            self.borrowed.append(ids.copy())
            at += n

    def _sorted_start(self):
        return np.concatenate([[0], np.cumsum(self.o.cellgrid[:, 0])])

    # ---- the four stages ----
    def slab_build(self):
        p = self.plan
        self.o.init_iframe(); self.o.build_grid()
        if self.sizes[HALO_OUT + 0]:
            self._pack_layers(p.send_down_lo, p.send_down_hi, HALO_OUT + 0)
        if self.sizes[HALO_OUT + 1]:
            self._pack_layers(p.send_up_lo, p.send_up_hi, HALO_OUT + 1)
        if self.all_pairs:
            self._pack_snapshot()

    # ---- all-pairs: the snapshot block of the own cells, and the far field from all ranks' blocks ----
    def _pack_snapshot(self):
        p, m = self.plan, self.msgs[ALLG_OUT]
        cells = list(self._cells(p.state_lo, p.state_hi))
        cg, t = self.o.cellgrid, self.o.tdata
        m[:] = 0
        m[0], m[3] = len(cells), cells[0]
        body = m[16 + self.ag_cells:].reshape(5, self.ag_cap)
        at = 0
        for j, c in enumerate(cells):
            n = int(cg[c, 0])
            m[16 + j] = n
            rows = t[cg[c, 1:1 + n]]
            for k, f in enumerate(("x", "y", "z", "w", "age")):
                body[k, at:at + n] = rows[f].view(np.int32)
            at += n
        m[1] = at

    def _far_field(self, st):
        """adds, for every particle this rank computes that feels a force, the field of the cells beyond its stencil"""
        p, G, GG = self.plan, self.G, self.GG
        blocks = self.msgs[ALLG_IN] if p.world > 1 else self.msgs[ALLG_OUT]
        ncells_total = G * GG
        count = np.zeros(ncells_total, np.int64)
        planes = []
        for r in range(p.world if p.world > 1 else 1):
            b = blocks[r * self.ag_words:(r + 1) * self.ag_words]
            nc, nb, first = int(b[0]), int(b[1]), int(b[3])
            count[first:first + nc] = b[16:16 + nc]
            planes.append(b[16 + self.ag_cells:].reshape(5, self.ag_cap)[:, :nb].view(np.float32))
        bx, by, bz, bw, bage = np.concatenate(planes, axis=1)             # ranks hold ascending layers: global cell order
        kid = np.float32(self.cfg.life_steps * self.cfg.dt / 10.0)
        w_eff = np.where(bage < kid, np.float32(0), bw).astype(np.float32)
        start = np.concatenate([[0], np.cumsum(count)])
        cell_of_body = np.repeat(np.arange(ncells_total), count)
        b3, b1, b2 = cell_of_body // GG, (cell_of_body % GG) // G, cell_of_body % G
        seg = start[:-1][count > 0]                                        # reduceat boundaries: the non-empty cells
        eps = np.float32(self.cfg.eps2)
        cg, part = self.o.cellgrid, self.o.particles
        for c in self._cells(p.cut_lo, p.cut_hi):
            n = int(cg[c, 0])
            if n == 0 or len(bx) == 0:
                continue
            i3, i1, i2 = c // GG, (c % GG) // G, c % G
            far = (np.abs(b3 - i3) > 1) | (np.abs(b1 - i1) > 1) | (np.abs(b2 - i2) > 1)
            wf = np.where(far, w_eff, np.float32(0))
            for e in range(n):
                gi = int(st[c]) + e
                pid = int(cg[c, 1 + e])
                if self.force[gi, 3].view(np.int32) != 0 or part["age"][pid] < kid:
                    continue
                rx, ry, rz = bx - part["x"][pid], by - part["y"][pid], bz - part["z"][pid]
                d = (rx * rx + ry * ry + rz * rz + eps).astype(np.float32)
                sc = (wf / (d * np.sqrt(d))).astype(np.float32)
                per_cell = np.stack([np.add.reduceat(rx * sc, seg), np.add.reduceat(ry * sc, seg), np.add.reduceat(rz * sc, seg)], 1)
                self.force[gi, :3] += per_cell.sum(0, dtype=np.float32)

    def slab_pairs_interior(self):
        pass                                   # an optimisation of the product (overlap); nothing to model

    def slab_pairs(self):
        p = self.plan
        self.borrowed = []
        if self.sizes[HALO_IN + 0]:
            self._unpack_layers(p.below_lo, p.below_hi, HALO_IN + 0)
        if self.sizes[HALO_IN + 1]:
            self._unpack_layers(p.above_lo, p.above_hi, HALO_IN + 1)
        st = self._sorted_start()
        self.force = np.zeros((int(st[-1]) + 8, 4), np.float32)
        self.o.calc_pairs(int(st[p.cut_lo * self.GG]), int(st[p.cut_hi * self.GG]), self.force)
        if self.all_pairs:
            self._far_field(st)
        if self.sizes[FORCE_OUT]:
            a, b = int(st[p.lentin_lo * self.GG]), int(st[p.lentin_hi * self.GG])
            m = self.msgs[FORCE_OUT]
            m[:] = 0
            m[0] = m[1] = b - a
            m[16:16 + 4 * (b - a)] = self.force[a:b].view(np.int32).ravel()

    def slab_apply(self):
        p = self.plan
        st = self._sorted_start()
        if self.sizes[FORCE_IN]:
            a, b = int(st[p.lentout_lo * self.GG]), int(st[p.lentout_hi * self.GG])
            m = self.msgs[FORCE_IN]
            assert m[0] == b - a, "force message does not match the lent-out layers"
            self.force[a:b] = m[16:16 + 4 * (b - a)].view(np.float32).reshape(-1, 4)
        ops = self.o.apply_collect(self.force)
        # give the borrowed particle records back
        part = self.o.particles
        for ids in self.borrowed:
            for f in ("x", "y", "z", "w", "age"):
                part[f][ids] = 0
        own = self.owned_rec[ops["rec"]]
        self.own_ops = ops[own]
        away = ops[~own]
        assert (away["kind"] != 0).all(), "an insert can only concern an own queue"
        # one layer up the ring -- or two, when the rounded sum lands exactly on the far face
        layers_up = ((away["body"]["cell"] // self.GG) - (away["old_cell"] // self.GG)) % self.G
        up = (layers_up == 1) | (layers_up == 2)
        for which, sel in ((XFER_OUT + 0, away[~up]), (XFER_OUT + 1, away[up])):
            if not self.sizes[which]:
                assert len(sel) == 0
                continue
            assert len(sel) <= self.xfer_cap
            m = self.msgs[which]
            m[:] = 0
            m[0] = len(sel)
            m[16:16 + len(sel) * OP_WORDS] = np.ascontiguousarray(sel).view(np.int32).ravel()
            self.sent += len(sel)

    def slab_finish(self):
        ops = [self.own_ops]
        for which in (XFER_IN + 0, XFER_IN + 1):
            if self.sizes[which]:
                m = self.msgs[which]
                n = int(m[0])
                ops.append(m[16:16 + n * OP_WORDS].copy().view(O.OP_DTYPE))
        allops = np.concatenate(ops)
        assert self.owned_rec[allops["rec"]].all(), "an operation arrived at a rank that does not own its queue"
        self.o.replay_ops(allops)
        self.o.advance_step()

    # ---- for the comparison with the single-system run ----
    def download_particles(self):
        return self.o.particles.copy()

    def download_queues(self):
        return self.o.queue_info.copy(), self.o.queue.copy()

    @property
    def counters(self):
        return self.o.counters
