"""particlesystem_amd -- MI355X-native step of abraj/particleSystem behind a C ABI.

This package is a thin ctypes mirror of ``include/psamd.h`` (the drop-in boundary):
the product is ``libpsamd.so`` (hand-written HIP kernels for gfx950 + a C++ host
context).  The Python layer exists for tests, the benchmark and torch.distributed
plumbing; it never computes anything itself and there is no CPU fallback: if the
library is not built, or no HIP device is visible, calls fail loudly.

Stage names follow the reference's task list (particleSystem.cpp:2269-2282):
``init_iframe`` (task 3), ``build_grid`` (task 8), ``calc_forces`` (task 6),
``fill_particles`` (task 5).
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSAMD_LIB") or os.path.join(HERE, "libpsamd.so")   # PSAMD_LIB: another build, for A/B measurements

ABI_VERSION = 6         # the struct layouts below are include/psamd.h's at this PSAMD_ABI_VERSION
MAX_RANKS = 64
FLAG_EXPLOSIONS = 0x1
FLAG_FAST_MATH = 0x2
FLAG_ALL_PAIRS = 0x4
FLAG_EULER = 0x8
NUM_TIMERS = 9
TIMER_NAMES = ("hist", "scan", "scatter", "sort_cells", "pairs", "apply", "lifecycle", "init_iframe", "collide")

# numpy images of the reference's records (common.h:94-145)
P_DTYPE = np.dtype({
    "names": ["id", "cell", "chunk", "seg_type", "seg_tid", "seg_fault", "is_parent",
              "w", "age", "fertility_age", "x", "y", "z", "vx", "vy", "vz", "ax", "ay", "az"],
    "formats": ["<i4"] * 5 + ["u1", "u1"] + ["<f4"] * 12,
    "offsets": [0, 4, 8, 12, 16, 20, 21] + list(range(24, 72, 4)),
    "itemsize": 72,
})
T_DTYPE = np.dtype([("id", "<i4"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("age", "<f4")])
Q_DTYPE = np.dtype([("front", "<i4"), ("rear", "<i4"), ("count", "<i4"), ("lock", "<i4"),
                    ("rloc", "<i4"), ("seg_size", "<i4")])


class Config(C.Structure):
    _fields_ = [("max_particles_num", C.c_int32), ("x_factor", C.c_int32),
                ("chunk_factor", C.c_int32), ("chunk_dim", C.c_int32),
                ("cell_size", C.c_double), ("eps2", C.c_double), ("collision_radius", C.c_double),
                ("particle_weight", C.c_double), ("dt", C.c_double), ("max_v", C.c_double),
                ("explosion_speed", C.c_double), ("life_steps", C.c_double),
                ("device", C.c_int32), ("flags", C.c_uint32), ("seed", C.c_uint64),
                ("rank", C.c_int32), ("world", C.c_int32),
                ("halo_cap_cell", C.c_int32), ("xfer_cap", C.c_int32),
                ("cuts", C.c_int32 * (MAX_RANKS + 1)),
                ("drag", C.c_double), ("force_sign", C.c_double),
                ("xfer_cap_max", C.c_int32), ("reserved0", C.c_int32)]


class Sizes(C.Structure):
    _fields_ = [("grid_dim", C.c_int32), ("num_cells", C.c_int32), ("num_chunks", C.c_int32),
                ("cells_per_chunk", C.c_int32), ("max_per_cell", C.c_int32), ("max_per_chunk", C.c_int32),
                ("container_size", C.c_int32), ("queue_info_size", C.c_int32),
                ("n_chunkgrid", C.c_int64), ("n_cellgrid", C.c_int64), ("n_pkgdistrib", C.c_int32),
                ("seg_count", C.c_int32 * 4), ("seg_size_t", C.c_int32 * 4), ("seg_size", C.c_int32 * 4)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in
                ("deaths_age", "deaths_collision", "survives", "integrated", "relocations",
                 "relocations_lost", "births", "births_failed", "cell_overflow_kills", "steps",
                 "particles_processed", "max_ops_one_queue")]


class DeviceView(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("pos4", "vel4", "acc4", "cell", "pflags", "sorted_id", "snap_soa", "force4", "cell_start")] + \
               [("container_size", C.c_int64), ("num_cells", C.c_int32), ("live", C.c_int32),
                ("sorted_cap", C.c_int64), ("stream", C.c_void_p)]


class SlabPlan(C.Structure):
    """psamd_slab_plan: which cell layers, slots and queue records one rank holds."""
    _fields_ = [(n, C.c_int32) for n in
                ("world", "rank", "grid_dim", "cut_lo", "cut_hi", "state_lo", "state_hi", "below_lo", "below_hi",
                 "above_lo", "above_hi", "lentin_lo", "lentin_hi", "lentout_lo", "lentout_hi",
                 "send_up_lo", "send_up_hi", "send_down_lo", "send_down_hi")] + \
               [("slot_lo", C.c_int32 * 4), ("slot_hi", C.c_int32 * 4), ("rec_lo", C.c_int32 * 4), ("rec_hi", C.c_int32 * 4),
                ("up_rank", C.c_int32), ("down_rank", C.c_int32)]


class SlabBuffers(C.Structure):
    _fields_ = [("halo_out", C.c_void_p * 2), ("halo_in", C.c_void_p * 2),
                ("halo_out_bytes", C.c_int64 * 2), ("halo_in_bytes", C.c_int64 * 2),
                ("force_out", C.c_void_p), ("force_in", C.c_void_p),
                ("force_out_bytes", C.c_int64), ("force_in_bytes", C.c_int64),
                ("xfer_out", C.c_void_p * 2), ("xfer_in", C.c_void_p * 2), ("xfer_bytes", C.c_int64),
                ("status_out", C.c_void_p), ("status_in", C.c_void_p), ("status_bytes", C.c_int64),
                ("allg_out", C.c_void_p), ("allg_in", C.c_void_p), ("allg_bytes", C.c_int64),
                ("xfer2_out", C.c_void_p * 2), ("xfer2_in", C.c_void_p * 2), ("xfer2_bytes", C.c_int64),
                ("far_out", C.c_void_p), ("far_in", C.c_void_p), ("far_bytes", C.c_int64), ("xfer_bytes_max", C.c_int64)]


# psamd_slab_msg_download / _upload `which`
MSG_HALO_OUT, MSG_HALO_IN, MSG_FORCE_OUT, MSG_FORCE_IN, MSG_XFER_OUT, MSG_XFER_IN, MSG_STATUS_OUT, MSG_STATUS_IN = 0, 2, 4, 5, 6, 8, 10, 11
MSG_ALLG_OUT, MSG_ALLG_IN = 12, 13
MSG_XFER2_OUT, MSG_XFER2_IN = 14, 16
MSG_FAR_OUT, MSG_FAR_IN = 18, 19


class PsamdError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("psamd status %d: %s" % (status, message))
        self.status = status


# every entry point include/psamd.h declares: (name, restype, argtypes)
_vp, _i32, _i64 = C.c_void_p, C.c_int32, C.c_int64
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
ABI = [
    ("psamd_abi_version", C.c_int, []),
    ("psamd_status_string", C.c_char_p, [C.c_int]),
    ("psamd_default_config", C.c_int, [C.POINTER(Config)]),
    ("psamd_create", C.c_int, [C.POINTER(Config), C.POINTER(_vp)]),
    ("psamd_destroy", C.c_int, [_vp]),
    ("psamd_last_error", C.c_char_p, [_vp]),
    ("psamd_get_sizes", C.c_int, [_vp, C.POINTER(Sizes)]),
    ("psamd_describe", C.c_int, [C.POINTER(Config), C.POINTER(Sizes), _vp, _vp, _vp, _vp]),
    ("psamd_get_config", C.c_int, [_vp, C.POINTER(Config)]),
    ("psamd_fill_particles", C.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64)]),
    ("psamd_uniform_cloud", C.c_int, [_vp, _i64, C.c_uint32, _vp]),
    ("psamd_upload_particles", C.c_int, [_vp, _vp, _i64, _i64]),
    ("psamd_download_particles", C.c_int, [_vp, _vp, _i64, _i64]),
    ("psamd_download_tdata", C.c_int, [_vp, _vp, _i64, _i64]),
    ("psamd_upload_queues", C.c_int, [_vp, _vp, _vp]),
    ("psamd_download_queues", C.c_int, [_vp, _vp, _vp]),
    ("psamd_download_cellgrid", C.c_int, [_vp, _vp]),
    ("psamd_download_chunkgrid", C.c_int, [_vp, _vp]),
    ("psamd_download_force_counts", C.c_int, [_vp, _vp]),
    ("psamd_get_pkgdistrib", C.c_int, [_vp, _vp]),
    ("psamd_get_cell_table", C.c_int, [_vp, _vp]),
    ("psamd_get_gridmax", C.c_int, [_vp, _ip]),
    ("psamd_init_iframe", C.c_int, [_vp]),
    ("psamd_build_grid", C.c_int, [_vp]),
    ("psamd_calc_forces", C.c_int, [_vp]),
    ("psamd_calc_forces_pairs", C.c_int, [_vp]),
    ("psamd_calc_forces_apply", C.c_int, [_vp]),
    ("psamd_step", C.c_int, [_vp, _i32]),
    ("psamd_synchronize", C.c_int, [_vp]),
    ("psamd_download_force4", C.c_int, [_vp, _vp, _i64, _i64]),
    ("psamd_snapshot_save", C.c_int, [_vp]),
    ("psamd_snapshot_restore", C.c_int, [_vp]),
    ("psamd_set_stream", C.c_int, [_vp, _vp]),
    ("psamd_get_stream", C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    ("psamd_slab_plan_describe", C.c_int, [C.POINTER(Config), C.POINTER(SlabPlan)]),
    ("psamd_get_slab_plan", C.c_int, [_vp, C.POINTER(SlabPlan)]),
    ("psamd_slab_buffers_get", C.c_int, [_vp, C.POINTER(SlabBuffers)]),
    ("psamd_slab_build", C.c_int, [_vp]),
    ("psamd_slab_pairs_interior", C.c_int, [_vp]),
    ("psamd_slab_pairs", C.c_int, [_vp]),
    ("psamd_slab_apply", C.c_int, [_vp]),
    ("psamd_slab_finish", C.c_int, [_vp]),
    ("psamd_slab_msg_download", C.c_int, [_vp, C.c_int, _vp, _i64]),
    ("psamd_slab_msg_upload", C.c_int, [_vp, C.c_int, _vp, _i64]),
    ("psamd_get_counters", C.c_int, [_vp, C.POINTER(Counters)]),
    ("psamd_live_count", C.c_int, [_vp, C.POINTER(_i64)]),
    ("psamd_device_view_get", C.c_int, [_vp, C.POINTER(DeviceView)]),
    ("psamd_debug_wave_trace", C.c_int, [_vp, _vp, _i64]),
    ("psamd_selftest_math", C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("psamd_set_graphs", C.c_int, [_vp, C.c_int]),
    ("psamd_get_graph_stats", C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    ("psamd_set_wait_policy", C.c_int, [_vp, C.c_int]),
    ("psamd_set_timing", C.c_int, [_vp, C.c_int]),
    ("psamd_set_timing_period", C.c_int, [_vp, C.c_int]),
    ("psamd_get_timing", C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(_i64)]),
    ("psamd_get_timing_stats", C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i64)]),
    ("psamd_set_run_ahead", C.c_int, [_vp, C.c_int]),
    ("psamd_set_tdata_mirror", C.c_int, [_vp, C.c_int]),
]

_lib = None


def build(force=False):
    """Compile libpsamd.so for gfx950 (hipcc cross-compiles without a GPU)."""
    return _build.build(force=force)


def load():
    """Load libpsamd.so and bind every ABI symbol. Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libpsamd.so is not built (run particlesystem_amd.build()); "
                               "the HIP library is the only implementation, there is no fallback")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(lib, name)  # AttributeError here = a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        have = lib.psamd_abi_version()
        if have != ABI_VERSION:
            # (PSAMD_LIB may point at another build: a library with other struct layouts would misread the
            # configuration and hand out garbage message pointers)
            raise RuntimeError("%s has ABI version %d, this module is written for %d: rebuild it (particlesystem_amd.build(force=True))"
                               % (LIB_PATH, have, ABI_VERSION))
        _lib = lib
    return _lib


def default_config(**over):
    cfg = Config()
    load().psamd_default_config(C.byref(cfg))
    for k, v in over.items():
        if k == "cuts":
            for i, c in enumerate(v):
                cfg.cuts[i] = c
        else:
            setattr(cfg, k, v)
    return cfg


def slab_plan(cfg):
    """Host-only: the slab plan of cfg.rank in a world of cfg.world ranks (no GPU needed)."""
    lib = load()
    plan = SlabPlan()
    st = lib.psamd_slab_plan_describe(C.byref(cfg), C.byref(plan))
    if st != 0:
        raise PsamdError(st, lib.psamd_status_string(st).decode())
    return plan


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def describe(cfg):
    """Host-only geometry of a configuration: sizes, cell table [num_cells, 3], package
    table [num_chunks, 54], initial QUEUE_INFO records and queue array.  Needs no GPU."""
    lib = load()
    sizes = Sizes()
    st = lib.psamd_describe(C.byref(cfg), C.byref(sizes), None, None, None, None)
    if st != 0:
        raise PsamdError(st, lib.psamd_status_string(st).decode())
    table = np.zeros((sizes.num_cells, 3), np.int32)
    pkg = np.zeros((sizes.num_chunks, 54), np.int32)
    qi = np.zeros(sizes.queue_info_size, Q_DTYPE)
    q = np.zeros(sizes.container_size, np.int32)
    st = lib.psamd_describe(C.byref(cfg), None, _ptr(table), _ptr(pkg), _ptr(qi), _ptr(q))
    if st != 0:
        raise PsamdError(st, lib.psamd_status_string(st).decode())
    return sizes, table, pkg, qi, q


class ParticleSystem:
    """One psamd context: the reference's nine buffers, resident on one MI355X."""

    def __init__(self, cfg=None, **over):
        self.lib = load()
        self.cfg = cfg if cfg is not None else default_config(**over)
        h = C.c_void_p()
        st = self.lib.psamd_create(C.byref(self.cfg), C.byref(h))
        self.h = h
        if st != 0:
            msg = self.lib.psamd_last_error(h).decode() if h else self.lib.psamd_status_string(st).decode()
            if h:
                self.lib.psamd_destroy(h)
            self.h = None
            raise PsamdError(st, msg)
        self.sizes = Sizes()
        self._ck(self.lib.psamd_get_sizes(self.h, C.byref(self.sizes)))

    def _ck(self, st):
        if st != 0:
            raise PsamdError(st, self.lib.psamd_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.lib.psamd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup ------------------------------------------------------------
    def uniform_cloud(self, n, seed):
        xyz = np.empty((n, 3), np.float32)
        self._ck(self.lib.psamd_uniform_cloud(self.h, n, seed, _ptr(xyz)))
        return xyz

    def fill_particles(self, xyz, age=None, fert_age=None, w=None, vxyz=None):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        n = len(xyz)

        def arr(v):
            if v is None:
                return None
            return np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float32), (n,)))
        age, fert_age, w = arr(age), arr(fert_age), arr(w)
        vxyz = None if vxyz is None else np.ascontiguousarray(vxyz, np.float32).reshape(-1, 3)
        ids = np.empty(n, np.int32)
        done = C.c_int64()
        self._ck(self.lib.psamd_fill_particles(self.h, n, _ptr(xyz), _ptr(vxyz), _ptr(w), _ptr(age),
                                               _ptr(fert_age), _ptr(ids), C.byref(done)))
        return ids

    # ---- reference-layout buffers ------------------------------------------
    def upload_particles(self, p, first=0):
        p = np.ascontiguousarray(p)
        assert p.dtype == P_DTYPE
        self._ck(self.lib.psamd_upload_particles(self.h, _ptr(p), first, len(p)))

    def download_particles(self, first=0, count=None):
        count = self.sizes.container_size - first if count is None else count
        p = np.zeros(count, P_DTYPE)
        self._ck(self.lib.psamd_download_particles(self.h, _ptr(p), first, count))
        return p

    def download_tdata(self, first=0, count=None):
        count = self.sizes.container_size - first if count is None else count
        t = np.zeros(count, T_DTYPE)
        self._ck(self.lib.psamd_download_tdata(self.h, _ptr(t), first, count))
        return t

    def upload_queues(self, queue_info, queue):
        qi = np.ascontiguousarray(queue_info)
        q = np.ascontiguousarray(queue, np.int32)
        assert qi.dtype == Q_DTYPE and len(qi) == self.sizes.queue_info_size and len(q) == self.sizes.container_size
        self._ck(self.lib.psamd_upload_queues(self.h, _ptr(qi), _ptr(q)))

    def download_queues(self):
        qi = np.zeros(self.sizes.queue_info_size, Q_DTYPE)
        q = np.zeros(self.sizes.container_size, np.int32)
        self._ck(self.lib.psamd_download_queues(self.h, _ptr(qi), _ptr(q)))
        return qi, q

    def download_cellgrid(self):
        out = np.zeros(self.sizes.n_cellgrid, np.int32)
        self._ck(self.lib.psamd_download_cellgrid(self.h, _ptr(out)))
        return out.reshape(self.sizes.num_cells, -1)

    def download_chunkgrid(self):
        out = np.zeros(self.sizes.n_chunkgrid, np.int32)
        self._ck(self.lib.psamd_download_chunkgrid(self.h, _ptr(out)))
        return out.reshape(self.sizes.num_chunks, -1)

    def download_force_counts(self):
        """Per cell: particles the last pair pass computed a force for."""
        out = np.zeros(self.sizes.num_cells, np.int32)
        self._ck(self.lib.psamd_download_force_counts(self.h, _ptr(out)))
        return out

    def pkgdistrib(self):
        out = np.zeros(self.sizes.n_pkgdistrib * 2, np.int32)
        self._ck(self.lib.psamd_get_pkgdistrib(self.h, _ptr(out)))
        return out.reshape(self.sizes.num_chunks, 54)

    def cell_table(self):
        out = np.zeros(self.sizes.num_cells * 3, np.int32)
        self._ck(self.lib.psamd_get_cell_table(self.h, _ptr(out)))
        return out.reshape(-1, 3)

    def gridmax(self):
        out = np.zeros(2, np.int32)
        self._ck(self.lib.psamd_get_gridmax(self.h, out.ctypes.data_as(_ip)))
        return out

    # ---- stages -------------------------------------------------------------
    def init_iframe(self):
        self._ck(self.lib.psamd_init_iframe(self.h))

    def build_grid(self):
        self._ck(self.lib.psamd_build_grid(self.h))

    def calc_forces(self):
        self._ck(self.lib.psamd_calc_forces(self.h))

    def calc_forces_pairs(self):
        self._ck(self.lib.psamd_calc_forces_pairs(self.h))

    def calc_forces_apply(self):
        self._ck(self.lib.psamd_calc_forces_apply(self.h))

    # ---- slab stages (multi-GPU; see include/psamd.h "slab partition") ------
    def slab_plan(self):
        plan = SlabPlan()
        self._ck(self.lib.psamd_get_slab_plan(self.h, C.byref(plan)))
        return plan

    def slab_buffers(self):
        b = SlabBuffers()
        self._ck(self.lib.psamd_slab_buffers_get(self.h, C.byref(b)))
        return b

    def slab_build(self):
        self._ck(self.lib.psamd_slab_build(self.h))

    def slab_pairs_interior(self):
        self._ck(self.lib.psamd_slab_pairs_interior(self.h))

    def slab_pairs(self):
        self._ck(self.lib.psamd_slab_pairs(self.h))

    def slab_apply(self):
        self._ck(self.lib.psamd_slab_apply(self.h))

    def slab_finish(self):
        self._ck(self.lib.psamd_slab_finish(self.h))

    def msg_bytes(self, which):
        """Size of message buffer `which` (psamd_slab_msg_download numbering); 0: no such message.  The transfer messages
        (6-9) may grow from step to step (config.xfer_cap_max): their size is asked for afresh."""
        if 6 <= which <= 9:
            return self.slab_buffers().xfer_bytes
        if getattr(self, "_msg_bytes", None) is None:
            b = self.slab_buffers()
            self._msg_bytes = [b.halo_out_bytes[0], b.halo_out_bytes[1], b.halo_in_bytes[0], b.halo_in_bytes[1],
                               b.force_out_bytes, b.force_in_bytes] + [b.xfer_bytes] * 4 + \
                              [b.status_bytes, b.status_bytes * max(1, self.cfg.world),
                               b.allg_bytes, b.allg_bytes * max(1, self.cfg.world)] + [b.xfer2_bytes] * 4 + \
                              [b.far_bytes, b.far_bytes * max(1, self.cfg.world)]
        return self._msg_bytes[which]

    def msg_download(self, which, nbytes=None):
        nbytes = self.msg_bytes(which) if nbytes is None else nbytes
        out = np.zeros(nbytes // 4, np.int32)
        self._ck(self.lib.psamd_slab_msg_download(self.h, which, _ptr(out), nbytes))
        return out

    def msg_upload(self, which, words):
        words = np.ascontiguousarray(words, np.int32)
        self._ck(self.lib.psamd_slab_msg_upload(self.h, which, _ptr(words), words.nbytes))

    def step(self, n=1):
        self._ck(self.lib.psamd_step(self.h, n))

    def synchronize(self):
        self._ck(self.lib.psamd_synchronize(self.h))

    def download_force4(self, first, count):
        out = np.zeros((count, 4), np.float32)
        self._ck(self.lib.psamd_download_force4(self.h, _ptr(out), first, count))
        return out

    def snapshot_save(self):
        self._ck(self.lib.psamd_snapshot_save(self.h))

    def snapshot_restore(self):
        self._ck(self.lib.psamd_snapshot_restore(self.h))

    def set_stream(self, hip_stream):
        self._ck(self.lib.psamd_set_stream(self.h, hip_stream))

    # ---- introspection ------------------------------------------------------
    @property
    def counters(self):
        c = Counters()
        self._ck(self.lib.psamd_get_counters(self.h, C.byref(c)))
        return {n: getattr(c, n) for n, _ in Counters._fields_}

    def live_count(self):
        n = C.c_int64()
        self._ck(self.lib.psamd_live_count(self.h, C.byref(n)))
        return n.value

    def device_view(self):
        v = DeviceView()
        self._ck(self.lib.psamd_device_view_get(self.h, C.byref(v)))
        return v

    def wave_trace(self):
        n = 3 * (self.sizes.num_cells * ((self.sizes.max_per_cell + 63) // 64) + 4)
        out = np.zeros(n, np.uint64)
        self._ck(self.lib.psamd_debug_wave_trace(self.h, _ptr(out), n))
        return out.reshape(-1, 3)

    def selftest_math(self, lo_bits, hi_bits):
        out = (C.c_uint64 * 24)()
        self._ck(self.lib.psamd_selftest_math(self.h, lo_bits, hi_bits, out))
        return list(out)

    def set_graphs(self, on=True):
        """stage sequences as hipGraphs: one submission per stage instead of one per kernel"""
        self._ck(self.lib.psamd_set_graphs(self.h, 1 if on else 0))

    def graph_stats(self):
        """(replays, captures); raises if the runtime refused to capture and the context fell back"""
        a, b = C.c_int64(), C.c_int64()
        self._ck(self.lib.psamd_get_graph_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_wait_policy(self, policy):
        self._ck(self.lib.psamd_set_wait_policy(self.h, int(policy)))

    def set_tdata_mirror(self, on):
        """build_grid also writes the reference's T_DATA rows (needed by download_tdata only); default on"""
        self._ck(self.lib.psamd_set_tdata_mirror(self.h, 1 if on else 0))

    def set_run_ahead(self, steps):
        """1 (default): a call that ends a step returns once the step BEFORE has reported; 0: waits for its own step"""
        self._ck(self.lib.psamd_set_run_ahead(self.h, int(steps)))

    def set_timing(self, on=True, every_stage=False, period=1):
        """HIP-event timing of the step's kernels: pair pass, apply and life cycle, or every
        stage (an event between two kernels costs ~6 us of idle GPU each); period n: on every
        n-th step only."""
        self._ck(self.lib.psamd_set_timing_period(self.h, int(period)))
        self._ck(self.lib.psamd_set_timing(self.h, (2 if every_stage else 1) if on else 0))

    def timing(self):
        us = (C.c_double * NUM_TIMERS)()
        n = C.c_int64()
        self._ck(self.lib.psamd_get_timing(self.h, us, C.byref(n)))
        return dict(zip(TIMER_NAMES, list(us))), n.value

    def timing_stats(self):
        """({timer: median us}, {timer: max us}, samples) over the timed steps"""
        med, mx = (C.c_double * NUM_TIMERS)(), (C.c_double * NUM_TIMERS)()
        n = C.c_int64()
        self._ck(self.lib.psamd_get_timing_stats(self.h, med, mx, C.byref(n)))
        return dict(zip(TIMER_NAMES, list(med))), dict(zip(TIMER_NAMES, list(mx))), n.value
