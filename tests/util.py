"""Shared helpers for the parity tests (oracle side is test infrastructure)."""
import os

import numpy as np

import oracle_py as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
M64 = (1 << 64) - 1


def g2_cloud():
    return np.fromfile(os.path.join(GOLDEN, "g2_cloud_n4096_seed12345.f32"), dtype=np.float32).reshape(-1, 3)


def cloud(n, seed, half=40.0):
    rng = np.random.default_rng(seed)
    return rng.uniform(-half, half, (n, 3)).astype(np.float32)


def oracle_cfg_from(cfg):
    """psamd Config -> oracle Config (same reference constants)."""
    return O.default_config(**{k: getattr(cfg, k) for k in
                               ("max_particles_num", "x_factor", "chunk_factor", "chunk_dim", "cell_size",
                                "eps2", "collision_radius", "particle_weight", "dt", "max_v",
                                "explosion_speed", "life_steps")})


def splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def explosion_rng(seed):
    """The product's counter-based explosion RNG restated for the oracle's callback
    (apply.hip k_apply / lifecycle.hip commit_move): keyed on (seed, step, parent id)."""
    def fn(pid, step):
        h0 = splitmix64(seed ^ ((step & 0xFFFFFFFF) << 32) ^ (pid & 0xFFFFFFFF))
        h1 = splitmix64(h0)
        h2 = splitmix64(h1)
        h3 = splitmix64(h2)
        ints = tuple(int((h >> 11) * (1.0 / 9007199254740992.0) * 100.0) - 50 for h in (h0, h1, h2))
        return ints, (h3 >> 11) * (1.0 / 9007199254740992.0)
    return fn


def assert_same_particles(gpu_p, ora_p, what=""):
    """Bit-exact comparison of two P_DATA_TYPE arrays, with a readable diff."""
    if gpu_p.tobytes() == ora_p.tobytes():
        return
    for f in gpu_p.dtype.names:
        a, b = gpu_p[f], ora_p[f]
        if a.dtype.kind == "f":
            a, b = a.view(np.uint32), b.view(np.uint32)
        bad = np.nonzero(a != b)[0]
        if len(bad):
            k = bad[0]
            raise AssertionError("%s field %s differs at %d slots, first slot %d: gpu %r oracle %r" %
                                 (what, f, len(bad), k, gpu_p[f][k], ora_p[f][k]))
    ga = np.frombuffer(gpu_p.tobytes(), np.uint8).reshape(-1, 72)
    oa = np.frombuffer(ora_p.tobytes(), np.uint8).reshape(-1, 72)
    rows, cols = np.nonzero(ga != oa)
    raise AssertionError("%s records differ only in padding bytes: %d slots, first slot %d byte %d gpu %d oracle %d, cell there %d" %
                         (what, len(set(rows.tolist())), rows[0], cols[0], ga[rows[0], cols[0]], oa[rows[0], cols[0]], gpu_p["cell"][rows[0]]))
