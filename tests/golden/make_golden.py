#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE's own code.

Runs only in the build container: it loads oracle/_ref/libref_l4.so, which
oracle/Makefile compiles from the reference's L4 helper sources where they lie
under /root/reference (common.h, app_common.cu, app.cu, unmodified).  Every
array stored is data -- inputs drawn here with numpy and the outputs the
reference functions returned for them -- never reference source text.

Outputs (all little-endian, field-wise; the 2 pad bytes of P_DATA_TYPE are never
stored):
  ref_constants.npz   struct layout + every macro of common.h as evaluated by the compiler
  ref_tables.npz      G4: get_cell_info / set_pkg_segments / fill_cells / rloc tables
  ref_setpos.npz      G5: set_pos_x / set_pos_i known answers incl. wrap cases
  ref_pairs.npz       bodyBodyInteraction / bodyBodyCollision known answers
  ref_accumulate.npz  serial fp32 accumulation over long neighbour lists (order matters)
  ref_queue.npz       a q_insert / q_remove script with every return value
  g2_cloud_n4096_seed12345.f32  (made by gen_cloud.cpp)
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import oracle_py as O  # noqa: E402

R = O.ref()
IP = C.POINTER(C.c_int)


def iptr(a):
    return a.ctypes.data_as(IP)


def blank_particles(n):
    p = np.zeros(n, dtype=O.P_DTYPE)
    p["cell"] = -1
    p["chunk"] = -1
    p["seg_type"] = -1
    p["seg_tid"] = -1
    return p


P_FIELDS = [n for n in O.P_DTYPE.names]


def split_fields(p):
    return {"p_" + n: np.ascontiguousarray(p[n]) for n in P_FIELDS}


def main():
    rng = np.random.default_rng(20261003)

    # ---- constants ------------------------------------------------------
    layout = np.zeros(24, np.int32)
    R.ref_struct_layout(iptr(layout))
    ints = np.zeros(32, np.int32)
    R.ref_int_constants(iptr(ints))
    reals = np.zeros(16, np.float64)
    R.ref_real_constants(reals.ctypes.data_as(C.POINTER(C.c_double)))
    np.savez_compressed(os.path.join(HERE, "ref_constants.npz"), layout=layout, ints=ints, reals=reals)
    ncells, nchunks, container = int(ints[5]), int(ints[6]), int(ints[27])
    seg_count = ints[15:19]

    # ---- G4 integer tables ---------------------------------------------
    cell_info = np.zeros((ncells, 3), np.int32)
    neib = np.full((ncells, 27), -1, np.int32)
    neib_n = np.zeros(ncells, np.int32)
    for c in range(ncells):
        R.ref_get_cell_info(c, iptr(cell_info[c]))
        neib_n[c] = R.ref_fill_cells(c, iptr(neib[c]))
    pkg = np.zeros((nchunks, 54), np.int32)
    for ch in range(nchunks):
        R.ref_set_pkg_segments(ch, iptr(pkg[ch]))
    cont_rloc, info_rloc = [], []
    for k, t in enumerate((1, 2, 4, 8)):
        for tid in range(int(seg_count[k])):
            cont_rloc.append((t, tid, R.ref_get_cont_rloc(t, tid)))
            info_rloc.append((t, tid, R.ref_get_info_rloc(t, tid)))
    # invalid segment types fall through both switches
    for t, tid in ((-1, -1), (0, 3), (3, 2)):
        cont_rloc.append((t, tid, R.ref_get_cont_rloc(t, tid)))
        info_rloc.append((t, tid, R.ref_get_info_rloc(t, tid)))
    ids = np.unique(np.concatenate([rng.integers(0, container, 4000),
                                    np.array([0, container - 1]),
                                    np.cumsum(ints[23:27])[:3] - 1, np.cumsum(ints[23:27])[:3]]))
    id_info = np.zeros((len(ids), 2), np.int32)
    for k, i in enumerate(ids):
        R.ref_get_id_info(int(i), iptr(id_info[k]))
    np.savez_compressed(os.path.join(HERE, "ref_tables.npz"), cell_info=cell_info, neib=neib,
                        neib_n=neib_n, pkg=pkg, cont_rloc=np.array(cont_rloc, np.int32),
                        info_rloc=np.array(info_rloc, np.int32), ids=ids.astype(np.int32),
                        id_info=id_info)

    # ---- G5 set_pos_x / set_pos_i --------------------------------------
    n = 3000
    pos = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
    # wrap cases: up to a bit more than one box length outside, and exact faces
    pos[:600] += rng.choice([-85, -45, -5, 5, 45, 85], (600, 3)).astype(np.float32)
    pos[600:640] = rng.choice([-40.0, 40.0, -35.0, 0.0, 5.0, 39.999996, -40.000004], (40, 3)).astype(np.float32)
    pos[640] = (41.0, -3.0, 0.1)  # SURVEY.md 8c G5 example
    p = blank_particles(n)
    # starting segment state: a third unset (-1,-1), a third "same as target", a third elsewhere
    start = blank_particles(n)
    for k in range(n):
        R.ref_set_pos_i(start[k:k + 1].ctypes.data, float(pos[k, 0]), float(pos[k, 1]), float(pos[k, 2]))
    mode = rng.integers(0, 3, n)
    p["seg_type"] = np.where(mode == 0, -1, np.where(mode == 1, start["seg_type"], 8))
    p["seg_tid"] = np.where(mode == 0, -1, np.where(mode == 1, start["seg_tid"], 5))
    before = p.copy()
    for k in range(n):
        R.ref_set_pos_x(p[k:k + 1].ctypes.data, float(pos[k, 0]), float(pos[k, 1]), float(pos[k, 2]))
    out = {"pos": pos, "in_seg_type": before["seg_type"], "in_seg_tid": before["seg_tid"]}
    out.update({"x_" + k: v for k, v in split_fields(p).items()})
    out.update({"i_" + k: v for k, v in split_fields(start).items()})
    np.savez_compressed(os.path.join(HERE, "ref_setpos.npz"), **out)

    # ---- pair kernels ---------------------------------------------------
    n = 6000
    bi = blank_particles(n)
    bj = np.zeros(n, dtype=O.T_DTYPE)
    bi["id"] = rng.integers(0, container, n)
    bj["id"] = np.where(rng.random(n) < 0.05, bi["id"], rng.integers(0, container, n))
    for f in ("x", "y", "z"):
        bi[f] = rng.uniform(-40, 40, n).astype(np.float32)
    sep = rng.choice([0.05, 0.3, 0.45, 2.0, 15.0], n)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    for k, f in enumerate(("x", "y", "z")):
        bj[f] = (bi[f] + (d[:, k] * sep * rng.uniform(0.5, 1.5, n))).astype(np.float32)
    ages = np.array([0.0, 1.4999999, 1.5, 1.5000001, 2.0, 7.5, 15.0, 15.000001, 20.0], np.float32)
    bi["age"] = rng.choice(ages, n)
    bj["age"] = rng.choice(ages, n)
    bi["w"] = 60.0
    bj["w"] = rng.choice([60.0, 1.0, 0.0, 123.5], n).astype(np.float32)
    # exact-radius cases for the collision compare (float 0.4 vs double 0.4)
    bj["x"][:50] = bi["x"][:50]
    bj["y"][:50] = bi["y"][:50]
    bj["z"][:50] = bi["z"][:50] + np.float32(0.4)
    acc_in = rng.normal(size=(n, 3)).astype(np.float32)
    acc_in[: n // 2] = 0
    acc = acc_in.copy()
    R.ref_body_body_interaction(n, bi.ctypes.data, bj.ctypes.data, acc.ctypes.data)
    flags = np.zeros(n, np.int32)
    R.ref_body_body_collision(n, bi.ctypes.data, bj.ctypes.data, flags.ctypes.data)
    np.savez_compressed(os.path.join(HERE, "ref_pairs.npz"),
                        bi_id=bi["id"], bi_age=bi["age"], bi_x=bi["x"], bi_y=bi["y"], bi_z=bi["z"],
                        bj_id=bj["id"], bj_age=bj["age"], bj_w=bj["w"],
                        bj_x=bj["x"], bj_y=bj["y"], bj_z=bj["z"],
                        acc_in=acc_in, acc_out=acc, flags=flags)

    # ---- serial accumulation over long lists ---------------------------
    m, npart = 6912, 8
    cases = {}
    for k in range(npart):
        me = blank_particles(1)
        me["id"] = 1000 + k
        me["age"] = 2.0
        ctr = rng.uniform(-30, 30, 3).astype(np.float32)
        me["x"], me["y"], me["z"] = ctr
        nb = np.zeros(m, dtype=O.T_DTYPE)
        nb["id"] = np.arange(m) + 5000
        nb["id"][m // 3] = 1000 + k  # self entry is skipped by id
        for a, f in enumerate(("x", "y", "z")):
            nb[f] = (ctr[a] + rng.uniform(-7.5, 7.5, m)).astype(np.float32)
        nb["w"] = 60.0
        nb["age"] = rng.choice([2.0, 3.0, 1.0], m, p=[0.6, 0.3, 0.1]).astype(np.float32)
        a3 = np.zeros(3, np.float32)
        R.ref_accumulate(me.ctypes.data, m, nb.ctypes.data, a3.ctypes.data)
        cases["me%d" % k] = np.array([me["id"][0], 0], np.int32)
        cases["mepos%d" % k] = np.array([me["x"][0], me["y"][0], me["z"][0], me["age"][0]], np.float32)
        cases["nbid%d" % k] = nb["id"].copy()
        cases["nb%d" % k] = np.stack([nb["x"], nb["y"], nb["z"], nb["w"], nb["age"]], 1)
        cases["acc%d" % k] = a3
    np.savez_compressed(os.path.join(HERE, "ref_accumulate.npz"), n=np.int32(npart), **cases)

    # ---- queue script ---------------------------------------------------
    # start state = q_start_fast (every slot free); replay a random script on a
    # few segments and record every q_remove result and the final records.
    s = O.System()
    qi = s.queue_info.copy()
    q = s.queue.copy()
    s.close()
    segs = [(1, 0), (1, 63), (2, 7), (4, 299), (8, 124), (8, 62)]
    script, results = [], []
    held = {sg: [] for sg in segs}
    for step in range(6000):
        sg = segs[rng.integers(0, len(segs))]
        burst = step % 1500 > 1200  # drain phases to hit the empty/wrap branches
        if held[sg] and (rng.random() < (0.2 if burst else 0.55)):
            x = held[sg].pop(rng.integers(0, len(held[sg])))
            R.ref_q_insert(qi.ctypes.data, q.ctypes.data, sg[0], sg[1], int(x))
            script.append((1, sg[0], sg[1], int(x)))
            results.append(0)
        else:
            r = R.ref_q_remove(qi.ctypes.data, q.ctypes.data, sg[0], sg[1])
            script.append((0, sg[0], sg[1], 0))
            results.append(r)
            if r >= 0:
                held[sg].append(r)
    # drain one small-ish segment completely to exercise underflow
    for _ in range(4200):
        r = R.ref_q_remove(qi.ctypes.data, q.ctypes.data, 8, 62)
        script.append((0, 8, 62, 0))
        results.append(r)
    R.ref_q_insert(qi.ctypes.data, q.ctypes.data, 8, 62, 777)
    script.append((1, 8, 62, 777))
    results.append(0)
    touched = sorted({R.ref_get_info_rloc(a, b) for a, b in segs})
    np.savez_compressed(os.path.join(HERE, "ref_queue.npz"), script=np.array(script, np.int32),
                        results=np.array(results, np.int32), touched=np.array(touched, np.int32),
                        final_info=np.stack([qi[n] for n in O.Q_DTYPE.names], 1)[touched],
                        final_queue_hash=np.array([int(np.bitwise_xor.reduce(
                            (q.astype(np.int64) + 1) * (np.arange(len(q), dtype=np.int64) * 2654435761 % (1 << 31))))]))
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
