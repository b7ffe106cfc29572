#!/usr/bin/env python3
"""bench.py -- particle-updates/s of the full reference step on MI355X.

One "step" = init_iframe -> build_grid -> calc_forces (pair loop, integrate, lifecycle)
over the whole cloud, state resident in HBM.  Workload = BASELINE.json configs[2]:
N = 2^20 uniform-random particles in the reference's default 16^3-cell box (256 per
cell, 27-cell cutoff => ~6.9k neighbours per particle), fp32, exact reference
arithmetic, every constant at the reference's shipped value (explosions off: their
RNG is non-deterministic in the reference).

The reference's physics does not hold N: the non-periodic stencil makes the cloud's
surface implode (a ~ 3000 in the outer cell layer) and collisions, cell overflow and
full segments remove half the particles within a few steps.  So each timed step is
one pass over the SAME batch: the N = 2^20 cloud is restored from a device-side
snapshot (an HBM-to-HBM copy, inside the timed region) and advanced by one step --
exactly the work the CPU baseline samples.  The line's `evolve` key reports a
free-running stretch beside it (--evolve makes that the timed loop).

    python bench.py --gpus N --steps K --warmup W

N > 1 starts one process per GPU (python -m torch.distributed.run; or is started that way by
the caller) and runs the SLAB-PARTITIONED path: every rank holds only the segments of its
cell layers -- slots, particles, free-slot queues -- and exchanges, with its two ring
neighbours over RCCL send/recv, the snapshot of its boundary layers, the force records of
lent layers and the particles that change owner.  The ranks that do the work are C++ programs
(host/ps_ring_rccl: libpsamd.so's stage calls, RCCL on a second HIP stream): each Python rank starts its own as a child process before anything here touches
a GPU, and rank 0 relays its record as the one JSON line.  --backend nccl / gloo runs the same
step from Python over torch.distributed instead (particlesystem_amd/slab.py), which is also
what the ranks fall back to, together, if a C++ rank fails.  Total work is fixed at N = 2^20
as the GPU count grows => "scaling": "strong".
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector (spec)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FLOP_PER_PAIR = 20           # the usual N-body convention
APPLY_BYTES_PER_UPDATE = 64  # read pos4+vel4, write pos4+vel4
TRAFFIC_FILE = os.path.join("profiles", "r5_traffic.json")
XGMI_LINK_GBS = 153.0        # MI355X_MICROARCH.md: one xGMI link, one direction (model only)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=0, help="particles (default 2^20; 2^18 with --all-pairs)")
    ap.add_argument("--fast-math", action="store_true", help="FMA/rsq pair arithmetic (not bit-exact)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=64, help="host threads for the many-thread CPU figure (0: every CPU this process may run on; "
                    "on the one-GPU boxes of this pool the job's CPU share is far below the 256 hardware threads it may be scheduled on: "
                    "256 threads measured 3.6e5 updates/s, 64 threads 7.5e5)")
    ap.add_argument("--all-pairs", action="store_true", help="BASELINE configs[1]: every particle against EVERY body (PSAMD_FLAG_ALL_PAIRS; "
                    "not in the reference, parity unpinned), default N = 2^18; roofline = 20 N^2 flop per step")
    ap.add_argument("--no-side-runs", action="store_true", help="skip the lifecycle-off / tolerance-mode / free-running runs beside the headline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--kernel-times", action="store_true", help="HIP events between all stage kernels, not only around the "
                    "pair pass / apply / life cycle (costs ~40 us of idle GPU per step)")
    ap.add_argument("--timing-period", type=int, default=8, help="record the HIP events that time the force pass / apply / life cycle on every "
                    "n-th step of the timed region (an event between two kernels idles the GPU for ~6 us; 1: every step)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--chunk-factor", type=int, default=4, help="grid = (chunk_factor*chunk_dim)^3 cells (reference: 4)")
    ap.add_argument("--chunk-dim", type=int, default=4)
    ap.add_argument("--evolve", action="store_true", help="free-running steps instead of one pass per step over the same cloud")
    ap.add_argument("--evolve-steps", type=int, default=10, help="length of the free-running stretch reported beside the headline")
    ap.add_argument("--halo-cap-cell", type=int, default=0, help="bodies per cell a halo message has room for (0: the cell capacity)")
    ap.add_argument("--settle-seconds", type=float, default=0.5, help="untimed steps before the warmup until the clocks have settled")
    ap.add_argument("--sim-world", type=int, default=0, help="projection on ONE GPU: all ranks of an N-rank slab run live in this "
                    "process and run one after the other; reports every rank's stage times and the modelled step")
    ap.add_argument("--launch-check", action="store_true", help="only start the ranks, form the process group, reduce one number and "
                    "print the line's skeleton (no GPU work): proves the --gpus N launcher on a machine without GPUs")
    ap.add_argument("--overlap-interior", action="store_true", help="multi-GPU: a pass of its own for the cells that need no halo, "
                    "run while the halo travels (measured slower than the single pass it splits; see DESIGN.md)")
    ap.add_argument("--backend", default="ring", help="--gpus > 1: ring = one C++ rank per GPU (host/ps_ring_rccl: RCCL on a second HIP stream, "
                    "stage sequences as hipGraphs; the default); nccl = the same step from Python over torch.distributed (RCCL); gloo: "
                    "messages staged through host memory, for rehearsals on one GPU")
    ap.add_argument("--host", default="ring", choices=("ring", "python"), help="who drives the timed loop: ring = the C++ host (host/ps_ring_rccl on include/psamd.h), "
                    "for EVERY --gpus N including 1 -- one host for the whole scaling curve (the default); python = this process through ctypes "
                    "(one GPU; what the side runs beside the headline use)")
    ap.add_argument("--graphs", action="store_true", help="one GPU / --sim-world / torch ranks: stage sequences as hipGraphs (psamd_set_graphs)")
    ap.add_argument("--wait-policy", type=int, default=-1, help="how the host waits for a step's scalars: 0 spin, 1 short spin then naps (default: the library's)")
    ap.add_argument("--ring-graphs", type=int, default=0, help="C++ ranks: stage sequences as hipGraphs (default 0, plain launches: a graph launch costs "
                    "~10 us on the GPU's timeline, four of them a step -- profiles/r4_ab_graphs.txt)")
    ap.add_argument("--ring-side-stream", type=int, default=0, help="C++ ranks: 0 = every RCCL call on the compute stream (default: a dependency that crosses "
                    "streams costs ~15 us each way on the GPU's timeline, measured); 1 = the status gather, and with --overlap-interior the halo, on a second HIP stream; "
                    "2 = every transfer on the second stream (round 4's form)")
    ap.add_argument("--ring-timeout", type=float, default=0.0, help="seconds a C++ rank may take before it is ended and the ranks fall back (0: from --steps)")
    ap.add_argument("--sustained-steps", type=int, default=200, help="one GPU: when --steps is shorter than this, a second timed region of this "
                    "many steps is reported beside the line's figure (which clock a figure was taken at is part of the figure); 0: none")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N without a launcher: start the N ranks as fresh child processes (nothing here has
    touched the GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, as
    torch.distributed.run would set them), relay rank 0's JSON line, exit with the job's status."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PSAMD_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is read by a thread; meanwhile watch the ranks: one that dies would leave
    # the others waiting in a collective for ever, so end them (exact PIDs) and report its status
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        time.sleep(0.2)
        failed = next((p.returncode for p in procs if p.poll() not in (None, 0)), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    codes = [p.wait() for p in procs]
    reader.join(timeout=10)
    out = "".join(chunks)
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    if lines:
        print(lines[-1])
    else:
        sys.stderr.write(out)
    rc = abs(failed) if failed is not None else max((abs(c) for c in codes), default=0)
    sys.exit(rc if rc else (0 if lines else 1))


def make_inputs(sysobj, n, seed):
    xyz = sysobj.uniform_cloud(n, seed)
    rng = np.random.default_rng(seed)
    life = 300.0 * sysobj.cfg.dt
    age = rng.uniform(life / 7.0, life / 2.0, n).astype(np.float32)   # [MIN_ADULT_AGE, MAX_ADULT_AGE)
    fert = np.full(n, 1e6, np.float32)
    return xyz, age, fert


def measured_traffic(kernel_prefix, args, world):
    """HBM bytes per launch from the committed rocprofv3 PMC capture -- only for the very
    configuration it was captured with (bench.py defaults on one GPU); else None."""
    if world != 1 or args.n != (1 << 20) or args.chunk_factor != 4 or args.chunk_dim != 4 or args.evolve or args.fast_math \
            or args.sim_world or args.all_pairs:
        return None
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            kernels = json.load(f)["kernels"]
        for name, rec in kernels.items():
            if name.startswith(kernel_prefix):
                return float(rec["hbm_bytes_per_launch_corrected"])
    except Exception:
        pass
    return None


def pair_count(cellgrid_counts, force_counts, G):
    """Force terms one pair pass evaluates: sum_c f_c * sum_{c' in stencil(c)} n_c', with n the
    particles per cell and f those among them the pass computes a force for (all of them in the
    one-pass modes; in the two-pass mode the ones the reference's force loop runs for)."""
    c = cellgrid_counts.astype(np.int64).reshape(G, G, G)
    f = force_counts.astype(np.int64).reshape(G, G, G)
    p = np.pad(c, 1)
    nb = np.zeros_like(c)
    for a in range(3):
        for b in range(3):
            for d in range(3):
                nb += p[a:a + G, b:b + G, d:d + G]
    return int((f * nb).sum())


def cpu_baseline(args, xyz, age, fert, cfg_over):
    """The oracle (CPU port of the reference `_host` path) on bounded samples of the SAME
    workload.  `cpu_baseline`: one thread, whole chunks of calc_forces until ~args.cpu_seconds
    have passed.  `cpu_baseline_many_threads`: the read-only pair pass (collision scan + force
    loop, >99.9 % of the CPU step) on up to --cpu-threads host threads, contiguous shares per thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    o = O.System(O.default_config(**cfg_over))
    o.fill(xyz, age=age, fert_age=fert)
    o.init_iframe()
    o.build_grid()
    many = None
    cores = len(os.sched_getaffinity(0)) if args.cpu_threads <= 0 else min(len(os.sched_getaffinity(0)), args.cpu_threads)
    if cores > 1:
        total = o.sorted_count()
        f = np.zeros((total + 8, 4), np.float32)
        probe = min(total, cores * 512)
        t0 = time.perf_counter()
        o.calc_pairs_threads(0, probe, f, cores)
        rate = probe / max(time.perf_counter() - t0, 1e-6)
        share = int(min(total, max(probe, rate * args.cpu_seconds * 0.6)))
        segs = 8 if share < total else 1          # spread over the box: edge cells are cheaper than interior ones
        seg = share // segs
        t0 = time.perf_counter()
        for k in range(segs):
            lo = k * total // segs
            o.calc_pairs_threads(lo, lo + seg, f, cores)
        tm = time.perf_counter() - t0
        many = {"value": seg * segs / tm, "unit": "particle-updates/s", "cores": cores, "kind": "port",
                "sample": "pair pass (collision scan + force loop) of %d particles (%d runs of cells spread over the "
                          "box) of the same N=%d cloud on %d host threads, %.1f s" % (seg * segs, segs, len(xyz), cores, tm),
                "host_cpus": os.cpu_count()}
        del f
    gm = int(o.gridmax[0])
    counts = o.chunkgrid[:, 0].copy()
    done, t = 0, 0.0
    chunks = 0
    t0 = time.perf_counter()
    while chunks < o.d.num_chunks and t < args.cpu_seconds:
        o.calc_forces_chunk(chunks, gm)
        done += int(counts[chunks])
        chunks += 1
        t = time.perf_counter() - t0
    o.close()
    one = {"value": done / t if t > 0 else 0.0, "unit": "particle-updates/s", "cores": 1, "kind": "port",
           "sample": "calc_forces of chunks 0..%d of the same N=%d cloud (%d particles, %.1f s), "
                     "1 thread; grid build excluded (0.07%% of the CPU step)" % (chunks - 1, len(xyz), done, t),
           "host_cpus": os.cpu_count()}
    return one, many


def pass_counts(g, restore):
    """(particles per cell, particles per cell the force pass visits) of one frame of context g."""
    if restore:
        g.snapshot_restore()
    g.init_iframe(); g.build_grid()
    n = g.download_cellgrid()[:, 0].copy()
    g.calc_forces_pairs()
    f = g.download_force_counts()
    g.calc_forces_apply()
    return n, f


def force_terms(n, f, G, all_pairs):
    """pair evaluations of one force pass: per visited particle its stencil's population, or -- all-pairs -- every listed body"""
    if all_pairs:
        return int(f.astype(np.int64).sum()) * int(n.astype(np.int64).sum())
    return pair_count(n, f, G)


def side_run(ps, cfg_over, device, xyz, age, fert, flags, steps, restore, warm=30, all_pairs=False, timing_period=1, warm_ctx=None, **cfg_extra):
    """A second, short measurement on a fresh context: the same timed loop with other flags (fast
    math), other constants (life cycle off) or free-running (restore=False).  Never the headline.
    Returns updates, seconds, live counts per step, and -- from HIP events on the context's stream
    inside the timed region -- the force pass's own roofline entry."""
    cfg = dict(cfg_over)
    cfg.update(cfg_extra)
    g = ps.ParticleSystem(ps.default_config(device=device, flags=flags, **cfg))
    g.set_tdata_mirror(False)           # (as the headline's context: nobody fetches T_DATA here)
    g.fill_particles(xyz, age=age, fert_age=fert)
    g.snapshot_save()
    live = []
    G = g.sizes.grid_dim
    # (the census before the warmup, so that the warmup runs straight into the timed region: a GPU left idle for the
    # downloads costs the first timed steps their clock)
    n0, f0 = pass_counts(g, restore) if restore else (None, None)
    g.set_timing(True, period=timing_period)
    g.set_timing(False)
    if restore:
        for _ in range(warm):
            g.snapshot_restore(); g.step(1)
    elif warm_ctx is not None:
        # a free-running stretch has no warmup of its own (its steps change its state): another context keeps the GPU
        # busy until the moment it starts, so that its few steps do not run on a chip that has just sat idle
        for _ in range(warm):
            warm_ctx.snapshot_restore(); warm_ctx.step(1)
        warm_ctx.synchronize()
    g.set_timing(True, period=timing_period)
    g.synchronize()
    p0 = g.counters["particles_processed"]
    t0 = time.perf_counter()
    for _ in range(steps):
        if restore:
            g.snapshot_restore()
        g.step(1)
        if not restore:
            live.append(int(g.device_view().live))      # host copy, no sync: the library reads a step's scalars a step late
    g.synchronize()
    dt = time.perf_counter() - t0
    if not restore:
        live = live[1:] + [int(g.device_view().live)]   # ... so entry k was step k - 1's: shift, the last step's is in now
    tim, launches = g.timing()
    med, mx, _ = g.timing_stats()
    g.set_timing(False)
    done = g.counters["particles_processed"] - p0
    roof = None
    if restore:
        pairs = force_terms(n0, f0, G, all_pairs)
        us = tim["pairs"] / max(launches, 1)
        ach = pairs * FLOP_PER_PAIR / (us * 1e-6) / 1e12 if us > 0 else 0.0
        roof = {"kernel": "force pass (HIP events on the context's stream, inside this run's timed region)", "bound": "valu",
                "achieved": ach, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / VALU_PEAK_TFLOPS, "traffic": None,
                "pairs_per_launch": float(pairs), "flop_per_pair": FLOP_PER_PAIR, "us_per_launch": us,
                "particles_with_a_force_term": int(f0.sum())}
    g.close()
    return done, dt, live, roof, {"median": {k: v for k, v in med.items() if v > 0}, "max": {k: v for k, v in mx.items() if v > 0},
                                  "mean": {k: v / max(launches, 1) for k, v in tim.items() if v > 0}, "timed_steps": launches}


def side_runs(out, args, ps, cfg_over, device, xyz, age, fert, flags, period, warm_ctx):
    """The figures beside the headline, each from a short run of its own on a fresh context (the Python host): SURVEY 8(d)'s
    lifecycle-off mode, the tolerance mode, a free-running stretch."""
    # lifecycle off: collision radius 0 (and no births, no deaths of age in these steps), so every one of the N particles
    # goes through the force loop and is integrated
    d, t, _, roof, kt = side_run(ps, cfg_over, device, xyz, age, fert, flags, 50, True, timing_period=min(5, period), collision_radius=0.0)
    out["lifecycle_off"] = {
        "what": "the same cloud and step with COLLISION_RADIUS = 0: nothing collides, all %d particles get a force and are "
                "integrated (the headline's step, at the reference's radius 0.4, integrates the particles the "
                "reference integrates: config.particles_with_a_force_term)" % args.n,
        "value": d / t, "unit": "particle-updates/s", "ms_per_step": 1e3 * t / 50, "steps": 50,
        "roofline": roof, "kernel_us_per_step": kt}
    d, t, _, roof, kt = side_run(ps, cfg_over, device, xyz, age, fert, ps.FLAG_FAST_MATH, 50, True, timing_period=min(5, period))
    out["within_tolerance_mode"] = {
        "arithmetic": "PSAMD_FLAG_FAST_MATH (FMA + v_rsq): accelerations deviate from the oracle's by the amounts "
                      "tests/test_gpu_fast.py measures and bounds (also at this density); not the headline",
        "value": d / t, "unit": "particle-updates/s", "ms_per_step": 1e3 * t / 50, "steps": 50,
        "roofline": roof, "kernel_us_per_step": kt}
    d, t, lv, _, _ = side_run(ps, cfg_over, device, xyz, age, fert, flags, args.evolve_steps, False, warm_ctx=warm_ctx)
    out["evolve"] = {"what": "%d free-running steps from the same cloud (the population collapses: surface implosion, "
                             "collisions), exact arithmetic" % args.evolve_steps,
                     "value": d / t, "unit": "particle-updates/s", "ms_per_step": 1e3 * t / args.evolve_steps,
                     "live_per_step": lv}


def sim_world(args, ps, cfg_over, flags):
    """All ranks of a world-`W` slab run in this process, on this one GPU, one after the other
    on one stream: every rank's stage then takes what it would take with a GPU to itself.
    Messages are copied device to device; their transfer is modelled from their sizes."""
    import torch
    from particlesystem_amd.slab import DeviceRing, routes
    W = args.sim_world
    stream = torch.cuda.Stream()
    ranks = [ps.ParticleSystem(ps.default_config(device=0, rank=r, world=W, flags=flags, halo_cap_cell=args.halo_cap_cell, **cfg_over))
             for r in range(W)]
    xyz, age, fert = make_inputs(ranks[0], args.n, args.seed)
    for g in ranks:
        g.set_tdata_mirror(False)
        g.fill_particles(xyz, age=age, fert_age=fert)
        g.snapshot_save()
        if args.graphs:
            g.set_graphs(True)
        if args.wait_policy >= 0:
            g.set_wait_policy(args.wait_policy)
    rings = [DeviceRing(g, None, r, W, stream) for r, g in enumerate(ranks)]
    stages = ("build", "pairs_interior", "pairs", "apply", "finish")
    ev = [[[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in stages] for _ in range(W)]
    tot = np.zeros((W, len(stages)))
    per_step = []                      # [step][rank][stage] in ms

    def deliver(phase):
        for r in range(W):
            for ph, out_slot, peer, in_slot in routes(r, W):
                if ph == phase and out_slot in rings[r].t:
                    n = ranks[r].msg_bytes(out_slot) // 4          # (a transfer message: what travels now, a prefix of its buffer)
                    rings[peer].t[in_slot][:n].copy_(rings[r].t[out_slot][:n], non_blocking=True)
        if phase == "halo":                  # the all-gathers: status records, and -- all-pairs forces -- the snapshot blocks
            for out_slot, in_slot in ((10, 11), (12, 13)):
                if out_slot not in rings[0].t:
                    continue
                n = rings[0].t[out_slot].numel()
                for r in range(W):
                    for q in range(W):
                        rings[q].t[in_slot][r * n:(r + 1) * n].copy_(rings[r].t[out_slot], non_blocking=True)

    def one_step(timed):
        with torch.cuda.stream(stream):
            for g in ranks:
                if not args.evolve:
                    g.snapshot_restore()
            for k, (name, phase) in enumerate(zip(stages, (None, "halo", "force", "xfer", None))):
                for r, g in enumerate(ranks):
                    ev[r][k][0].record(stream)
                    if name != "pairs_interior" or args.overlap_interior:
                        getattr(g, "slab_" + name)()
                    ev[r][k][1].record(stream)
                if phase:
                    deliver(phase)
        if timed:
            torch.cuda.synchronize()
            per_step.append([[ev[r][k][0].elapsed_time(ev[r][k][1]) for k in range(len(stages))] for r in range(W)])

    for _ in range(args.warmup):
        one_step(False)
    p0 = [g.counters["particles_processed"] for g in ranks]
    for _ in range(args.steps):
        one_step(True)
    # A rank's stage is timed from the moment the GPU reaches it to the moment it leaves it: a hiccup of the ONE host
    # thread that feeds all eight ranks here (a nap that overran while it waited for a rank's scalars) lands in whichever
    # rank's interval it falls -- the median over the timed steps leaves those out, the mean is reported beside it.
    mean = np.mean(np.array(per_step), axis=0)
    tot = np.median(np.array(per_step), axis=0)
    updates = sum(g.counters["particles_processed"] - a for g, a in zip(ranks, p0)) / args.steps
    msg = {}
    for r, g in enumerate(ranks):
        # the bigger of what the rank sends and receives per phase (both directions run at once)
        msg[r] = {"halo": max(g.msg_bytes(1), g.msg_bytes(2), g.msg_bytes(0), g.msg_bytes(3)),
                  "force": max(g.msg_bytes(4), g.msg_bytes(5)), "xfer": g.msg_bytes(6) + g.msg_bytes(7),
                  "allgather": g.msg_bytes(12)}
    # model: a message costs 10 us + bytes / one xGMI link, one direction.  The halo travels on
    # RCCL's stream while the rank works on its interior cells (DeviceRing.step): only what
    # outlasts that pass is on the critical path; force and xfer are exposed once each.  The
    # all-pairs snapshot all-gather is priced as a ring: W - 1 hops of one block.
    def t_ms(b):
        return 1e-2 + 1e3 * b / (XGMI_LINK_GBS * 1e9) if b else 0.0
    k_int = stages.index("pairs_interior")
    phase_ms = {"halo": [max(0.0, max(t_ms(msg[r]["halo"]), (W - 1) * t_ms(msg[r]["allgather"])) - float(tot[r, k_int])) for r in range(W)],
                "force": [t_ms(msg[r]["force"]) for r in range(W)], "xfer": [t_ms(msg[r]["xfer"] / 2) for r in range(W)]}
    comm_ms = [phase_ms["halo"][r] + phase_ms["force"][r] + phase_ms["xfer"][r] for r in range(W)]
    per_rank = tot.sum(1)
    # Two bounds of the step.  Optimistic: the slowest rank's own stages plus its own transfers, as if
    # no rank ever waited for a neighbour.  Coupled: every stage waits for the slowest rank's previous
    # stage (the halo needs the neighbours' build, force_in their pairs, xfer their apply): the sum
    # over stages of the per-stage maximum, plus the largest transfer of every phase.  A real ring lies
    # between the two; the coupled figure is the one quoted.
    step_lo = float((per_rank + np.array(comm_ms)).max())
    step_hi = float(tot.max(0).sum() + sum(max(v) for v in phase_ms.values()))
    out = {"sim_world": W, "n": args.n, "stage_ms_per_rank": {name: [round(float(x), 4) for x in tot[:, k]] for k, name in enumerate(stages)},
           "stage_ms_per_rank_is": "median over the timed steps", "stage_ms_per_rank_mean": {name: [round(float(x), 4) for x in mean[:, k]] for k, name in enumerate(stages)},
           "compute_ms_per_rank": [round(float(x), 4) for x in per_rank], "modelled_comm_ms_per_rank": [round(x, 4) for x in comm_ms],
           "message_bytes_rank1": msg[min(1, W - 1)], "halo_cap_cell": int(args.halo_cap_cell),
           "modelled_step_ms": step_hi, "modelled_step_ms_optimistic": step_lo,
           "modelled_updates_per_s": updates / (step_hi * 1e-3), "updates_per_step": updates, "timed_steps": args.steps,
           "all_pairs": bool(args.all_pairs), "max_ops_one_queue_per_rank": [int(g.counters["max_ops_one_queue"]) for g in ranks],
           "note": "one GPU runs the ranks one after the other; compute times are measured (HIP events); a transfer is modelled "
                   "as 10 us + bytes / %.0f GB/s (halo up, force, the two xfer messages in parallel; the all-pairs snapshot "
                   "all-gather as a ring of W - 1 such hops); the halo overlaps the interior pass, the rest is not overlapped.  "
                   "modelled_step_ms = sum over stages of the slowest rank's stage + the largest transfer of every phase "
                   "(neighbour-coupled: an upper bound); _optimistic = the slowest rank's own stages and transfers (a lower bound).  "
                   "Left out: host launch overhead." % XGMI_LINK_GBS}
    for g in ranks:
        g.close()
    return out

class ClockWatch:
    """Samples the GPU's shader clock (sysfs pp_dpm_sclk, the level marked current) while a timed region runs: the chip is
    power-bound under this load, and which clock a figure was taken at is part of the figure.  The sampler is a CHILD
    PROCESS (a few lines of Python that never touch the GPU): a thread of this process would share the interpreter lock
    with the timed loop -- round 4's driver run lost 0.6 ms on three timed steps to exactly that."""

    SAMPLER = (
        "import sys,time\n"
        "path,period=sys.argv[1],float(sys.argv[2])\n"
        "sys.stdout.write('ready\\n'); sys.stdout.flush()\n"
        "sys.stdin.readline()\n"                           # 'go'
        "import select\n"
        "while True:\n"
        "    try:\n"
        "        for line in open(path):\n"
        "            if line.strip().endswith('*'):\n"
        "                sys.stdout.write(''.join(ch for ch in line.split(':')[1] if ch.isdigit())+'\\n')\n"
        "    except Exception: pass\n"
        "    if select.select([sys.stdin],[],[],period)[0]: break\n"
        "sys.stdout.flush()\n")

    def __init__(self, index=0, period=0.02):
        import glob
        self.path, self.proc, self.samples, self.period = None, None, [], period
        try:
            # the card whose PCI address is the HIP device's (a box may show the host's other GPUs in sysfs)
            import torch
            pr = torch.cuda.get_device_properties(index)
            want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            for f in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
                if os.path.basename(os.path.realpath(os.path.dirname(f))) == want:
                    self.path = f
        except Exception:
            pass
        if self.path:
            # started here, ahead of the warmup (an interpreter takes tens of milliseconds to come up); it samples
            # between 'go' and the next line on its stdin
            try:
                self.proc = subprocess.Popen([sys.executable, "-c", self.SAMPLER, self.path, str(period)], stdin=subprocess.PIPE,
                                             stdout=subprocess.PIPE, text=True)
                self.proc.stdout.readline()
            except Exception:
                self.proc = None

    def __enter__(self):
        if self.proc:
            try:
                self.proc.stdin.write("go\n"); self.proc.stdin.flush()
            except Exception:
                self.proc = None
        return self

    def __exit__(self, *a):
        if self.proc:
            try:
                out, _ = self.proc.communicate("stop\n", timeout=5.0)
                self.samples = [int(x) for x in out.split() if x.isdigit()]
            except Exception:
                self.proc.kill()
            self.proc = None

    def summary(self):
        if not self.samples:
            return None
        v = sorted(self.samples)
        return {"min": v[0], "median": v[len(v) // 2], "max": v[-1], "samples": len(v), "source": self.path}


def ring_paths():
    """the id file of this job's C++ ranks and the job's nonce: the same on every rank of one launch (the ranks share a
    parent -- torch.distributed.run's agent, or bench.py's own launcher -- and MASTER_PORT), different from any other's"""
    port = int(os.environ.get("MASTER_PORT", "0") or 0)
    job = (port % 100000) * 10000000 + os.getppid() % 10000000
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "psamd_ring_%d" % job), job


def ring_args(args, world, rank, local_rank, cfg_over):
    exe = os.path.join(ROOT, "host", "ps_ring_rccl")
    id_file, job = ring_paths()
    cmd = [exe, "--world", str(world), "--rank", str(rank), "--device", str(local_rank), "--id-file", id_file, "--job", str(job)]
    if os.environ.get("PSAMD_BENCH_RING_BREAK"):
        cmd.append("--no-such-option")          # (test hook: every C++ rank fails at once; the ranks must fall back together)
    if args.launch_check:
        return cmd + ["--launch-check", "--steps", str(args.steps), "--warmup", str(args.warmup)]
    cmd += ["--bench", "--n", str(args.n), "--seed", str(args.seed), "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--settle-seconds", str(args.settle_seconds), "--timing-period", str(args.timing_period),
            "--chunk-factor", str(args.chunk_factor), "--chunk-dim", str(args.chunk_dim),
            "--halo-cap-cell", str(args.halo_cap_cell), "--xfer-cap", str(cfg_over.get("xfer_cap", 0)),
            "--max-particles", str(cfg_over["max_particles_num"]),
            "--graphs", str(args.ring_graphs), "--side-stream", str(args.ring_side_stream),
            "--sustained-steps", str(args.sustained_steps if (world == 1 and not args.evolve) else 0)]
    for flag, on in (("--all-pairs", args.all_pairs), ("--fast-math", args.fast_math), ("--evolve", args.evolve),
                     ("--overlap-interior", args.overlap_interior)):
        if on:
            cmd.append(flag)
    return cmd


def run_ring_rank(args, world, rank, local_rank, cfg_over):
    """Start this rank's C++ program as a child process (nothing in this process has touched a GPU), wait for it --
    not for ever: a rank that hangs is ended by its exact PID, and so is this rank's program the moment ANOTHER rank
    reports a failure (its peers would otherwise sit in RCCL until the limit) --, tell the other ranks how it went
    through a file beside the id file, and hear from them.
    Returns (every rank succeeded, rank 0's record or None, what went wrong, whether a program was ended by a signal or the limit)."""
    id_file, _ = ring_paths()
    mine = "%s.rc%d" % (id_file, rank)
    if os.path.exists(mine):
        os.remove(mine)
    limit = args.ring_timeout or (240.0 + 0.05 * (args.steps + args.warmup + (args.sustained_steps if world == 1 else 0)) * max(1, args.n >> 20) * (40 if args.all_pairs else 1))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    rec, err, violent = None, "", False

    def peer_codes():
        codes = {}
        for r in range(world):
            try:
                with open("%s.rc%d" % (id_file, r)) as f:
                    codes[r] = int(f.read().strip() or "1")
            except (OSError, ValueError):
                pass
        return codes

    try:
        p = subprocess.Popen(ring_args(args, world, rank, local_rank, cfg_over), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        import threading
        got = {}
        t = threading.Thread(target=lambda: got.update(zip(("out", "err"), p.communicate())), daemon=True)
        t.start()
        t_end = time.time() + limit
        why = ""
        while t.is_alive():
            t.join(timeout=0.25)
            if not t.is_alive():
                break
            if time.time() > t_end:
                why = "ended after %.0f s" % limit
            elif any(v != 0 for r, v in peer_codes().items() if r != rank):
                why = "ended because another rank failed"
            if why:
                p.kill()                                # this child, by its PID
                t.join(timeout=30)
                break
        out, errtxt = got.get("out", "") or "", got.get("err", "") or ""
        rc = p.returncode if p.returncode is not None else -9
        if why:
            rc, errtxt = (rc or -9), errtxt + "\n(" + why + ")"
            violent = violent or why.startswith("ended after")
        if rc < 0 and not why:
            violent = True                              # died of a signal: a fault, an abort
        lines = [l for l in out.splitlines() if l.startswith("{") and '"psamd_ring"' in l]     # (librccl prints its banner on stdout too)
        if rc == 0 and (lines or rank != 0):
            rec = json.loads(lines[-1]) if lines else None
        else:
            err = "rank %d: exit %s: %s" % (rank, rc, (errtxt or out)[-400:].strip())
            rc = rc or 1
    except OSError as e:
        rc, err = 127, "rank %d: %s" % (rank, e)
    with open(mine + ".tmp", "w") as f:
        f.write("%d\n" % rc)
    os.replace(mine + ".tmp", mine)
    # how did the others do?  (they write their file when their child has ended: within the same limit)
    codes = {}
    t_end = time.time() + limit + 30.0
    while len(codes) < world and time.time() < t_end:
        codes = peer_codes()
        if len(codes) < world:
            time.sleep(0.05)
    ok = len(codes) == world and all(v == 0 for v in codes.values())
    if any(v < 0 for v in codes.values()):
        violent = True
    if not ok and not err:
        err = "ranks %s failed or never reported" % sorted(set(range(world)) - {r for r, v in codes.items() if v == 0})
    if rank == 0:
        time.sleep(1.0)                      # (every rank has read the files by now, or will not)
        for r in range(world):
            try:
                os.remove("%s.rc%d" % (id_file, r))
            except OSError:
                pass
    return ok, rec, err, violent


def line_from_ring_record(args, r):
    """the driver's JSON line from the record rank 0's C++ program printed"""
    world, G, steps = r["world"], r["grid_dim"], r["steps"]
    elapsed = r["elapsed_s"]
    kt = r["kernel_us"]
    us_pairs, us_apply = kt.get("pairs", 0.0), kt.get("apply", 0.0)
    ach_tflops = r["pairs_rank0"] * FLOP_PER_PAIR / (us_pairs * 1e-6) / 1e12 if us_pairs > 0 else 0.0
    ach_gbs = r["own_updates"] / steps * APPLY_BYTES_PER_UPDATE / (us_apply * 1e-6) / 1e9 if us_apply > 0 else 0.0
    host = ("C++ ranks (host/ps_ring_rccl on include/psamd.h): RCCL send/recv/all-gather %s, %s" %
            ({0: "on the compute stream, between the stage kernels", 1: "on the compute stream, the status gather (and an overlapped halo) on a second HIP stream",
              2: "on a second HIP stream, ordered against the stage kernels by events"}[int(r.get("side_stream_mode", 2 if r["side_stream"] else 0))],
             "every stage's kernels as one hipGraph (%d replays, %d captures on rank 0)" % (r["graph_replays"], r["graph_captures"])
             if r["graphs"] else "plain kernel launches"))
    out = {
        "metric": "particle-updates/sec at N=2^20" if not args.all_pairs else "particle-updates/sec, all-pairs forces (BASELINE configs[1]), N=%d" % args.n,
        "value": r["updates"] / elapsed, "unit": "particle-updates/s",
        "n_gpus": world, "steps": steps, "warmup": r["warmup"],
        "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: N=%d uniform cloud, %d^3 cells x 5.0, ALL-PAIRS gravity (every particle against every "
                                "body: the 27-cell stencil in the reference's order, then every other cell; not in the reference, parity "
                                "unpinned) + reference collisions/integrate/wrap/relocation, %s" if args.all_pairs else
                                "BASELINE configs[2]: N=%d uniform cloud, %d^3 cells x 5.0, 27-cell cutoff gravity "
                                "+ reference collisions/integrate/wrap/relocation, all constants at reference defaults, "
                                "%s") % (args.n, G, "free-running" if args.evolve else "each step = one pass over the same cloud (restored in HBM)"),
                   "arithmetic": "fast-math" if args.fast_math else "reference-exact fp32 (bitwise parity mode)",
                   "parallelism": "%d slabs of cell layers, one per GPU (state partitioned by segment); halo / force / transfer messages between ring "
                                  "neighbours, the status records%s all-gathered: RCCL over xGMI" % (world, " and the snapshot blocks" if args.all_pairs else ""),
                   "host": host,
                   "updates_in_timed_region": r["updates"], "live_after": r["live_after"], "settle_steps_before_warmup": r["settle_steps"],
                   "particles_with_a_force_term": r["particles_with_a_force_term"],
                   "relocations": r["relocations"], "relocations_lost": r["relocations_lost"], "cell_overflow_kills": r["cell_overflow_kills"],
                   "halo_cap_cell": r["halo_cap_cell"], "message_bytes_rank0": r["message_bytes_rank0"], "rccl_mb_rank0": r["rccl_mb_rank0"]},
        "roofline": {"kernel": "k_pairs_balanced of rank 0 (its own share of the pairs against its own launch time)" if not args.all_pairs else
                               "k_pairs (all-pairs walk: stencil in the reference's order, then every other cell, summed per cell) of rank 0",
                     "bound": "valu", "achieved": ach_tflops, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tflops / VALU_PEAK_TFLOPS,
                     "traffic": None, "traffic_source": None, "pairs_per_launch": r["pairs_rank0"], "flop_per_pair": FLOP_PER_PAIR, "us_per_launch": us_pairs},
        "roofline_streaming": {"kernel": "k_apply of rank 0", "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach_gbs / HBM_PEAK_GBS, "traffic": None, "traffic_source": None,
                               "bytes_per_update": APPLY_BYTES_PER_UPDATE, "us_per_launch": us_apply},
        "shader_clock_mhz": r.get("shader_clock_mhz"),
        "sustained": ({"steps": r["sustained_steps"], "ms_per_step": r["sustained_ms_per_step"], "shader_clock_mhz": r.get("sustained_shader_clock_mhz"),
                       "what": "the same timed loop over %d steps, run right after the line's %d (no timing events): the clock a long run holds" % (r["sustained_steps"], steps)}
                      if r.get("sustained_steps") else None),
        "kernel_us_per_step": r.get("kernel_us_median", kt),
        "kernel_us_per_step_max": r.get("kernel_us_max"),
        "kernel_us_per_step_mean": kt,
        "kernel_times_from": "HIP events on rank 0's compute stream on %d of the %d timed steps (every %d%s; those steps run as plain launches): kernel_us_per_step "
                             "is the MEDIAN over those steps, _max the slowest, _mean what the rooflines use" %
                             (r["timed_launches"], steps, r["timing_period"], "th" if r["timing_period"] > 1 else "st"),
        # what an N-GPU run needs to explain itself: how many ranks RCCL really joined, every rank's own stage times (median
        # over the timed steps, HIP events on its compute stream), how long the compute stream WAITED for each phase's messages
        # (end of a stage to the start of the next: the host enqueues at once, only the transfer stream's event holds it back;
        # minimum and maximum over the ranks), and the bytes rank 0 sends per phase
        "rccl_ranks": r.get("rccl_ranks"),
        "stage_ms_per_rank": r.get("stage_ms_per_rank"),
        "wait_for_messages_ms": r.get("wait_ms"),
        "bytes_per_phase_rank0": r.get("bytes_per_phase_rank0"),
        "cpu_baseline": None,
    }
    return out


def cpu_cache_path(args):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), "psamd_cpu_baseline_n%d_s%d_g%dx%d.json" % (args.n, args.seed, args.chunk_factor, args.chunk_dim))


def cpu_baseline_cached(args, cfg_over, world):
    """The CPU oracle's figure for the multi-rank line: copied from a one-rank run on this box if there was one (bench.py
    --gpus 1 leaves it in TMPDIR), else measured now on a shorter sample -- after the ranks have finished."""
    path = cpu_cache_path(args)
    try:
        with open(path) as f:
            d = json.load(f)
        d["one"]["from"] = d["many"]["from"] = "copied from this box's one-rank run (%s)" % path
        return d["one"], d["many"]
    except Exception:
        pass
    return None, None



def main():
    args = parse_args()
    if args.n <= 0:
        args.n = (1 << 18) if args.all_pairs else (1 << 20)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "PSAMD_BENCH_CHILD" not in os.environ:
        launch_ranks(args)          # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    # the host driver of this pool only supports dmabuf IPC (RCCL between processes needs it); set before HIP loads
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if os.environ.get("PSAMD_BENCH_FAIL_RANK") == str(rank) and world > 1:
        sys.exit(3)                     # (test hook: a rank that dies before the rendezvous)
    if args.launch_check and args.backend == "ring" and world > 1:
        # the launcher of the C++ ranks, without a GPU: every rank's program meets the others through the id file
        import particlesystem_amd as ps
        if rank == 0:
            ps.build()
            ps._build.build_ring()
        else:
            for _ in range(600):
                if not ps._build.needs_build() and os.path.exists(ps._build.RING) and os.path.getmtime(ps._build.RING) >= os.path.getmtime(ps._build.LIB):
                    break
                time.sleep(0.5)
        ok, rec, err, _ = run_ring_rank(args, world, rank, local_rank, {})
        if not ok:
            sys.stderr.write("launch check of the C++ ranks failed: %s\n" % err)
            sys.exit(1)
        if rank == 0:
            print(json.dumps({"metric": "particle-updates/sec at N=2^20", "launch_check": True, "n_gpus": rec["world"], "steps": rec["steps"],
                              "warmup": rec["warmup"], "host": "C++ ranks (host/ps_ring_rccl)"}))
        return
    if args.launch_check:
        import torch.distributed as dist
        dist.init_process_group(args.backend if args.backend not in ("nccl", "ring") else "gloo")
        import torch
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"metric": "particle-updates/sec at N=2^20", "launch_check": True, "n_gpus": int(t.item()),
                              "steps": args.steps, "warmup": args.warmup}))
        dist.barrier()
        dist.destroy_process_group()
        return
    import particlesystem_amd as ps
    # build before anything touches the GPU or the process group; only one process compiles
    if rank == 0:
        ps.build()                   # no-op when the library is newer than its sources
    else:
        for _ in range(600):
            if not ps._build.needs_build():
                break
            time.sleep(0.5)
    cfg_over = dict(chunk_factor=args.chunk_factor, chunk_dim=args.chunk_dim,
                    max_particles_num=max(args.n, 1 << 20))
    if args.halo_cap_cell == 0 and not args.evolve and not args.all_pairs and (world > 1 or args.sim_world):
        # Slab messages have a fixed size, cells x halo_cap_cell bodies (the library's default is the
        # cell capacity, 2x the mean density at the reference's settings), and their room is pooled over a
        # cell layer.  The replayed step never changes the cloud, so size them for it: 1.15x the mean
        # density of the uniform cloud + 16 per cell -- a layer of G^2 cells holds n / G particles give or
        # take a few hundred (sqrt of it), so a fifth of headroom is hundreds of sigma; a message that did
        # not fit would be a loud error, not a truncation.  A free-running cloud (--evolve) keeps the default.
        cells = (args.chunk_factor * args.chunk_dim) ** 3
        args.halo_cap_cell = int(1.15 * args.n / cells) + 16
        # likewise the transfer messages (particles changing owner per step and direction): the library's
        # default has room for a quarter of what a layer can hold (a fast, dense cloud); this cloud's busiest
        # face is the box surface, whose layer implodes by up to a cell in the replayed step: an eighth of a
        # layer's population (n / G) is what it sends, measured; a message that did not fit is a loud error
        G = args.chunk_factor * args.chunk_dim
        cfg_over["xfer_cap"] = max(4096, int(args.n / G / 8) + 1024)
    if args.all_pairs and (world > 1 or args.sim_world):
        # all-pairs forces pull the whole uniform cloud inwards by the step's clamp (MAX_DX = one cell): in the
        # replayed step up to a whole cell layer (n / G particles) changes owner across a cut; the library's
        # default message has room for a quarter of a layer's capacity (half its mean population here)
        G = args.chunk_factor * args.chunk_dim
        cfg_over["xfer_cap"] = int(1.25 * args.n / G) + 4096
    flags = (ps.FLAG_FAST_MATH if args.fast_math else 0) | (ps.FLAG_ALL_PAIRS if args.all_pairs else 0)
    use_ring = (args.backend == "ring" if world > 1 else args.host == "ring") and not args.sim_world
    if use_ring:
        # The ranks that do the work are C++ programs -- for ONE GPU too: the whole scaling curve is driven by the same host
        # (host/ps_ring_rccl).  This rank's is started as a child process (nothing here has touched a GPU), rank 0 relays its
        # record.  If any rank's program fails, all ranks hear of it (files beside the id file) and run the step from Python
        # instead (world > 1: over torch.distributed) -- the line then says so, and a program that was ended by a signal or
        # by the time limit (a fault, a hang: not a clean refusal) makes this command exit non-zero behind its line.
        if rank == 0:
            ps._build.build_ring()
        else:
            for _ in range(600):
                if os.path.exists(ps._build.RING) and os.path.getmtime(ps._build.RING) >= os.path.getmtime(ps._build.LIB):
                    break
                time.sleep(0.5)
        ok, rec, err, violent = run_ring_rank(args, world, rank, local_rank, cfg_over)
        if ok:
            if rank == 0:
                out = line_from_ring_record(args, rec)
                if world == 1:
                    out["config"]["parallelism"] = "single GPU"
                    for key, prefix in (("roofline", "k_pairs_balanced"), ("roofline_streaming", "k_apply")):
                        out[key]["traffic"] = measured_traffic(prefix, args, world)
                        out[key]["traffic_source"] = TRAFFIC_FILE if out[key]["traffic"] is not None else None
                    out["roofline"]["kernel"] = "k_pairs_balanced (the packs of partly filled slices are workgroups of the same launch)"
                    out["roofline_streaming"]["kernel"] = "k_apply"
                    if not args.all_pairs and not args.no_cpu:
                        # the side runs and the CPU oracle, from this process (the C++ program has ended: the GPU is free)
                        g = ps.ParticleSystem(ps.default_config(device=local_rank, flags=flags, **cfg_over))
                        g.set_tdata_mirror(False)
                        xyz, age, fert = make_inputs(g, args.n, args.seed)
                        if not args.no_side_runs and not args.fast_math and not args.evolve:
                            g.fill_particles(xyz, age=age, fert_age=fert)
                            g.snapshot_save()
                            side_runs(out, args, ps, cfg_over, local_rank, xyz, age, fert, flags, max(1, min(args.timing_period, args.steps)), g)
                        g.close()
                        out["cpu_baseline"], out["cpu_baseline_many_threads"] = cpu_baseline(args, xyz, age, fert, cfg_over)
                        try:
                            with open(cpu_cache_path(args), "w") as f:
                                json.dump({"one": out["cpu_baseline"], "many": out["cpu_baseline_many_threads"]}, f)
                        except OSError:
                            pass
                elif not args.all_pairs:
                    out["cpu_baseline"], out["cpu_baseline_many_threads"] = cpu_baseline_cached(args, cfg_over, world)
                print(json.dumps(out))
            return
        sys.stderr.write("bench.py rank %d: the C++ host failed (%s); falling back to the Python host%s\n" %
                         (rank, err, " over torch.distributed" if world > 1 else ""))
        args.backend = os.environ.get("PSAMD_BENCH_FALLBACK_BACKEND", "nccl")      # (gloo: rehearsals on one GPU)
        args.ring_failed = err
        args.ring_violent = violent
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        gpu_backend = args.backend == "nccl"
        if gpu_backend:
            torch.cuda.set_device(local_rank)
        # librccl prints a version banner on stdout when the first communicator comes up;
        # the driver wants exactly one JSON line there, so park fd 1 on stderr meanwhile
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if gpu_backend:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                warm = torch.zeros(8, device="cuda")
                dist.all_reduce(warm)
                torch.cuda.synchronize()
            else:
                dist.init_process_group(args.backend)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    if args.sim_world:
        print(json.dumps(sim_world(args, ps, cfg_over, flags)))
        return
    device = local_rank if (world == 1 or args.backend == "nccl") else 0
    cfg = ps.default_config(device=device, rank=rank, world=world, flags=flags, halo_cap_cell=args.halo_cap_cell, **cfg_over)
    g = ps.ParticleSystem(cfg)
    # The reference's T_DATA rows are a mirror the library keeps for callers that fetch that buffer (psamd_download_tdata);
    # nothing in the step reads them and this host never fetches them: off (the parity tests run with it on AND compare
    # the rows; the snapshot the pair stage reads is built by the same kernel either way).
    g.set_tdata_mirror(False)
    xyz, age, fert = make_inputs(g, args.n, args.seed)
    g.fill_particles(xyz, age=age, fert_age=fert)       # a slab rank keeps the particles of its own segments
    G = g.sizes.grid_dim
    if args.graphs:
        g.set_graphs(True)
    if args.wait_policy >= 0:
        g.set_wait_policy(args.wait_policy)

    ring = None
    if world > 1:
        from particlesystem_amd.slab import DeviceRing, HostRing
        if args.backend == "nccl":
            # One explicit (non-default) stream carries the stage kernels AND is the stream
            # RCCL orders its transfers against, so pack -> send/recv -> unpack are ordered by
            # the stream alone.  (The legacy default stream has handle 0, which psamd_set_stream
            # reads as "use the context's own stream": never pass it.)
            stream = torch.cuda.Stream(device=local_rank)
            assert stream.cuda_stream != 0
            ring = DeviceRing(g, dist, rank, world, stream, overlap_interior=args.overlap_interior)
        else:
            ring = HostRing(g, dist, rank, world)

    g.snapshot_save()

    def one_step():
        if not args.evolve:
            g.snapshot_restore()
        if ring is None:
            g.step(1)
        else:
            ring.step()

    def sync():
        g.synchronize()
        if world > 1:
            if args.backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()

    def allsum(a):
        if world == 1:
            return a
        t = torch.from_numpy(np.ascontiguousarray(a, np.int64))
        if args.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t)
        return t.cpu().numpy()

    # The frame's census (downloads, host work) comes BEFORE the settling steps and the warmup, so that the warmup runs
    # straight into the timed region: with the census between them the GPU sat idle for tens of milliseconds and the
    # first timed steps ran at a lower clock -- a fixed 2.4 ms that a 20-step region showed as 5 % (2.33 against 2.21 ms
    # per step) and a 200-step region hid.  The events of the kernel timers are created here too.
    def frame_counts():
        """particles per cell and particles the force pass visits per cell, whole system"""
        if not args.evolve:
            g.snapshot_restore()
        if ring is None:
            g.init_iframe(); g.build_grid()
            n = g.download_cellgrid()[:, 0].copy()
            g.calc_forces_pairs()
            f = g.download_force_counts()
            mine = f
            g.calc_forces_apply()
        else:
            g.slab_build()
            n = g.download_cellgrid()[:, 0].copy()
            ring.exchange("halo")
            ring.finish_snapshot(ring.gather_snapshot())
            ring.finish_status(ring.gather_status())
            g.slab_pairs()
            mine = g.download_force_counts()
            ring.exchange("force")
            g.slab_apply()
            ring.exchange("xfer")
            ring.finish_far(ring.gather_far())
            g.slab_finish()
            n, f = allsum(n), allsum(mine)
        return n, f, mine

    if args.evolve:
        counts0 = fcounts0 = mine0 = None
    else:
        counts0, fcounts0, mine0 = frame_counts()
    g.set_timing(True)
    g.set_timing(False)
    # (the clock watchers are made here as well: finding the card in sysfs imports torch and asks it for the device's PCI
    # address -- hundreds of milliseconds with the GPU idle if it happened between the warmup and the timed region)
    clock = ClockWatch(local_rank if args.backend == "nccl" or world == 1 else 0)
    clock2 = ClockWatch(local_rank)
    # untimed: let the clocks settle (the first launches of a process run at a lower clock).
    # All ranks must take the same number of steps: they decide together, ten steps at a time.
    settle = 0
    t_end = time.perf_counter() + args.settle_seconds
    while not args.evolve and int(allsum(np.array([1 if time.perf_counter() < t_end else 0]))[0]) == world:
        for _ in range(10):
            one_step()
        settle += 10
    for _ in range(args.warmup):
        one_step()
    sync()

    # the events that time the kernels go in on every timing_period-th step (each costs ~6 us of idle GPU
    # between two kernels); kernel_us_per_step and the rooflines are means over those steps
    period = 1 if args.kernel_times else max(1, min(args.timing_period, args.steps))
    g.set_timing(True, every_stage=args.kernel_times, period=period)
    sync()
    processed0 = g.counters["particles_processed"]
    with clock:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            one_step()
        sync()
        elapsed = time.perf_counter() - t0
    tim, launches = g.timing()
    tmed, tmax, _ = g.timing_stats()
    g.set_timing(False)
    ctr = g.counters
    own_updates = float(ctr["particles_processed"] - processed0)
    sustained = None
    if world == 1 and not args.evolve and 0 < args.steps < args.sustained_steps:
        # A timed region of tens of milliseconds (the driver's --steps 20) says little about the clock a long run holds:
        # the same loop again, long enough to show the sustained figure beside it.  (Until the census and the clock
        # watchers' setup moved ahead of the warmup, the short region was the SLOWER of the two: it started on a GPU
        # that had sat idle, and its first dozen steps ran at a lower clock.)
        sync()
        with clock2:
            t1 = time.perf_counter()
            for _ in range(args.sustained_steps):
                one_step()
            sync()
            t2 = time.perf_counter()
        sustained = {"steps": args.sustained_steps, "ms_per_step": 1e3 * (t2 - t1) / args.sustained_steps, "shader_clock_mhz": clock2.summary(),
                     "what": "the same timed loop over %d steps, run right after the line's %d (no timing events): the clock a long run holds"
                             % (args.sustained_steps, args.steps)}
    counts1, fcounts1, mine1 = frame_counts()
    if counts0 is None:
        counts0, fcounts0, mine0 = counts1, fcounts1, mine1
    live = int(counts1.sum())
    updates = float(allsum(np.array([int(own_updates)]))[0])
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        if args.backend == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # a particle-update = one live particle taken through one step; the population is
        # not constant (cell-overflow and full-segment losses as the cloud collapses)
        value = updates / elapsed
        # the force kernel of THIS rank: its own share of the pairs against its own launch time
        pairs_rank = 0.5 * (force_terms(counts0, mine0, G, args.all_pairs) + force_terms(counts1, mine1, G, args.all_pairs))
        us_pairs = tim["pairs"] / max(launches, 1)
        us_apply = tim["apply"] / max(launches, 1)
        ach_tflops = pairs_rank * FLOP_PER_PAIR / (us_pairs * 1e-6) / 1e12 if us_pairs > 0 else 0.0
        ach_gbs = own_updates / args.steps * APPLY_BYTES_PER_UPDATE / (us_apply * 1e-6) / 1e9 if us_apply > 0 else 0.0
        traffic_pairs = measured_traffic("k_pairs_balanced", args, world)
        traffic_apply = measured_traffic("k_apply", args, world)
        out = {
            "metric": "particle-updates/sec at N=2^20" if not args.all_pairs else "particle-updates/sec, all-pairs forces (BASELINE configs[1]), N=%d" % args.n,
            "value": value, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: N=%d uniform cloud, %d^3 cells x 5.0, ALL-PAIRS gravity (every particle against every "
                                    "body: the 27-cell stencil in the reference's order, then every other cell; not in the reference, parity "
                                    "unpinned) + reference collisions/integrate/wrap/relocation, %s" if args.all_pairs else
                                    "BASELINE configs[2]: N=%d uniform cloud, %d^3 cells x 5.0, 27-cell cutoff gravity "
                                    "+ reference collisions/integrate/wrap/relocation, all constants at reference defaults, "
                                    "%s") % (args.n, G, "free-running" if args.evolve else "each step = one pass over the same cloud (restored in HBM)"),
                       "arithmetic": "fast-math" if args.fast_math else "reference-exact fp32 (bitwise parity mode)",
                       "parallelism": ("%d slabs of cell layers, one per GPU (state partitioned by segment); halo / force / "
                                       "transfer messages between ring neighbours over %s" % (world, "RCCL send/recv" if args.backend == "nccl" else args.backend))
                                      if world > 1 else "single GPU",
                       "graphs": ("stage sequences as hipGraphs: %d replays, %d captures" % g.graph_stats()) if args.graphs else None,
                       "tdata_mirror": False, "run_ahead_steps": 1,
                       "updates_in_timed_region": updates, "live_after": live, "settle_steps_before_warmup": settle,
                       "particles_with_a_force_term": int(fcounts1.sum()),
                       "relocations": ctr["relocations"], "relocations_lost": ctr["relocations_lost"],
                       "cell_overflow_kills": ctr["cell_overflow_kills"]},
            "roofline": {"kernel": ("k_pairs + k_allpairs_combine (all-pairs walk)" if args.all_pairs else "k_pairs_balanced (the packs of partly filled slices are workgroups of the same launch)"), "bound": "valu", "achieved": ach_tflops, "peak": VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach_tflops / VALU_PEAK_TFLOPS, "traffic": traffic_pairs,
                         "traffic_source": TRAFFIC_FILE if traffic_pairs is not None else None,
                         "pairs_per_launch": pairs_rank, "flop_per_pair": FLOP_PER_PAIR, "us_per_launch": us_pairs},
            "roofline_streaming": {"kernel": "k_apply", "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic_apply,
                                   "traffic_source": TRAFFIC_FILE if traffic_apply is not None else None,
                                   "bytes_per_update": APPLY_BYTES_PER_UPDATE, "us_per_launch": us_apply},
            "shader_clock_mhz": clock.summary(),
            "sustained": sustained,
            "kernel_us_per_step": {k: v for k, v in tmed.items() if v > 0},
            "kernel_us_per_step_max": {k: v for k, v in tmax.items() if v > 0},
            "kernel_us_per_step_mean": {k: v / max(launches, 1) for k, v in tim.items() if v > 0},
            "kernel_times_from": "HIP events on the context's stream on %d of the %d timed steps (every %d%s): kernel_us_per_step is the MEDIAN over "
                                 "those steps, _max the slowest, _mean what the rooflines use" % (launches, args.steps, period, "th" if period > 1 else "st"),
        }
        out["config"]["host"] = ("Python ranks over torch.distributed (particlesystem_amd/slab.py)" if world > 1 else "this Python process through ctypes (psamd_step)") + \
            (": FALLBACK, the C++ host failed (%s)" % args.ring_failed if getattr(args, "ring_failed", None) else "")
        if getattr(args, "ring_failed", None):
            out["ring_failed"] = args.ring_failed
        if world > 1:
            out["config"]["halo_cap_cell"] = int(args.halo_cap_cell)
            out["config"]["message_bytes_rank0"] = {name: int(g.msg_bytes(k)) for name, k in
                                                    (("halo_up", 1), ("halo_down", 0), ("force_in", 5), ("xfer_each", 6))}
        if args.all_pairs:
            out["roofline"]["kernel"] = "k_pairs (all-pairs walk: stencil in the reference's order, then every other cell, summed per cell)"
            out["roofline"]["flop_per_step_convention"] = "20 flop x N^2 (SURVEY 8d): %.3g" % (20.0 * args.n * args.n)
            out["cpu_baseline"] = None       # the reference has no all-pairs force; the oracle restates the reference only
        if world == 1 and not args.no_cpu and not args.all_pairs:
            if not args.fast_math and not args.evolve and not args.no_side_runs and not args.all_pairs:
                side_runs(out, args, ps, cfg_over, local_rank, xyz, age, fert, flags, period, g)
            out["cpu_baseline"], out["cpu_baseline_many_threads"] = cpu_baseline(args, xyz, age, fert, cfg_over)
        elif not args.all_pairs:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    g.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if getattr(args, "ring_violent", False):
        sys.exit(4)         # the line above is valid (the Python host's), but a C++ rank was ended by a signal or the limit: look into it


if __name__ == "__main__":
    main()
