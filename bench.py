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
exactly the work the CPU baseline samples.  --evolve runs free instead and counts the
live particles of every step.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: every rank keeps the whole (bit-identical) container; the pair loop -- 99.9 % of
the step -- is sharded by sorted-particle range, and one RCCL all-gather per step
exchanges the float4 (ax, ay, az, flag) results (16 B per particle, the size of the
position all-gather it replaces); integrate + lifecycle are replicated streaming work.
Total work is fixed at N = 2^20 as the GPU count grows => "scaling": "strong".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 vector (spec)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
FLOP_PER_PAIR = 20           # the usual N-body convention
APPLY_BYTES_PER_UPDATE = 64  # read pos4+vel4, write pos4+vel4


def make_inputs(ps, sysobj, n, seed):
    xyz = sysobj.uniform_cloud(n, seed)
    rng = np.random.default_rng(seed)
    life = 300.0 * sysobj.cfg.dt
    age = rng.uniform(life / 7.0, life / 2.0, n).astype(np.float32)   # [MIN_ADULT_AGE, MAX_ADULT_AGE)
    fert = np.full(n, 1e6, np.float32)
    return xyz, age, fert


def measured_traffic(kernel_prefix):
    """HBM bytes per launch from the committed rocprofv3 PMC capture (profiles/), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r1_traffic.json")) as f:
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
    have passed.  `cpu_baseline_all_cores`: the read-only pair pass (collision scan + force
    loop, >99.9 % of the CPU step) on every usable host core, contiguous shares per thread."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    o = O.System(O.default_config(**cfg_over))
    o.fill(xyz, age=age, fert_age=fert)
    o.init_iframe()
    o.build_grid()
    many = None
    cores = min(len(os.sched_getaffinity(0)), args.cpu_threads)
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
                          "box) of the same N=%d cloud on %d host threads, %.1f s" % (seg * segs, segs, len(xyz), cores, tm)}
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


def fast_math_rate(ps, cfg_over, device, xyz, age, fert, steps=5):
    """For the record, never the headline: the same timed loop with PSAMD_FLAG_FAST_MATH
    (FMA + v_rsq pair arithmetic: accelerations within 1e-5 relative of the oracle,
    tests/test_gpu_fast.py; the headline runs the bit-exact arithmetic)."""
    g = ps.ParticleSystem(ps.default_config(device=device, flags=ps.FLAG_FAST_MATH, **cfg_over))
    g.fill_particles(xyz, age=age, fert_age=fert)
    g.snapshot_save()
    for _ in range(2):
        g.snapshot_restore(); g.step(1)
    g.synchronize()
    p0 = g.counters["particles_processed"]
    t0 = time.perf_counter()
    for _ in range(steps):
        g.snapshot_restore(); g.step(1)
    g.synchronize()
    dt = time.perf_counter() - t0
    done = g.counters["particles_processed"] - p0
    g.close()
    return {"arithmetic": "PSAMD_FLAG_FAST_MATH (FMA + v_rsq), within 1e-5 relative of the oracle; not the headline",
            "value": done / dt, "unit": "particle-updates/s", "ms_per_step": 1e3 * dt / steps, "steps": steps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--fast-math", action="store_true", help="FMA/rsq pair arithmetic (not bit-exact)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-threads", type=int, default=64, help="cap on host threads for the all-core CPU figure")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--kernel-times", action="store_true", help="HIP events between all stage kernels, not only around the "
                    "pair pass / apply / life cycle (costs ~40 us of idle GPU per step)")
    ap.add_argument("--seed", type=int, default=2026)
    ap.add_argument("--chunk-factor", type=int, default=4, help="grid = (chunk_factor*chunk_dim)^3 cells (reference: 4)")
    ap.add_argument("--chunk-dim", type=int, default=4)
    ap.add_argument("--evolve", action="store_true", help="free-running steps instead of one pass per step over the same cloud")
    ap.add_argument("--force-dist", action="store_true", help="take the collective code path even with one rank (rehearsal)")
    ap.add_argument("--sim-world", type=int, default=0, help="diagnostic: time ONE rank's work of an N-rank run on one GPU "
                    "(its pair shard + the replicated stages; no collective, results are not valid physics)")
    ap.add_argument("--sim-rank", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import particlesystem_amd as ps
    if not os.path.exists(ps.LIB_PATH):
        if rank == 0:
            ps.build()
    from particlesystem_amd.sharded import step_sharded
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if world == 1 and "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29544", RANK="0", WORLD_SIZE="1")
        # librccl prints a version banner on stdout when the first communicator comes up;
        # the driver wants exactly one JSON line there, so park fd 1 on stderr meanwhile
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            warm = torch.zeros(world * 4, device="cuda")
            dist.all_gather_into_tensor(warm, warm[rank * 4:(rank + 1) * 4])
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    cfg_over = dict(chunk_factor=args.chunk_factor, chunk_dim=args.chunk_dim,
                    max_particles_num=max(args.n, 1 << 20))
    flags = ps.FLAG_FAST_MATH if args.fast_math else 0
    cfg = ps.default_config(device=local_rank, rank=args.sim_rank if args.sim_world else rank,
                            world=args.sim_world if args.sim_world else world, flags=flags, **cfg_over)
    g = ps.ParticleSystem(cfg)
    xyz, age, fert = make_inputs(ps, g, args.n, args.seed)
    g.fill_particles(xyz, age=age, fert_age=fert)
    G = g.sizes.grid_dim

    force = None
    if use_dist:
        # One explicit (non-default) stream carries the stage kernels AND is the stream
        # RCCL orders its collectives against, so pair pass -> all-gather -> apply are
        # ordered by the stream alone.  (The legacy default stream has handle 0, which
        # psamd_set_stream reads as "use the context's own stream": never pass it.)
        torch.cuda.set_device(local_rank)
        stream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(stream)
        assert stream.cuda_stream != 0
        cap = g.sizes.container_size + world
        force = torch.zeros((cap, 4), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        g.bind_force4(force.data_ptr(), cap)
        g.set_stream(stream.cuda_stream)

    g.snapshot_save()

    def one_step():
        if not args.evolve:
            g.snapshot_restore()
        if args.sim_world:
            g.init_iframe(); g.build_grid(); g.force_shard(); g.calc_forces_pairs(); g.calc_forces_apply()
            return
        if not use_dist:
            g.step(1)
            return
        step_sharded(g, force, dist, rank, world, always_gather=True)

    def sync():
        g.synchronize()
        if use_dist:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    sync()
    # pairs per launch, measured on the state the timed steps start from
    if not args.evolve:
        g.snapshot_restore()
    def frame_counts():
        g.init_iframe(); g.build_grid()
        n = g.download_cellgrid()[:, 0].copy()
        if world > 1 or args.sim_world:
            g.force_shard()
        g.calc_forces_pairs()
        f = g.download_force_counts()
        if use_dist and world > 1:        # a rank lists only its own share's particles
            t = torch.from_numpy(f.astype(np.int64)).cuda()
            dist.all_reduce(t)
            f = t.cpu().numpy()
        elif args.sim_world:              # one rank's share timed alone: scale to the whole for the rate below
            f = f * args.sim_world
        return n, f

    counts0, fcounts0 = frame_counts()
    g.set_timing(True, every_stage=args.kernel_times)
    sync()
    processed0 = g.counters["particles_processed"]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    sync()
    elapsed = time.perf_counter() - t0
    tim, launches = g.timing()
    g.set_timing(False)
    ctr = g.counters
    if not args.evolve:
        g.snapshot_restore()
    counts1, fcounts1 = frame_counts()
    live = int(counts1.sum())

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # a particle-update = one live particle taken through one step; the population is
        # not constant (cell-overflow and full-segment losses as the cloud collapses)
        updates = float(g.counters["particles_processed"] - processed0)
        value = updates / elapsed
        pairs = 0.5 * (pair_count(counts0, fcounts0, G) + pair_count(counts1, fcounts1, G))
        pairs_rank = pairs / world
        us_pairs = tim["pairs"] / max(launches, 1)
        us_apply = tim["apply"] / max(launches, 1)
        ach_tflops = pairs_rank * FLOP_PER_PAIR / (us_pairs * 1e-6) / 1e12 if us_pairs > 0 else 0.0
        ach_gbs = updates / args.steps * APPLY_BYTES_PER_UPDATE / (us_apply * 1e-6) / 1e9 if us_apply > 0 else 0.0
        out = {
            "metric": "particle-updates/sec at N=2^20", "value": value, "unit": "particle-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: N=%d uniform cloud, %d^3 cells x 5.0, 27-cell cutoff gravity "
                                   "+ reference collisions/integrate/wrap/relocation, all constants at reference defaults, "
                                   "%s" % (args.n, G, "free-running" if args.evolve else "each step = one pass over the same cloud (restored in HBM)"),
                       "arithmetic": "fast-math" if args.fast_math else "reference-exact fp32 (bitwise parity mode)",
                       "parallelism": "pair loop sharded x%d + RCCL all-gather of float4 results" % world if world > 1 else "single GPU",
                       "updates_in_timed_region": updates, "live_after": live,
                       "particles_with_a_force_term": int(fcounts1.sum()),
                       "relocations": ctr["relocations"], "relocations_lost": ctr["relocations_lost"],
                       "cell_overflow_kills": ctr["cell_overflow_kills"]},
            "roofline": {"kernel": "k_pairs", "bound": "valu", "achieved": ach_tflops, "peak": VALU_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach_tflops / VALU_PEAK_TFLOPS,
                         "traffic": measured_traffic("k_pairs<1") if world == 1 and not args.fast_math else None,
                         "pairs_per_launch": pairs_rank, "flop_per_pair": FLOP_PER_PAIR, "us_per_launch": us_pairs},
            "roofline_streaming": {"kernel": "k_apply", "bound": "hbm", "achieved": ach_gbs, "peak": HBM_PEAK_GBS,
                                   "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS, "traffic": measured_traffic("k_apply"),
                                   "bytes_per_update": APPLY_BYTES_PER_UPDATE, "us_per_launch": us_apply},
            "kernel_us_per_step": {k: v / max(launches, 1) for k, v in tim.items() if v > 0},
        }
        if world == 1 and not args.no_cpu:
            if not args.fast_math and not args.evolve and not args.sim_world:
                out["within_tolerance_mode"] = fast_math_rate(ps, cfg_over, local_rank, xyz, age, fert)
            out["cpu_baseline"], out["cpu_baseline_all_cores"] = cpu_baseline(args, xyz, age, fert, cfg_over)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    g.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
