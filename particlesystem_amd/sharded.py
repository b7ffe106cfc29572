"""One step of the sharded (multi-GPU) path: every rank holds the whole container; the
pair loop is split by sorted-particle range; one all-gather of the float4 results per
step; integrate + lifecycle replicated (so all ranks stay bit-identical to a single-GPU
run).  Used by bench.py with RCCL on device memory and by the tests with gloo.

`system` is anything with the stage interface of particlesystem_amd.ParticleSystem
(init_iframe, build_grid, force_shard, calc_forces_pairs, calc_forces_apply) whose pair
pass writes into `force` (a [>= world*share, 4] float32 torch tensor)."""


def shard_bounds(total, rank, world):
    """[lo, hi) of sorted particles rank `rank` evaluates, and the padded common share."""
    share = -(-total // world)
    lo = min(total, share * rank)
    return lo, min(total, lo + share), share


def step_sharded(system, force, dist, rank, world, always_gather=False):
    system.init_iframe()
    system.build_grid()
    _, _, share = system.force_shard()
    system.calc_forces_pairs()
    if world > 1 or always_gather:     # a one-rank gather is only a rehearsal of the collective
        full = force[: world * share]
        dist.all_gather_into_tensor(full, full[rank * share:(rank + 1) * share])
    system.calc_forces_apply()
