"""What BASELINE.json names and the reference does not contain ("parity
unpinned"): all-pairs forces (configs[1]), a drag term, an explicit-Euler position update,
repulsion.  Nothing in the reference can pin them, so the checks are: the limit in which they
must coincide with the pinned cutoff path (bit for bit), an fp64 direct sum (1e-5 relative,
BASELINE's bar), and closed forms."""
import numpy as np
import pytest

import oracle_py as O
import particlesystem_amd as ps
from util import assert_same_particles, cloud, oracle_cfg_from

pytestmark = pytest.mark.gpu


def force_of(g, n):
    g.init_iframe(); g.build_grid(); g.calc_forces_pairs()
    f = g.download_force4(0, n)
    ids = g.download_cellgrid()
    order = np.concatenate([row[1:1 + row[0]] for row in ids])
    g.calc_forces_apply()
    return order, f


def test_all_pairs_is_the_cutoff_result_when_everything_is_within_one_stencil():
    """a cloud inside a 2x2x2 block of cells: every cell of the block is in every other's
    stencil, so the all-pairs walk (stencil first, in the reference's order, then the rest --
    empty) must reproduce the reference's cutoff sums bit for bit, and the whole step too."""
    n = 3000
    rng = np.random.default_rng(7)
    xyz = rng.uniform(-4.99, 4.99, (n, 3)).astype(np.float32)           # cells 7..8 on every axis
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    a = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS))
    b = ps.ParticleSystem(ps.default_config())
    o = O.System(oracle_cfg_from(b.cfg))
    for s in (a, b):
        s.fill_particles(xyz, age=age, fert_age=np.float32(1e6))
    o.fill(xyz, age=age, fert_age=np.float32(1e6))
    oa, fa = force_of(a, n)
    ob, fb = force_of(b, n)
    assert np.array_equal(oa, ob) and fa.tobytes() == fb.tobytes()
    o.step(1)
    assert_same_particles(a.download_particles(), o.particles, "all-pairs step on a confined cloud")
    a.close(); b.close(); o.close()


def test_all_pairs_against_an_fp64_direct_sum():
    """a cloud spread over the whole box: every particle feels every other (softened gravity,
    kids excluded as in bodyBodyInteraction); |a| within 1e-5 relative of the fp64 direct sum"""
    n = 4000
    xyz = cloud(n, 8)
    rng = np.random.default_rng(8)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    age[::17] = 0.5                                                       # some kids: exert and feel nothing
    g = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS, collision_radius=1e-6))
    ids = g.fill_particles(xyz, age=age, fert_age=np.float32(1e6))
    order, f = force_of(g, n)
    p = g.download_particles()
    live = np.nonzero(p["cell"] >= 0)[0]
    # the positions of the frame the forces were computed in = the fill (apply ran afterwards)
    pos = xyz.astype(np.float64)
    kid = age < 1.5
    d = pos[None, :, :] - pos[:, None, :]
    r2 = (d * d).sum(2) + 0.2
    s = np.where(kid[None, :], 0.0, 60.0 / (r2 * np.sqrt(r2)))
    np.fill_diagonal(s, 0.0)
    exact = (d * s[:, :, None]).sum(1)
    exact[kid] = 0.0
    # map sorted order -> input order through the slot ids the fill returned
    where = {int(s_): i for i, s_ in enumerate(ids)}
    idx = np.array([where[int(s_)] for s_ in order])
    got = f[:, :3].astype(np.float64)
    want = exact[idx]
    assert (f[:, 3].view(np.int32) == 0).all()
    adults = ~kid[idx]
    rel = np.linalg.norm(got[adults] - want[adults], axis=1) / np.linalg.norm(want[adults], axis=1)
    print("all-pairs vs fp64 direct sum, N=%d: max relative deviation %.3g" % (n, rel.max()))
    assert rel.max() < 1e-5 and not got[~adults].any() and len(live) == n
    g.close()


def lone_particles(v0, **over):
    """particles far apart (no neighbour within a stencil): no force, no collision"""
    xyz = np.array([[-30.0, -30.0, -30.0], [0.0, 0.0, 0.0], [30.0, 30.0, 30.0]], np.float32)
    g = ps.ParticleSystem(ps.default_config(**over))
    ids = g.fill_particles(xyz, age=np.float32(1.0), fert_age=np.float32(1e6), vxyz=np.asarray(v0, np.float32))
    return g, ids, xyz


def test_drag_decays_like_v0_exp_minus_kt():
    k, dt, steps = 0.8, 0.01, 100
    v0 = np.array([[1.0, -2.0, 0.5], [0.3, 0.2, -0.1], [-1.5, 0.0, 2.0]], np.float32)
    g, ids, _ = lone_particles(v0, drag=k, dt=dt)
    v = v0.copy()
    for _ in range(steps):
        g.step(1)
        a = -(np.float32(k) * v)                                        # a = 0 - k*v, fp32
        v = v + a * np.float32(dt)
    p = g.download_particles()
    # the relocated slots: find the three live particles by their ages
    live = np.nonzero(p["cell"] >= 0)[0]
    assert len(live) == 3
    got = np.stack([p["vx"][live], p["vy"][live], p["vz"][live]], 1)
    assert np.array_equal(np.sort(got, axis=0), np.sort(v, axis=0)), "explicit update v += (a - k v) dt, fp32"
    exact = v0 * np.exp(-k * dt * steps)
    assert np.abs(np.sort(got, axis=0) - np.sort(exact, axis=0)).max() < 2 * k * k * dt * steps * dt * np.abs(v0).max()
    g.close()


def test_euler_switch_and_default_position_update():
    v0 = np.array([[1.0, -2.0, 0.5], [0.3, 0.2, -0.1], [-1.5, 0.0, 2.0]], np.float32)
    for flags in (0, ps.FLAG_EULER):
        g, ids, xyz = lone_particles(v0, flags=flags)
        g.step(1)
        p = g.download_particles()
        live = np.nonzero(p["cell"] >= 0)[0]
        got = np.stack([p["x"][live], p["y"][live], p["z"][live]], 1)
        want = xyz + v0 * np.float32(0.05)                                # a = 0: both forms coincide
        assert np.array_equal(np.sort(got, axis=0), np.sort(want, axis=0))
        g.close()
    # with a force: two bodies 8 apart; Euler leaves out 0.5*a*dt^2
    two = np.array([[-4.0, 0, 0], [4.0, 0, 0]], np.float32)
    xs = {}
    for flags in (0, ps.FLAG_EULER):
        g = ps.ParticleSystem(ps.default_config(flags=flags))
        g.fill_particles(two, age=np.float32(2.0), fert_age=np.float32(1e6))
        g.step(1)
        p = g.download_particles()
        live = np.nonzero(p["cell"] >= 0)[0]
        xs[flags] = (np.sort(p["x"][live]), np.sort(p["vx"][live]), np.sort(p["ax"][live]))
        g.close()
    assert np.float32(xs[0][0][0]) == np.float32(-3.99883366)            # the reference's own number
    assert np.array_equal(xs[ps.FLAG_EULER][0], np.array([-4.0, 4.0], np.float32))
    assert np.array_equal(xs[0][1], xs[ps.FLAG_EULER][1]) and np.array_equal(xs[0][2], xs[ps.FLAG_EULER][2])


def test_repulsion_is_gravity_with_the_sign_flipped():
    n = 20000
    xyz = cloud(n, 9)
    rng = np.random.default_rng(9)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    f = {}
    for sign in (1.0, -1.0):
        g = ps.ParticleSystem(ps.default_config(force_sign=sign))
        g.fill_particles(xyz, age=age, fert_age=np.float32(1e6))
        f[sign] = force_of(g, n)[1]
        g.close()
    assert np.array_equal(f[1.0][:, 3].view(np.int32), f[-1.0][:, 3].view(np.int32))
    keep = f[1.0][:, 3].view(np.int32) == 0
    a, b = f[1.0][keep, :3], f[-1.0][keep, :3]
    assert np.array_equal(a.view(np.uint32) ^ np.uint32(0x80000000), b.view(np.uint32)) or np.array_equal(-a, b)
    assert np.abs(a).max() > 0


@pytest.mark.timeout(900)
def test_all_pairs_at_config1_size():
    """BASELINE configs[1]: N = 2^18, all-pairs gravity.  No CPU can re-do 6.9e10 pairs in a test,
    so: a size-independent property -- equal masses, so the accelerations of all adults sum to
    zero (Newton's third law; each pair's two terms differ only by rounding) -- plus 200 particles
    against an fp64 direct sum over all 2^18 bodies (1e-5 relative, BASELINE's bar)."""
    n = 1 << 18
    g = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS, collision_radius=1e-6))
    xyz = g.uniform_cloud(n, 18)
    rng = np.random.default_rng(18)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    ids = g.fill_particles(xyz, age=age, fert_age=np.float32(1e6))
    order, f = force_of(g, n)
    a = f[:, :3].astype(np.float64)
    assert (f[:, 3].view(np.int32) == 0).all()
    net = np.linalg.norm(a.sum(0)) / np.linalg.norm(a, axis=1).sum()
    print("all-pairs N=2^18: |sum a| / sum |a| = %.3g" % net)
    assert net < 1e-5
    where = np.empty(g.sizes.container_size, np.int64)
    where[ids] = np.arange(n)
    idx = where[order]
    pos = xyz.astype(np.float64)
    pick = rng.choice(n, 200, replace=False)
    d = pos[None, :, :] - pos[idx[pick], None, :]
    r2 = (d * d).sum(2) + 0.2
    s = 60.0 / (r2 * np.sqrt(r2))
    s[np.arange(200), idx[pick]] = 0.0
    exact = (d * s[:, :, None]).sum(1)
    rel = np.linalg.norm(a[pick] - exact, axis=1) / np.linalg.norm(exact, axis=1)
    print("all-pairs N=2^18 vs fp64 direct sum (200 particles): max relative deviation %.3g" % rel.max())
    assert rel.max() < 1e-5
    g.close()


@pytest.mark.parametrize("chunk_factor,chunk_dim,world", [(3, 4, 3), (3, 5, 3), (5, 3, 2)])
def test_all_pairs_on_grids_that_are_no_multiple_of_four(chunk_factor, chunk_dim, world):
    """The far walk takes four consecutive global cells as one chain of additions and 64-cell blocks as the unit of
    its 16 parts: grids of 12^3 and 15^3 cells (chains straddle rows of the grid, the last block is ragged, a part
    may hold no block at all) against an fp64 direct sum, and across slabs against one GPU, byte for byte."""
    from particlesystem_amd.slab import merge_owned, step_local
    over = dict(chunk_factor=chunk_factor, chunk_dim=chunk_dim)
    G = chunk_factor * chunk_dim
    n = 5000
    rng = np.random.default_rng(300 + G)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    age[::19] = 0.5
    fert = (1e6 + np.arange(n)).astype(np.float32)
    g = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS, collision_radius=1e-6, **over))
    xyz = g.uniform_cloud(n, 300 + G)                  # (an odd grid's box is not centred on the origin)
    ids = g.fill_particles(xyz, age=age, fert_age=fert)
    order, f = force_of(g, n)
    pos = xyz.astype(np.float64)
    kid = age < 1.5
    d = pos[None, :, :] - pos[:, None, :]
    r2 = (d * d).sum(2) + 0.2
    sc = np.where(kid[None, :], 0.0, 60.0 / (r2 * np.sqrt(r2)))
    np.fill_diagonal(sc, 0.0)
    exact = (d * sc[:, :, None]).sum(1)
    where = np.empty(g.sizes.container_size, np.int64)
    where[ids] = np.arange(n)
    idx = where[order]
    adults = ~kid[idx]
    got = f[:, :3].astype(np.float64)[adults]
    want = exact[idx][adults]
    rel = np.linalg.norm(got - want, axis=1) / np.linalg.norm(want, axis=1)
    print("all-pairs on %d^3 cells vs fp64 direct sum: max relative deviation %.3g" % (G, rel.max()))
    assert rel.max() < 1e-5
    g.close()
    one = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS, **over))
    ranks = [ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS, rank=r, world=world, **over)) for r in range(world)]
    for s_ in [one] + ranks:
        s_.fill_particles(xyz, age=age, fert_age=fert)
    plans = [q.slab_plan() for q in ranks]
    for step in range(4):
        one.step(1)
        step_local(ranks)
        union = merge_owned([q.download_particles() for q in ranks], plans)
        assert_same_particles(union, one.download_particles(), "all-pairs, %d^3 cells, %d slabs, step %d" % (G, world, step + 1))
    for s_ in [one] + ranks:
        s_.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_all_pairs_across_slabs_equals_one_gpu(world):
    """SURVEY 8(e) row 1 / BASELINE configs[3]'s exchange: every rank contributes the snapshot of its
    own cells to an ALL-GATHER once per step and walks the gathered buffer -- stencil first, then every
    other cell in global index order, summed per cell -- exactly as a single GPU walks its own
    snapshot.  Same order, same bits: the union of the ranks' states must equal the one-GPU all-pairs
    run byte for byte, every step (particles changing owner, collisions and relocations included)."""
    from particlesystem_amd.slab import merge_owned, step_local
    n = 20000
    xyz = cloud(n, 40 + world)
    rng = np.random.default_rng(40 + world)
    age = rng.uniform(15 / 7, 7.5, n).astype(np.float32)
    age[::23] = 0.5                                                       # kids: feel and exert nothing
    fert = (1e6 + np.arange(n)).astype(np.float32)
    one = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS))
    ranks = [ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS, rank=r, world=world)) for r in range(world)]
    for s in [one] + ranks:
        s.fill_particles(xyz, age=age, fert_age=fert)
    assert ranks[0].msg_bytes(ps.MSG_ALLG_OUT) > 0 and ranks[0].msg_bytes(ps.MSG_ALLG_IN) == world * ranks[0].msg_bytes(ps.MSG_ALLG_OUT)
    plans = [g.slab_plan() for g in ranks]
    for step in range(6):
        one.step(1)
        step_local(ranks)
        union = merge_owned([g.download_particles() for g in ranks], plans)
        assert_same_particles(union, one.download_particles(), "all-pairs, %d slabs, step %d" % (world, step + 1))
    c = one.counters
    assert c["relocations"] > 0 and c["integrated"] > 0
    # and the far field really is in the sums: the cutoff-only run of the same cloud differs
    cut = ps.ParticleSystem(ps.default_config())
    cut.fill_particles(xyz, age=age, fert_age=fert)
    cut.step(1)
    one2 = ps.ParticleSystem(ps.default_config(flags=ps.FLAG_ALL_PAIRS))
    one2.fill_particles(xyz, age=age, fert_age=fert)
    one2.step(1)
    assert cut.download_particles().tobytes() != one2.download_particles().tobytes()
    for s in [one, cut, one2] + ranks:
        s.close()
