"""Known answers that SURVEY.md section 8(c) records from the reference's own
`_host` stage loop (unmodified particleSystem.cpp run in the survey container):
they pin the oracle's restatement of the stage bodies and of the lifecycle."""
import os

import numpy as np
import pytest

import oracle_py as O


def f32(x):
    return float(np.float32(x))


def test_two_body_known_answers():
    """Scenario of ps.cpp:1033-1034: bodies at (-4,0,0) and (4,0,0), age 40*DT, mass 60."""
    s = O.System()
    ids = s.fill([[-4, 0, 0], [4, 0, 0]], age=np.float32(40 * 0.05), fert_age=1e6)
    assert list(ids) == [2738592, 2738593]
    p = s.particles
    assert [tuple(int(p[f][i]) for f in ("cell", "chunk", "seg_type", "seg_tid")) for i in ids] == \
        [(2183, 41, 8, 62), (2184, 42, 8, 62)]
    s.step(1)
    a, b = s.particles[ids[0]], s.particles[ids[1]]
    assert f32(a["ax"]) == f32(0.933122575) and f32(b["ax"]) == f32(-0.933122575)
    assert f32(a["vx"]) == f32(0.046656128) and f32(b["vx"]) == f32(-0.046656128)
    assert f32(a["x"]) == f32(-3.99883366) and f32(b["x"]) == f32(3.99883366)
    assert f32(a["age"]) == f32(2.04999995)
    assert a["ay"] == a["az"] == a["vy"] == a["vz"] == 0
    s.step(1)
    a, b = s.particles[ids[0]], s.particles[ids[1]]
    assert f32(a["ax"]) == f32(0.933664382) and f32(b["ax"]) == f32(-0.933664382)
    assert f32(a["vx"]) == f32(0.0933393463)
    assert f32(a["x"]) == f32(-3.99533367)
    assert f32(a["age"]) == f32(2.0999999)
    # analytic cross-check quoted in SURVEY: 60*8/64.2^1.5
    assert abs(60 * 8 / 64.2 ** 1.5 - 0.93312) < 1e-5


@pytest.fixture(scope="module")
def g2_cloud(golden_dir):
    xyz = np.fromfile(os.path.join(golden_dir, "g2_cloud_n4096_seed12345.f32"), dtype=np.float32)
    return xyz.reshape(-1, 3)


def _g2(xyz, dt):
    s = O.System(dt=dt)
    fert = (1e6 + np.arange(len(xyz))).astype(np.float32)  # explosions off; doubles as a tag
    s.fill(xyz, age=np.float32(40 * dt), fert_age=fert)
    return s


def test_g2_lifecycle_counts_dt001(g2_cloud):
    """BASELINE config 0 (N=4096, dt=0.01, 100 steps): live / relocation counts."""
    s = _g2(g2_cloud, 0.01)
    done = 0
    for steps, live, reloc in ((1, 4090, 0), (10, 4054, 34), (100, 2724, 1537)):
        s.step(steps - done)
        done = steps
        c = s.counters
        assert s.live_count() == live
        assert c["relocations"] == reloc
    assert c["deaths_age"] + c["deaths_collision"] == 1372
    assert c["relocations_lost"] == 0 and c["cell_overflow_kills"] == 0


def test_g2_lifecycle_counts_dt005(g2_cloud):
    s = _g2(g2_cloud, 0.05)
    s.step(100)
    c = s.counters
    assert s.live_count() == 1716
    assert c["deaths_age"] + c["deaths_collision"] == 2380
    assert c["relocations"] == 11375


def test_tag_survives_relocation(g2_cloud):
    """fertility_age is carried by copy_particle, so live tags stay unique."""
    s = _g2(g2_cloud, 0.05)
    s.step(20)
    p = s.particles
    live = p[(p["cell"] >= 0)]
    assert len(np.unique(live["fertility_age"])) == len(live)
    assert (live["id"] == np.nonzero(p["cell"] >= 0)[0]).all()  # id == slot invariant
