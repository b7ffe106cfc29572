"""Births draw from a counter-based generator (splitmix64 of seed, step, parent id: k_apply /
k_replay_commit, restated in tests/util.py and fed to the oracle, against which the GPU is
exact).  The reference draws from std::random_device (ps.cpp:29-56), which nobody can
reproduce; what can be checked is that the draws follow the reference's DISTRIBUTIONS:
three integers uniform on [-50, 49] (get_random_uvector_h, ps.cpp:38-56) and a fertility age
uniform on [MIN_FERTILITY_AGE, MAX_FERTILITY_AGE) (get_random_number_h, ps.cpp:29-36)."""
import numpy as np
from scipy import stats

from util import explosion_rng


def test_explosion_draws_follow_the_reference_distributions():
    rng = explosion_rng(seed=1)
    ints, us = [], []
    for step in range(40):
        for pid in range(0, 3000000, 6007):
            (a, b, c), u = rng(pid, step)
            ints.append((a, b, c)); us.append(u)
    ints = np.array(ints)
    us = np.array(us)
    assert ints.min() == -50 and ints.max() == 49 and 0.0 <= us.min() and us.max() < 1.0
    for axis in range(3):                                     # each component uniform on its 100 values
        counts = np.bincount(ints[:, axis] + 50, minlength=100)
        assert stats.chisquare(counts).pvalue > 1e-3, axis
    assert stats.kstest(us, "uniform").pvalue > 1e-3          # fertility = lo + u * (hi - lo)
    # components independent of each other and of the fertility draw
    for x, y in ((0, 1), (0, 2), (1, 2)):
        assert abs(np.corrcoef(ints[:, x], ints[:, y])[0, 1]) < 0.02
    assert abs(np.corrcoef(ints[:, 0], us)[0, 1]) < 0.02
    # consecutive parents / steps are not correlated
    assert abs(np.corrcoef(us[:-1], us[1:])[0, 1]) < 0.02
