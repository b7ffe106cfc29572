import os
import sys

import pytest

# PyTorch-ROCm bundles its own HIP runtime.  Load it first so that libpsamd.so (linked
# against libamdhip64 by SONAME) binds to the same copy: two HIP runtimes in one process
# do not both see the device.  Only the stream-sharing test needs torch on the GPU.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
