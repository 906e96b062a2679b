"""pytest configuration: registers the `gpu` marker and puts the host-side mirror modules
(super-resolution-system_amd/) and the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "super-resolution-system_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


# The marched gather (csrc/sr_march.inc) is taken from 90 MP of canvas up (below that its four launches lose to the block
# kernel alone); the suite's small canvases march all the same, so that both forms of the gather stay under test -- tests that
# compare the two switch SR_MARCH themselves (it is read when a plan is made).
os.environ.setdefault("SR_MARCH", "2")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """Process-wide device context; GPU tests fail loudly (no skip) when the HIP path is unavailable."""
    import _native
    return _native.default_context(0)


@pytest.fixture(scope="session")
def rng():
    import numpy as np
    return np.random.default_rng(20260313)
