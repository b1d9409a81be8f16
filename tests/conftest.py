import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ai-video-detector_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure): oracle/avd_oracle.c through ctypes."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session (fails loudly if the extension is missing)."""
    import avd_hip
    avd_hip.build()                 # no-op when the in-tree .so is up to date (hipcc is on the GPU box too)
    c = avd_hip.Context(0)
    yield c
    c.close()
