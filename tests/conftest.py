import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import oracle
    oracle.load()
    return oracle


@pytest.fixture(scope="session")
def engine():
    """BakeEngine initialised on device 0; fails loudly when the HIP library is missing."""
    from aobaker_amd import BakeEngine
    BakeEngine.Init(0)
    yield BakeEngine
    BakeEngine.Options = None
    BakeEngine.Dispose()
