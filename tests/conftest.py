import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def mm():
    return ge.load()


@pytest.fixture(scope="session")
def po():
    """CPU oracle binding (test infrastructure)."""
    m = ge.load_oracle()
    m.lib()
    return m


@pytest.fixture(scope="session")
def synth(mm):
    from map_merge_amd import synth as s
    return s


@pytest.fixture(scope="session")
def ctx(mm):
    c = mm.Context(0)   # raises when no GPU / no libmm3d.so: there is no CPU fallback to hide behind
    yield c
    c.close()
