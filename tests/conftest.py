import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def pytest_sessionstart(session):
    """A checkout without the built library (the .so is git-ignored): build it once, in-tree, before any test needs it."""
    lib = os.path.join(ROOT, "uda_clr_amd", "lib", "libuda_clr_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__
        __graft_entry__.build()
