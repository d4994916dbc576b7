import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure the HIP library and the oracle are built (no-op when up to date)."""
    graft.build()
    return True


@pytest.fixture(scope="session")
def orc(built):
    return graft.load_oracle()


@pytest.fixture(scope="session")
def pkg(built):
    return graft.load_package()


GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR
