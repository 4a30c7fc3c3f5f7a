import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library must exist for every test session (built in-tree by __graft_entry__.build())."""
    from bayeformers_amd import _C

    if not os.path.exists(_C.LIB_PATH):
        from bayeformers_amd.build import build

        build(verbose=False)
    return _C.LIB_PATH
