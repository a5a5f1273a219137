import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library normally travels prebuilt (in-tree .so); if it is absent -- a checkout without build
    artefacts -- build it once here (hipcc cross-compiles gfx950 without a GPU).  Never a fallback path: the tests
    still fail loudly if the build fails."""
    from gaussianvi_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_lib()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
