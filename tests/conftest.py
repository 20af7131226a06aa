import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# The library's default (MRK_RANK_JIT=auto) never makes a request wait for the compiler: a model whose specialised kernels
# are not on disk is ranked by the interpreting kernels until a background compile finishes.  The parity tests want to know
# WHICH kernel they cover, so the suite runs with "1" (the first rank of a model waits for its kernel); the tests of the
# default itself (test_rank_one_gpu.py::test_default_jit_mode_*) set the variable themselves.
os.environ.setdefault("MRK_RANK_JIT", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ctx():
    """One mrk_ctx for the whole GPU session.  Fails loudly if the HIP library or device is missing."""
    import metarank_amd as M

    c = M.Context(0)
    yield c
    c.close()
