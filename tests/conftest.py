import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)
# the torch restatements of the reference (second opinion) live beside their fixture generators; only CPU tests
# (test_oracle_vs_torch.py, test_oracle_kat.py) import them, no -m gpu test does
GOLDEN = os.path.join(TESTS, "golden")
if GOLDEN not in sys.path:
    sys.path.append(GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hip():
    """The product C-ABI library; GPU tests fail loudly if it is not built."""
    from l4dc_mpc_ocd_amd import abi
    return abi.load_hip_library()
