import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# the parity tests also walk the branches behind the library's development options (column-by-column separators, the
# lane-per-task kernels at sizes where they are not the default, the distributed factorization, injected failures):
# those options are refused unless the process says it is a development run (csrc/plan.h, include/slampp_hip.h)
os.environ.setdefault("SLAMPP_HIP_DEV", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Builds the product library and the oracle once per session (no-op when up to date)."""
    import __graft_entry__ as ge
    ge.build()
    return True
