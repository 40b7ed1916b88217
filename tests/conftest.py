import os
import sys

# The in-process tensor-parallel test group runs up to 8 ranks as host threads of ONE process; every rank's gather kernel
# polls for its peers on the device, so each rank's stream needs its own hardware queue (the HIP runtime's default is 4 queues
# per process, onto which streams are multiplexed: a polling kernel would then block the peer queued behind it).  Must be set
# before the HIP runtime initialises.  Production runs one process per GPU and does not need this.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c
