import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The parity tests force every tile / split variant through the library's schedule knobs, which are honoured only under MVAE_TUNING=1
# (a production process ignores stray MVAE_* variables; tests/test_host_logic.py checks that gate itself).
os.environ["MVAE_TUNING"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fresh_spin_failure_bookkeeping():
    """ops counts launches with bounded spins that gave up and switches those schedules off for the process after a few: tests that provoke
    failures must not leak that state into the next test."""
    from molecular_vae_amd import ops
    ops.PERSIST_STATS.update(failures=0, reruns=0, disabled=False)
    ops._PERSIST_WARNED[0] = False
    del ops._PERSIST_PENDING[:]
    yield
