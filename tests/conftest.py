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
