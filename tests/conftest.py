import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/libpats_oracle.so) - the CHECKER, never the thing under test in
    the -m gpu parity tests."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import pats_oracle
    pats_oracle.lib()
    return pats_oracle
