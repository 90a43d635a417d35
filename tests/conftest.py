import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    o.lib()
    return o


@pytest.fixture(scope="session")
def mcx():
    import __graft_entry__
    __graft_entry__.build()
    import mccortex_amd
    mccortex_amd.lib()
    return mccortex_amd
