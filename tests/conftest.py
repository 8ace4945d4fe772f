import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")


@pytest.fixture(scope="session")
def built():
    """Build the CPU-side helpers (oracle + generator) and make sure libspx.so exists."""
    import __graft_entry__ as ge
    ge.build_cpu_helpers()
    from secphase_amd import api
    if not os.path.exists(api.LIB_PATH):
        ge.build()
    return True
