import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) device")
    # torch brings its own copy of the HIP runtime: when libspx's runtime initialises first, torch later finds "no HIP
    # GPUs" in the same process.  Tests that hand torch tensors to libspx (decision records for the RCCL gather) need
    # both, so torch goes first -- as in bench.py.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
    try:
        import torch
        torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        pass


@pytest.fixture(scope="session")
def built():
    """Build the CPU-side helpers (oracle + generator) and make sure libspx.so exists."""
    import __graft_entry__ as ge
    ge.build_cpu_helpers()
    from secphase_amd import api
    if not os.path.exists(api.LIB_PATH):
        ge.build()
    return True
