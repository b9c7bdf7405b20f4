import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    # Some GPU tests share the process with PyTorch, which ships its own libamdhip64: whichever HIP runtime is loaded
    # first serves both (same soname), loading torch's second leaves it without devices.  Load torch's first.
    import torch  # noqa: F401
    from bftkv_amd import Context
    ctx = Context(0)
    # every batched verify call of the suite is also made through the small-call route and must agree (bftkv_amd/_native.py)
    ctx.check_small = True
    yield ctx
    ctx.close()
