import os
import sys

import pytest

# The GPU tests use torch for caller-owned device memory and for torch.distributed.  torch ships its own copies of the
# ROCm runtime libraries; they only find the GPU when they are loaded BEFORE the system copies the engine's library
# binds (measured: engine first -> torch reports "No HIP GPUs are available"; torch first -> both work, which is also
# the order bench.py uses).  So the test process imports torch before anything loads libtrajopt_hipcc.so.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Native pieces are built beforehand by __graft_entry__.build(); on the GPU box the prebuilt files are used."""
    need = [os.path.join(ROOT, "oracle", "liboracle.so"), os.path.join(ROOT, "trajopt_amd", "libtrajopt_hipcc.so"),
            os.path.join(ROOT, "tests", "emu", "libemu.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()
    return True
