import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# torch first, in every test process: it ships its own HIP runtime; a process that loaded libdock_gpu.so (the system runtime) BEFORE torch ends
# up with two runtimes, and whichever initialises second finds no device (seen as DGPU_E_NODEVICE from dgpu_init after an unrelated test)
try:
    import torch  # noqa: F401
except Exception:  # noqa: BLE001
    pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


import pytest


@pytest.fixture
def twin():
    """The test body runs against the DEVELOPMENT twin libdock_gpu_dev.so (include/dock_gpu_dev.h: tuning knobs, stage timers, self-test hooks) —
    the same objects as the product library plus that surface, which the product does not export.  Everything the test creates lives in the twin."""
    import crypto_amd as ca
    with ca.twin() as T:
        yield T
