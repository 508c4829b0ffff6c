"""GPU (-m gpu): allocation failures on every entry-point family (tests/fault_driver.py, run in a process of its own against the development
library crypto_amd/libdock_gpu_dev.so, which has the hook dgpu_dev_fail_alloc_after; the product library has no such symbol): each failing
allocation is answered with DGPU_E_OOM / DGPU_E_HIP, the library stays usable, six calls in flight survive failures underneath them, and the
device memory held after dgpu_shutdown matches a run without injected failures."""
import json
import os
import subprocess
import sys
import pytest
import torch  # noqa: F401  (before the library: torch ships its own HIP runtime, and a process must not end up with two of them)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = os.path.join(ROOT, "crypto_amd", "libdock_gpu_dev.so")


def test_product_library_has_no_fault_hook():
    import ctypes
    from crypto_amd._native import lib
    with pytest.raises(AttributeError):
        lib().dgpu_dev_fail_alloc_after
    del ctypes


def test_allocation_failures_are_answered_cleanly():
    assert os.path.exists(DEV), "crypto_amd/libdock_gpu_dev.so is built together with the product library (make -C crypto_amd/csrc)"
    env = dict(os.environ, DGPU_LIB=DEV)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fault_driver.py")], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert len(rep["per_workload"]) == 9 and all(v["answered_with_an_error"] >= 1 for v in rep["per_workload"].values())
    assert rep["six_in_flight"]["failed"] > 0 and rep["six_in_flight"]["ok"] > 0
    assert rep["cache_gives_way"]["evicted_on_a_failed_allocation"] == 1        # the resident-bases cache releases its entries when the device is full
    # nothing leaked: a second pass over all the fault cycles leaves the device where the first left it (the first may differ from a clean cycle by
    # what the HIP runtime keeps in its own pools: a one-time 240 MB after the first absorbed failure of the table workload, not ours)
    assert abs(rep["leaked_bytes_per_fault_pass"]) <= 64 << 20, rep
    assert rep["kept_by_the_runtime_after_the_first_fault_pass"] <= 1 << 30, rep
    print(json.dumps(rep))
