"""No device allocation on an MSM path in steady state: the slots' workspaces are sized when a handle is uploaded /
converted to a table / by dgpu_reserve_*, and dgpu_device_alloc_count must not move afterwards,
whichever slot a call lands on and however many host threads call at once."""
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import pytest
import torch
import oracle_c as O
import crypto_amd as ca
import util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "GPU tests need a device"
    ca.init(0)
    yield


def _hammer(fn, calls=24, threads=8):
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda _: fn(), range(calls)))


@pytest.mark.parametrize("name,n", [("G1", 1 << 16), ("G2", 1 << 13)])
def test_no_allocation_after_upload_and_precompute(name, n):
    curve, G = (ca.G1, O.G1) if name == "G1" else (ca.G2, O.G2)
    bases, _, _ = U.seq_bases(G, n, 5)
    sc = O.rand_scalars(6, n)
    db = ca.DeviceBases(curve, bases)            # sizes every slot for MSMs over the handle
    ds = ca.DeviceScalars(sc)
    a0 = ca.device_alloc_count()
    ref = db.msm_resident(ds)
    outs = _hammer(lambda: db.msm_resident(ds)) + _hammer(lambda: db.msm_bigint(sc)) + _hammer(lambda: db.msm_bigint(sc, offset=1))[:0]
    assert all((o == ref).all() for o in outs)
    assert ca.device_alloc_count() == a0, "a plain-handle MSM allocated device memory"
    db.precompute(16)
    a1 = ca.device_alloc_count()
    assert a1 > a0                               # the table itself (and the slots' table workspaces)
    outs = _hammer(lambda: db.msm_resident(ds)) + _hammer(lambda: db.msm_bigint(sc))
    assert all((o == ref).all() for o in outs)
    assert ca.device_alloc_count() == a1, "a table MSM allocated device memory"


def test_no_allocation_on_one_shot_calls_after_reserve():
    n = 50000
    bases, _, _ = U.seq_bases(O.G1, n, 8)
    sc = O.rand_scalars(9, n)
    st = ca.to_affine_structs(ca.G1, bases)
    ca.reserve(ca.G1, n)
    a0 = ca.device_alloc_count()
    ref = ca.msm_bigint(ca.G1, bases, sc)
    outs = _hammer(lambda: ca.msm_bigint(ca.G1, bases, sc)) + _hammer(lambda: ca.msm_strided(ca.G1, st, sc)) + _hammer(lambda: ca.msm_bigint(ca.G1, bases[:n // 2], sc))[:0]
    assert all((o == ref).all() for o in outs)
    assert ca.device_alloc_count() == a0, "a one-shot MSM allocated device memory after dgpu_reserve_g1"


def test_first_one_shot_call_of_a_new_size_sizes_the_idle_slots():
    n = 70001                                       # a size nothing before this test reserved
    ca.bases_cache(bytes=0)                         # this test is about the ONE-SHOT path: without this the second call would make the bases resident (tests/test_gpu_bases_cache.py)
    bases, _, _ = U.seq_bases(O.G2, n, 10)
    sc = O.rand_scalars(11, n)
    ref = ca.msm_bigint(ca.G2, bases, sc)           # grows its slot and every idle one
    a0 = ca.device_alloc_count()
    try:
        outs = _hammer(lambda: ca.msm_bigint(ca.G2, bases, sc), calls=12, threads=6)
    finally:
        ca.bases_cache(bytes=(1 << 64) - 1)         # DGPU_CACHE_BYTES_AUTO
    assert all((o == ref).all() for o in outs)
    assert ca.device_alloc_count() == a0


def test_callers_beyond_the_slots_are_served_in_turn():
    """More host threads than the library keeps calls in flight (six slots per context): the waiting callers queue first come, first served.  Twelve threads,
    240 resident MSMs of 2^17 terms: no single call may last a large part of the whole run (with the try-lock hand-out of rounds 1 - 6 the threads beyond the
    sixth waited until the others had nothing left to do: the longest call WAS the run), and every result is the same point."""
    import threading
    import time
    n = 1 << 17
    bases, _, _ = U.seq_bases(O.G1, n, 11)
    sc = O.rand_scalars(12, n)
    db = ca.DeviceBases(ca.G1, bases); db.precompute(16)
    ds = ca.DeviceScalars(sc)
    ref = db.msm_resident(ds)
    T, COUNT = 12, 240
    for _ in range(2 * T):
        db.msm_resident(ds)
    nxt = iter(range(COUNT)); lock = threading.Lock()
    longest, bad = [0.0] * T, [0] * T
    start = threading.Barrier(T + 1)

    def run(t):
        start.wait()
        while True:
            with lock:
                i = next(nxt, None)
            if i is None:
                return
            t0 = time.perf_counter()
            r = db.msm_resident(ds)
            longest[t] = max(longest[t], time.perf_counter() - t0)
            bad[t] += int(not (r == ref).all())
    th = [threading.Thread(target=run, args=(t,)) for t in range(T)]
    for x in th:
        x.start()
    t0 = time.perf_counter(); start.wait()
    for x in th:
        x.join()
    total = time.perf_counter() - t0
    assert sum(bad) == 0
    assert max(longest) < 0.35 * total, (max(longest), total)      # first come, first served: about T / COUNT of the run (0.05), with slack for the interpreter
    db.free(); ds.free()
