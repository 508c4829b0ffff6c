"""Run by tests/test_gpu_fault_paths.py in a process of its own, with DGPU_LIB = crypto_amd/libdock_gpu_dev.so (`make -C crypto_amd/csrc dev`: the
product's objects + dock_core.hip built -DDGPU_DEV, i.e. with the allocation-failure hook dgpu_dev_fail_alloc_after).

For every entry-point family: the library is brought up cold (dgpu_shutdown + dgpu_init: every slot's workspace is empty again), the k-th device
allocation of the workload is made to fail for k = 0, 1, 2, ... until the workload no longer notices, and each time
  * the call answers DGPU_E_OOM (or DGPU_E_HIP where the failing allocation sits behind a HIP check) — no crash, no hang, no wrong result;
  * the SAME workload then succeeds with the reference result (the library stays usable, handles of the failed call are not leaked into
    a state that breaks the next one);
then the same with six host threads in flight, and at the end the whole set of fault cycles runs a second time: the device memory still held after
dgpu_shutdown must not shrink from the first pass to the second (a leak repeats; the HIP runtime's own pools do not).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import oracle_c as O  # noqa: E402
import util as U  # noqa: E402
import lego_setup as LS  # noqa: E402
import crypto_amd as ca  # noqa: E402
from crypto_amd import pairing, qap, fixed_base as FB, legogroth16 as LG  # noqa: E402
from crypto_amd._native import lib, DockGpuError  # noqa: E402

L = lib()
L.dgpu_dev_fail_alloc_after.argtypes = [C.c_int64, C.c_int64]
fail_after = lambda k, count=2: L.dgpu_dev_fail_alloc_after(k, count)      # two in a row: Buf::ensure retries a failed allocation once


def cold():
    assert L.dgpu_shutdown() == 0
    ca.init(0)


# ---- inputs (host side, made once) ----
b1s, _, _ = U.seq_bases(O.G1, 3000, 11, threads=32); s_s = O.rand_scalars(12, 3000)
b1l, _, _ = U.seq_bases(O.G1, 40000, 13, threads=32); s_l = O.rand_scalars(14, 40000)
b2, _, _ = U.seq_bases(O.G2, 2000, 15, threads=32)
P64, _, _ = U.seq_bases(O.G1, 64, 16, threads=8); Q64, _, _ = U.seq_bases(O.G2, 64, 17, threads=8)
cs = LS.circuit(60, x0=5)
key = LS.setup(cs, 2, seed=321)
zl = LS.scalars(cs["z"])


def w_small():
    return ca.msm_bigint(ca.G1, b1s, s_s)


def w_buckets():
    return ca.msm_bigint(ca.G1, b1l, s_l)


def w_g2():
    return ca.msm_bigint(ca.G2, b2, s_s[:2000])


def w_resident():
    db = ca.DeviceBases(ca.G1, b1l)
    try:
        db.precompute(16)
        ds = ca.DeviceScalars(s_l)
        try:
            return db.msm_resident(ds)
        finally:
            ds.free()
    finally:
        db.free()


def w_small_resident():
    """a plain handle of 3000 bases met three times by the small path: the per-call table, the build of the handle's own table (an allocation that may
    fail: the call then keeps building its eight multiples itself), the tree over that table"""
    db = ca.DeviceBases(ca.G1, b1s)
    try:
        return np.concatenate([db.msm_bigint(s_s) for _ in range(3)])
    finally:
        db.free()


def w_miller():
    return np.concatenate([ca.multi_miller_loop(P64, Q64), pairing.G2Prepared.from_affine(Q64[:5]).coeffs.reshape(-1)[:72]])


def w_witness_map():
    dr = qap.DeviceR1cs(*[qap.csr(cs[k]) for k in "ABC"], len(cs["z"]), cs["n_inst"], len(cs["A"]))
    try:
        h, _ = dr.witness_map(zl)
        return h.reshape(-1)
    finally:
        dr.free()


def w_prove():
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], 2)
    pk = LG.ProvingKey(vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"], key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    dr = qap.DeviceR1cs(*[qap.csr(cs[k]) for k in "ABC"], len(cs["z"]), cs["n_inst"], len(cs["A"]))
    try:
        pr = LG.prove_abi(pk, 12345, 67890, 4242, zl, cs["n_inst"], circuit=dr)
        return np.concatenate([pr[k].reshape(-1) for k in ("a", "b", "c", "d")])
    finally:
        dr.free()
        for q in (pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query, pk.l_query):
            try:
                q.free()
            except Exception:
                pass


def w_fixed_base():
    with FB.WindowTable(ca.G1, O.G1.generator()) as t:
        pts, _ = t.multiply_many(s_s[:500])
        return pts.reshape(-1)


WORK = {"msm one-shot, tree path": w_small, "msm one-shot, bucket pipeline": w_buckets, "msm G2 one-shot": w_g2, "upload + table + resident MSM": w_resident, "small resident handle (its own table)": w_small_resident,
        "Miller loop + G2Prepared": w_miller, "witness map": w_witness_map, "LegoGroth16 prove": w_prove, "fixed-base table": w_fixed_base}


def free_bytes():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def run_all(inject):
    out = {}
    for name, fn in WORK.items():
        cold()
        a0 = ca.device_alloc_count()
        ref = fn()
        n_alloc = int(ca.device_alloc_count() - a0)               # device allocations of the workload on a cold library
        if not inject:
            continue
        codes, swallowed = [], 0
        ks = list(range(n_alloc)) if n_alloc <= 48 else sorted(set(list(range(24)) + list(range(24, n_alloc, max(1, (n_alloc - 24) // 24)))))
        for k in ks:
            cold()
            fail_after(k)
            try:
                got = fn()
                failed = None
            except DockGpuError as e:
                failed = e.code
            finally:
                fail_after(-1)
            if failed is None:
                # (reservations of the idle slots are best effort by design: a failure there is not the call's error)
                assert (got == ref).all(), (name, k, "a call that reported success returned a different result")
                swallowed += 1
            else:
                assert failed in (-2, -4), (name, k, failed)
                codes.append(failed)
                assert (fn() == ref).all(), (name, k, "library not usable after a failed call")
        assert codes, (name, "no allocation failed: the hook is not wired")
        cold(); fail_after(0, 1)                # ONE failing allocation is absorbed by the retry with the exact size wherever a workspace grows
        try:
            single = fn()
            assert (single == ref).all(), (name, "single failure")
        except DockGpuError as e:
            assert e.code in (-2, -4), (name, e.code)      # (handle allocations have no retry)
        finally:
            fail_after(-1)
        out[name] = {"allocations_on_a_cold_library": n_alloc, "failing_allocations_tried": len(ks), "answered_with_an_error": len(codes), "absorbed_by_best_effort_reservations": swallowed, "codes": sorted(set(codes))}
    return out


def in_flight():
    """six host threads, allocations failing underneath them; every call answers OK (right result) or an allocation error"""
    cold()
    refs = {n: f() for n, f in WORK.items() if n in ("msm one-shot, tree path", "msm G2 one-shot", "Miller loop + G2Prepared", "msm one-shot, bucket pipeline")}
    bad = []
    seen = {"ok": 0, "failed": 0}
    for k in (0, 1, 2, 3, 5, 8, 13):
        cold()
        fail_after(k)

        def worker(name):
            try:
                r = WORK[name]()
                if not (r == refs[name]).all():
                    bad.append((k, name, "wrong result"))
                seen["ok"] += 1
            except DockGpuError as e:
                seen["failed"] += 1
                if e.code not in (-2, -4):
                    bad.append((k, name, e.code))
        th = [threading.Thread(target=worker, args=(n,)) for n in list(refs) + list(refs)[:2]]
        [t.start() for t in th]; [t.join() for t in th]
        fail_after(-1)
        for name in refs:
            assert (WORK[name]() == refs[name]).all(), (k, name, "not usable afterwards")
    assert not bad, bad
    assert seen["failed"] > 0
    return seen


def cache_gives_way():
    """the resident-bases cache holds memory nobody asked for by name: when an allocation fails it releases its least recently used entry and the allocation
    is tried again — an explicit upload does not fail because cached keys are in the way (dock_core.hip dev_malloc, bases_cache.hpp cache_release_lru)"""
    cold()
    ca.bases_cache_clear(); ca.bases_cache(min_n=1 << 12)
    st = ca.to_affine_structs(ca.G1, b1l)
    ref = ca.msm_strided(ca.G1, st, s_l); ca.msm_strided(ca.G1, st, s_l); ca.msm_strided(ca.G1, st, s_l)
    s0 = ca.bases_cache_stats()
    assert s0["entries"] == 1 and s0["fills"] >= 1, s0
    fail_after(0, 1)                       # the next allocation fails ONCE: the cache gives its entry up, the retry succeeds
    try:
        db = ca.DeviceBases(ca.G1, b1s)
    finally:
        fail_after(-1)
    s1 = ca.bases_cache_stats()
    assert s1["entries"] == 0 and s1["evictions"] == s0["evictions"] + 1 and s1["bytes"] == 0, (s0, s1)
    assert (db.msm_bigint(s_s) == ca.msm_bigint(ca.G1, b1s, s_s)).all()
    db.free()
    # ... and the slice is simply noted and made resident again by the calls that follow
    for _ in range(3):
        assert (ca.msm_strided(ca.G1, st, s_l) == ref).all()
    assert ca.bases_cache_stats()["entries"] == 1
    # with nothing left to release a failing allocation is still answered with an error
    ca.bases_cache_clear()
    fail_after(0, 2)
    try:
        ca.DeviceBases(ca.G1, b1s)
        raise AssertionError("an allocation that failed twice with an empty cache succeeded")
    except DockGpuError as e:
        assert e.code in (-2, -4), e.code
    finally:
        fail_after(-1)
    ca.bases_cache(min_n=1 << 16)
    return {"evicted_on_a_failed_allocation": 1}


if __name__ == "__main__":
    assert torch.cuda.is_available()
    ca.init(0)
    run_all(False); L.dgpu_shutdown()
    base0 = free_bytes()
    ca.init(0); run_all(False); L.dgpu_shutdown()
    base1 = free_bytes()
    ca.init(0)
    res = run_all(True)
    fl = in_flight()
    gw = cache_gives_way()
    L.dgpu_shutdown()
    after = free_bytes()
    # A leak of the library repeats with every failed call; what the HIP runtime keeps back from its own pools after an unusual allocation
    # pattern does not (measured: 240 MB once, at the first absorbed failure of the table workload, with round 3's library too).  So the fault
    # cycles run twice and the SECOND pass is the one that must not cost memory.
    ca.init(0)
    run_all(True)
    L.dgpu_shutdown()
    again = free_bytes()
    print(json.dumps({"per_workload": res, "six_in_flight": fl, "cache_gives_way": gw, "free_bytes_after_clean_cycles": [base0, base1], "free_bytes_after_fault_cycles": [after, again],
                      "kept_by_the_runtime_after_the_first_fault_pass": base1 - after, "leaked_bytes_per_fault_pass": after - again}))
