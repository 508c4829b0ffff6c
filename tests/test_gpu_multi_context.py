"""GPU (-m gpu): several device contexts inside ONE process behind the C ABI (SURVEY 8b `dgpu_msm_g1_sharded`, 8e): the
box has one GPU, so two contexts are opened on it (dgpu_init_device_list([0, 0]): own streams and workspaces each) —
every code path of the in-library sharding runs (per-device host threads, handles routed to their owner, host fold)
except a second physical device.  Also: handles pinned against a concurrent free, and the size threshold."""
import ctypes as C
import threading
import time
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd._native import lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _contexts():
    assert torch.cuda.is_available()
    ca.init_devices([0, 0])
    assert lib().dgpu_context_count() == 2
    yield
    lib().dgpu_set_device(0)


def _inputs(G, n, seed):
    bases, _, _ = U.seq_bases(G, n, seed, threads=16)
    return bases, O.rand_scalars(seed + 7, n)


@pytest.mark.parametrize("gname,n", [("G1", 0), ("G1", 1), ("G1", 3), ("G1", 20001), ("G1", 1 << 17), ("G2", 3001)])
def test_sharded_oneshot_equals_single_context_and_oracle(gname, n):
    curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
    bases, sc = _inputs(G, n, 300 + n) if n else (np.zeros((0, G.AW), np.uint64), np.zeros((0, 4), np.uint64))
    inf = np.zeros(n, np.uint8)
    if n > 2:
        inf[2] = 1
    single = ca.msm_bigint(curve, bases, sc, inf)
    assert (ca.msm_bigint_sharded(curve, bases, sc, inf) == single).all()
    assert (ca.msm_bigint_sharded(curve, bases, sc, inf, ngpus=1) == single).all()
    if n <= 20001:
        ref = G.msm(bases, sc, inf, threads=16)
        assert U.jac_to_model(G, single) == U.jac_to_model(G, ref)
    with pytest.raises(ca.DockGpuError):
        ca.msm_bigint_sharded(curve, bases, sc, inf, ngpus=3)


def test_sharded_resident_query_and_truncation():
    G, curve = O.G1, ca.G1
    n = 50000
    bases, sc = _inputs(G, n, 77)
    one = ca.DeviceBases(curve, bases)
    sh = ca.ShardedDeviceBases(curve, bases)
    full = one.msm_bigint(sc)
    assert (sh.msm_bigint(sc) == full).all()
    assert (sh.msm_bigint(O.fr_to_mont(sc), montgomery=True) == full).all()
    for m in (1, 24999, 25000, 25001, 49999):                 # fewer scalars than bases: the first m terms (prover.rs:286), across the shard border
        assert (sh.msm_bigint(sc[:m]) == one.msm_bigint(sc[:m])).all(), m
    ds = sh.upload_scalars(sc)
    assert (sh.msm_resident(ds) == full).all()
    ds2 = sh.upload_scalars(sc[:30000])
    assert (sh.msm_resident(ds2) == one.msm_bigint(sc[:30000])).all()
    ds.free(); ds2.free(); sh.free(); one.free()
    with pytest.raises(ca.DockGpuError):
        sh2 = ca.ShardedDeviceBases(curve, bases); h = sh2.handle; sh2.free(); sh2.handle = h; sh2.msm_bigint(sc)


def test_handles_run_on_their_owner_context():
    """a handle created on context 1 is usable from a thread whose current context is 0 (and the other way round)"""
    G, curve = O.G1, ca.G1
    bases, sc = _inputs(G, 5000, 91)
    ref = ca.msm_bigint(curve, bases, sc)
    assert lib().dgpu_set_device(1) == 0
    db1 = ca.DeviceBases(curve, bases); ds1 = ca.DeviceScalars(sc)
    assert lib().dgpu_set_device(0) == 0
    ds0 = ca.DeviceScalars(sc)
    assert (db1.msm_bigint(sc) == ref).all() and (db1.msm_resident(ds1) == ref).all()
    with pytest.raises(ca.DockGpuError):
        db1.msm_resident(ds0)                      # operands on two different contexts
    res = {}

    def other():
        lib().dgpu_set_device(1)
        res["r"] = ca.msm_bigint(curve, bases, sc)
    t = threading.Thread(target=other); t.start(); t.join()
    assert (res["r"] == ref).all()
    assert lib().dgpu_set_device(5) == -3


def test_free_waits_for_calls_in_flight():
    G, curve = O.G1, ca.G1
    bases, sc = _inputs(G, 1 << 16, 17)
    db = ca.DeviceBases(curve, bases)
    ref = db.msm_bigint(sc)
    out, errs = [], []

    def worker():
        for _ in range(30):
            try:
                out.append(db.msm_bigint(sc))
            except ca.DockGpuError as e:            # after the free: bad handle
                errs.append(e.code)
    ths = [threading.Thread(target=worker) for _ in range(3)]
    for t in ths:
        t.start()
    time.sleep(0.01)
    h = db.handle
    assert lib().dgpu_bases_free(h) == 0           # blocks until the calls using it have returned
    db.handle = 0
    for t in ths:
        t.join()
    assert out and all((o == ref).all() for o in out)
    assert all(c == -3 for c in errs)
    assert lib().dgpu_bases_free(h) == -3


def test_miller_loop_sharded_equals_single_context():
    from crypto_amd import pairing
    for n in (1, 2, 7, 300):
        ps = O.G1.gen_seq(O.rand_scalars(4, 1)[0], O.rand_scalars(5, 1)[0], n, threads=8)
        qs = O.G2.gen_seq(O.rand_scalars(6, 1)[0], O.rand_scalars(7, 1)[0], n, threads=8)
        skip = np.zeros(n, np.uint8)
        if n > 2:
            skip[1] = 1; ps[2] = 0
        one = pairing.multi_miller_loop(ps, qs, skip)
        assert (pairing.multi_miller_loop_sharded(ps, qs, skip) == one).all()
        assert (pairing.multi_miller_loop_sharded(ps, qs, skip, ngpus=1) == one).all()
    with pytest.raises(ca.DockGpuError):
        pairing.multi_miller_loop_sharded(ps, qs, None, ngpus=5)


def test_size_threshold_default_and_override():
    L = lib()
    b = O.G1.generator().reshape(1, 12); s = np.ones((1, 4), np.uint64); out = np.zeros(18, np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    try:
        L.dgpu_set_min_gpu_n(256)                  # the library default (DGPU_DEFAULT_MIN_GPU_N)
        assert L.dgpu_get_min_gpu_n() == 256
        assert L.dgpu_msm_g1(p(b), None, p(s), 1, p(out)) == -6          # DGPU_E_TOO_SMALL: the Rust shim stays on arkworks
        # bases behind a handle bring their small-path table: refused only below DGPU_MIN_GPU_N_HANDLE = 8 terms
        hb16, _, _ = U.seq_bases(O.G1, 16, 4242); hs16 = O.rand_scalars(4243, 16); h16 = C.c_uint64(0)
        assert L.dgpu_bases_upload_g1(p(hb16), None, 16, C.byref(h16)) == 0
        assert L.dgpu_msm_g1(p(hb16), None, p(hs16), 16, p(out)) == -6
        assert L.dgpu_msm_g1_handle(h16.value, 0, p(hs16), 7, 0, p(out)) == -6
        assert L.dgpu_msm_g1_handle(h16.value, 0, p(hs16), 16, 0, p(out)) == 0
        L.dgpu_set_min_gpu_n(0); ref16 = np.zeros(18, dtype=np.uint64)
        assert L.dgpu_msm_g1(p(hb16), None, p(hs16), 16, p(ref16)) == 0 and (ref16 == out).all()
        L.dgpu_set_min_gpu_n(256)
        assert L.dgpu_bases_free(h16.value) == 0
        assert L.dgpu_msm_g1_sharded(p(b), None, p(s), 1, 0, p(out)) == -6
    finally:
        L.dgpu_set_min_gpu_n(0)
    assert L.dgpu_msm_g1(p(b), None, p(s), 1, p(out)) == 0


def test_scalars_copy_range_between_contexts():
    """dgpu_scalars_copy_range: a slice of a resident scalar vector becomes a vector of its own on another context, device to device (two
    contexts on the box's one GPU: the same-device branch; between two GPUs the same call is a peer copy over xGMI).  The copy multiplies
    like the original slice; bad ranges and contexts are refused."""
    import ctypes as C
    import oracle_c as O
    import util as U
    import crypto_amd as ca
    from crypto_amd._native import lib
    ca.init_devices([0, 0])
    L = lib()
    n = 3000
    bases, _, _ = U.seq_bases(O.G1, n, 31, threads=16); sc = O.rand_scalars(32, n)
    L.dgpu_set_device(0)
    ds = ca.DeviceScalars(sc)
    h = C.c_uint64(0)
    lo, hi = 700, 2900
    assert L.dgpu_scalars_copy_range(ds.handle, lo, hi, 1, C.byref(h)) == 0
    ctx = C.c_int32(-1); ln = C.c_size_t(0)
    assert L.dgpu_handle_context(h.value, C.byref(ctx)) == 0 and ctx.value == 1
    assert L.dgpu_handle_len(h.value, C.byref(ln)) == 0 and ln.value == hi - lo
    L.dgpu_set_device(1)
    db = ca.DeviceBases(ca.G1, bases[lo:hi])                      # on context 1, next to the copy
    out = np.zeros(18, np.uint64)
    assert L.dgpu_msm_g1_resident(db.handle, 0, h.value, 0, hi - lo, out.ctypes.data_as(C.c_void_p)) == 0
    assert (out == ca.msm_bigint(ca.G1, bases[lo:hi], sc[lo:hi])).all()
    assert L.dgpu_scalars_free(h.value) == 0
    for args in ((ds.handle, 10, 5, 1), (ds.handle, 0, n + 1, 1), (ds.handle, 0, 10, 7), (db.handle, 0, 10, 1), (12345, 0, 1, 0)):
        assert L.dgpu_scalars_copy_range(*args, C.byref(h)) == -3
    assert L.dgpu_scalars_copy_range(ds.handle, 5, 5, 0, C.byref(h)) == 0 and L.dgpu_scalars_free(h.value) == 0        # an empty slice is a vector of length 0
    db.free(); ds.free()
    L.dgpu_set_device(0)


def test_unmodified_one_shot_calls_shard_themselves_when_asked():
    """dgpu_set_auto_shard_min_n: `msm_bigint(&[G1Affine], ..)` on a process that drives several devices — the one-shot entry points split the call into one
    contiguous chunk per device context (each context caching ITS chunk of the key) and fold the partial points: the single-context answer, limb for limb,
    first from host memory, then while the chunks become resident, then from the resident chunks; a sub-slice too"""
    L = lib()
    for gname, n in (("G1", 70000), ("G2", 20000)):
        curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
        bases, sc = _inputs(G, n, 900 + n)
        inf = np.zeros(n, np.uint8); inf[5] = 1
        st = ca.to_affine_structs(curve, bases, inf)
        single = ca.msm_strided(curve, st, sc)                       # (auto-sharding off: the calling thread's context)
        ref = U.jac_to_model(G, G.msm(bases, sc, inf, threads=16))
        assert U.jac_to_model(G, single) == ref
        ca.bases_cache_clear(); ca.bases_cache(min_n=1 << 12)
        assert L.dgpu_set_auto_shard_min_n(1 << 14) == 0
        try:
            s0 = ca.bases_cache_stats()
            for call in range(4):
                assert (ca.msm_strided(curve, st, sc) == single).all(), (gname, call)
                assert (ca.msm_strided(curve, st, O.fr_to_mont(sc), montgomery=True) == single).all()
                assert (ca.msm_bigint(curve, bases, sc, inf) == single).all()
            s1 = ca.bases_cache_stats()
            assert s1["fills"] - s0["fills"] == 4 and s1["hits"] - s0["hits"] >= 8, (s0, s1)      # two chunks of the structs + two of the packed arrays became resident
            sub = ca.msm_strided(curve, st[1:], sc[:n - 1])
            assert U.jac_to_model(G, sub) == U.jac_to_model(G, G.msm(bases[1:], sc[:n - 1], inf[1:], threads=16))
            assert (ca.msm_strided(curve, st[:1000], sc) == ca.msm_bigint(curve, bases[:1000], sc[:1000], inf[:1000])).all()    # below the threshold: one context
        finally:
            L.dgpu_set_auto_shard_min_n(0)
            ca.bases_cache_clear(); ca.bases_cache(min_n=1 << 16)
