"""GPU (-m gpu): parity of the HIP MSM path, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and — at BASELINE.json's full size — through
size-independent properties (closed form over known discrete logs, linearity in the scalars, split/merge).
Bar: bit-exact (integer arithmetic); the ABI returns the normalised Jacobian representative so equal group
elements are compared limb for limb.  Nothing here reads /root/reference."""
import ctypes as C
import os
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd._native import lib
from crypto_amd.aggregation import ops

pytestmark = pytest.mark.gpu
CUR = {"G1": (ca.G1, O.G1), "G2": (ca.G2, O.G2)}


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "GPU tests need a device"
    ca.init(0)
    yield


def normalised(G, jac):
    """oracle result -> the ABI's canonical Jacobian triple (affine, Z = one / Z = 0)"""
    a, inf = G.to_affine(jac)
    h = G.AW // 2
    z = np.zeros(h, np.uint64)
    if inf:
        one = O.fp_to_mont(np.array([[1, 0, 0, 0, 0, 0]], np.uint64)).reshape(-1)
        x = np.zeros(h, np.uint64); x[:6] = one
        return np.concatenate([x, x, z])
    z[:6] = O.fp_to_mont(np.array([[1, 0, 0, 0, 0, 0]], np.uint64)).reshape(-1)
    return np.concatenate([a, z])


def test_device_field_selftest(twin):
    a = O.fp_to_mont(O.rand_scalars(11, 900).reshape(-1, 6)[:400] & np.uint64(0x00FFFFFFFFFFFFFF))
    b = O.fp_to_mont(O.rand_scalars(12, 900).reshape(-1, 6)[:400] & np.uint64(0x00FFFFFFFFFFFFFF))
    out = np.zeros_like(a)
    rc = lib().dgpu_selftest_fp_mul(a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), len(a), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    for i in range(len(a)):
        assert U.fp_int(out[i]) == U.fp_int(a[i]) * U.fp_int(b[i]) % U.P


@pytest.mark.parametrize("name", ["g1_msm", "g2_msm"])
def test_golden_fixtures(name):
    for case in U.load(name):
        curve, G = CUR[case["group"]]
        bases, inf, sc, exp = U.case_arrays(case)
        got = ca.msm_bigint(curve, bases, sc, inf)
        assert U.jac_to_model(G, got) == exp, (name, case["n"], case.get("note"))
        if case["n"]:
            # same through msm_unchecked (Montgomery scalars) and Pairs (utils/src/pairs.rs:143-156)
            n = min(len(bases), len(sc))
            got2 = ca.msm_unchecked(curve, bases, O.fr_to_mont(sc), inf)
            assert (got2 == got).all()
            if len(bases) == len(sc):
                assert (ca.Pairs(curve, bases, sc, inf).msm_bigint() == got).all()


@pytest.mark.parametrize("gname,n", [("G1", 0), ("G1", 1), ("G1", 2), ("G1", 31), ("G1", 32), ("G1", 33), ("G1", 1 << 10), ("G1", 1 << 16),
                                     ("G2", 1), ("G2", 33), ("G2", 1 << 12)])
def test_vs_oracle_seeded(gname, n):
    curve, G = CUR[gname]
    bases, _, _ = U.seq_bases(G, n, 1000 + n, threads=16) if n else (np.zeros((0, G.AW), np.uint64), 0, 0)
    sc = O.rand_scalars(2000 + n, n)
    ref = normalised(G, G.msm(bases, sc, threads=16))
    got, got_buckets = U.on_both_paths(lambda: ca.msm_bigint(curve, bases, sc))     # (n <= 8192: the tree path, then the bucket pipeline)
    assert (got == ref).all() and (got_buckets == ref).all()


@pytest.mark.parametrize("path", ["tree", "buckets"])
def test_edge_cases_g1(path):
    """P == Q and P == -Q inside a bucket, heavy buckets, identities: on the bucket pipeline these exercise k_accumulate's special cases, the heavy-bucket
    folds and the reduction; on the tree path (what a 200-term call takes by default) the same inputs meet the general addition's"""
    import contextlib
    with (U.bucket_pipeline() if path == "buckets" else contextlib.nullcontext()):
        _edge_cases_g1()


def _edge_cases_g1():
    G, curve = O.G1, ca.G1
    n = 200
    bases, k0, d = U.seq_bases(G, n, 5)
    sc = O.rand_scalars(6, n)
    inf = np.zeros(n, np.uint8)
    sc[0] = 0                                  # zero scalar
    sc[1] = O.int_to_limbs(U.R - 1, 4)         # r - 1
    bases[3] = bases[2]                        # duplicate base (P == Q path inside a bucket when scalars match)
    sc[3] = sc[2]
    bases[5] = bases[4]; bases[5][6:] = U.fp_abi((-U.fp_int(bases[4][6:])) % U.P)   # P and -P with equal scalars
    sc[5] = sc[4]
    inf[7] = 1                                 # flagged identity base
    bases[8] = 0                               # all-zero words == identity
    sc[10:60] = O.int_to_limbs(1, 4)           # many equal tiny scalars: one heavy bucket
    sc[60:90] = O.int_to_limbs(12345, 4)
    got = ca.msm_bigint(curve, bases, sc, inf)
    inf2 = inf.copy(); inf2[8] = 1
    ref = normalised(G, G.msm(bases, sc, inf2, threads=4))
    assert (got == ref).all()
    # all scalars equal -> every term of a window in one bucket (load-balance worst case)
    sc[:] = O.int_to_limbs(0xDEADBEEFCAFEF00D1234, 4)
    got = ca.msm_bigint(curve, bases, sc, inf)
    assert (got == normalised(G, G.msm(bases, sc, inf2, threads=4))).all()
    # everything cancels -> identity, Z == 0
    b2 = np.concatenate([bases[20:40], bases[20:40]]); b2[20:, 6:] = np.stack([U.fp_abi((-U.fp_int(y)) % U.P) for y in bases[20:40, 6:]])
    s2 = np.concatenate([sc[20:40], sc[20:40]])
    got = ca.msm_bigint(curve, b2, s2)
    assert not got[12:].any()


def test_degenerate_inputs():
    G, curve = O.G1, ca.G1
    bases, _, _ = U.seq_bases(G, 64, 11)
    sc = O.rand_scalars(12, 64)
    ident = normalised(G, G.msm(bases[:0], sc[:0]))
    # every base is the identity / every scalar is zero -> identity (Z == 0), no term reaches a bucket
    assert (ca.msm_bigint(curve, bases, sc, np.ones(64, np.uint8)) == ident).all()
    assert (ca.msm_bigint(curve, np.zeros_like(bases), sc) == ident).all()
    assert (ca.msm_bigint(curve, bases, np.zeros_like(sc)) == ident).all()
    # n >= 2^31 is refused before anything is allocated
    out = np.zeros(18, np.uint64)
    rc = lib().dgpu_msm_g1(bases.ctypes.data_as(C.c_void_p), None, sc.ctypes.data_as(C.c_void_p), 1 << 31, out.ctypes.data_as(C.c_void_p))
    assert rc == -3


def test_scalars_with_bit_255():
    """A scalar >= 2^255 means different things to arkworks depending on ITS window width (oracle/oracle.c ark_make_digits reads bit 255 in the top
    window unless the width divides 255), so no width-independent result exists: every MSM entry point refuses it (DGPU_E_BADARG) and the
    caller stays on its CPU path.  Scalars in [r, 2^255) are fine: both sides multiply by the integer."""
    G, curve = O.G1, ca.G1
    for n in (31, 64, 3000):            # arkworks' c = 3 (divides 255: bit ignored), 6 and 10 (bit read)
        bases, _, _ = U.seq_bases(G, n, 400 + n)
        sc = O.rand_scalars(401 + n, n)
        hi = sc.copy(); hi[n // 2, 3] |= np.uint64(1 << 63)
        c_ark = O.window_c(n)
        ark_reads_bit = (255 % c_ark) != 0
        # what the reference would return: the oracle agrees with the masked scalar exactly when arkworks' width divides 255
        same = (G.to_affine(G.msm(bases, hi))[0] == G.to_affine(G.msm(bases, sc))[0]).all()
        assert same == (not ark_reads_bit)
        for call in (lambda s_: ca.msm_bigint(curve, bases, s_),
                     lambda s_: ca.DeviceBases(curve, bases).msm_bigint(s_),
                     lambda s_: ca.DeviceBases(curve, bases).msm_resident(ca.DeviceScalars(s_)),
                     lambda s_: ca.DeviceBases(curve, bases).precompute(16).msm_bigint(s_),
                     lambda s_: ca.msm_bigint(ca.G2, U.seq_bases(O.G2, n, 7)[0], s_) if n == 64 else ca.msm_bigint(curve, bases, s_)):
            with pytest.raises(ca.DockGpuError) as e:
                call(hi)
            assert e.value.code == -3
            call(sc)                                          # the library is usable afterwards
        tab = ca.DeviceBases(curve, bases).precompute(16)
        with pytest.raises(ca.DockGpuError) as e:
            ca.SortedScalars(tab, ca.DeviceScalars(hi), n)
        assert e.value.code == -3
    # [r, 2^255): multiplied as the integer it is, like arkworks
    bases, _, _ = U.seq_bases(G, 64, 11)
    sc = O.rand_scalars(12, 64)
    big = sc.copy(); big[:, 3] |= np.uint64(1 << 62); big[:, 3] |= np.uint64(0x3000000000000000)      # >= r, < 2^255
    assert (ca.msm_bigint(curve, bases, big) == normalised(G, G.msm(bases, big))).all()


def test_strided_bases_are_the_callers_affine_structs():
    """dgpu_msm_*_strided / dgpu_bases_upload_*_strided: ark-ec's in-memory Affine { x, y, infinity } (104 / 200 bytes) goes to the device as it is"""
    for name, n in (("G1", 3000), ("G2", 700)):
        curve, G = CUR[name]
        bases, _, _ = U.seq_bases(G, n, 90)
        sc = O.rand_scalars(91, n)
        inf = np.zeros(n, np.uint8); inf[::7] = 1
        st = ca.to_affine_structs(curve, bases, inf)
        assert st.dtype.itemsize == (104 if name == "G1" else 200)
        ref = normalised(G, G.msm(bases[inf == 0], sc[inf == 0], threads=8))
        assert (ca.msm_strided(curve, st, sc) == ref).all()
        assert (ca.msm_strided(curve, st, O.fr_to_mont(sc), montgomery=True) == ref).all()
        assert (ca.msm_bigint(curve, bases, sc, inf) == ref).all()
        db = ca.DeviceBases.from_structs(curve, st)
        assert (db.msm_bigint(sc) == ref).all()
        # another field order (the offsets are arguments): infinity first, then y, then x
        h = curve.AW // 2
        dt = np.dtype({"names": ["infinity", "y", "x"], "formats": [np.uint8, (np.uint64, h), (np.uint64, h)], "offsets": [0, 8, 8 + 8 * h], "itemsize": 16 + 16 * h})
        st2 = np.zeros(n, dt); st2["x"], st2["y"], st2["infinity"] = st["x"], st["y"], st["infinity"]
        assert (ca.msm_strided(curve, st2, sc) == ref).all()
        # misaligned offsets / a stride shorter than the fields are refused
        out = np.zeros(curve.JW, np.uint64)
        fn = curve.fn("dgpu_msm_%s_strided")
        assert fn(st.ctypes.data_as(C.c_void_p), st.dtype.itemsize, 4, 8 * h, 16 * h, sc.ctypes.data_as(C.c_void_p), n, 0, out.ctypes.data_as(C.c_void_p)) == -3
        assert fn(st.ctypes.data_as(C.c_void_p), 8 * h, 0, 8 * h, 16 * h, sc.ctypes.data_as(C.c_void_p), n, 0, out.ctypes.data_as(C.c_void_p)) == -3


def test_truncation_and_handles():
    G, curve = O.G1, ca.G1
    bases, _, _ = U.seq_bases(G, 300, 77)
    sc = O.rand_scalars(78, 200)
    full = ca.msm_bigint(curve, bases, sc)                       # min(len) like legogroth16/src/prover.rs:286
    assert (full == normalised(G, G.msm(bases[:200], sc))).all()
    db = ca.DeviceBases(curve, bases)
    assert (db.msm_bigint(sc) == full).all()
    # &query[1..] (prover.rs:592)
    assert (db.msm_bigint(sc, offset=1) == normalised(G, G.msm(bases[1:201], sc))).all()
    assert (db.msm_bigint(O.fr_to_mont(sc), montgomery=True) == full).all()
    ds = ca.DeviceScalars(sc)
    assert (db.msm_resident(ds) == full).all()
    assert (db.msm_resident(ds, n=50, base_offset=10, scalar_offset=20) == normalised(G, G.msm(bases[10:60], sc[20:70]))).all()
    with pytest.raises(ca.DockGpuError):
        db.msm_resident(ds, n=400)


@pytest.mark.parametrize("c", [7, 10, 13, 15, 16, 18])
def test_any_window_width_same_point(c, twin):
    """every window width and chunk length of the BUCKET pipeline gives the oracle's point (n = 5000 and 9001: below and above the size up to which
    calls take the tree path by default — the smaller one runs with that path switched off, else the knobs would not be exercised at all)"""
    G, curve = O.G1, ca.G1
    bases, k0, d = U.seq_bases(G, 9001, 31, threads=16)
    sc = O.rand_scalars(32, 9001)
    refs = {n: normalised(G, G.msm(bases[:n], sc[:n], threads=16)) for n in (5000, 9001)}
    lib().dgpu_set_window_bits(c)
    try:
        for ch in (16, 128):
            assert lib().dgpu_set_chunk(ch) == 0
            assert (ca.msm_bigint(curve, bases, sc) == refs[9001]).all()
            with U.bucket_pipeline():
                assert (ca.msm_bigint(curve, bases[:5000], sc[:5000]) == refs[5000]).all()
    finally:
        lib().dgpu_set_window_bits(0)
        lib().dgpu_set_chunk(0)


def test_skewed_scalar_distributions():
    # Groth16-like witnesses: mostly 0/1/small values; 16-bit scalars
    G, curve = O.G1, ca.G1
    n = 1 << 14
    bases, _, _ = U.seq_bases(G, n, 91, threads=16)
    rng = np.random.default_rng(5)
    sc = np.zeros((n, 4), np.uint64)
    kind = rng.integers(0, 4, n)
    sc[kind == 1, 0] = 1
    sc[kind == 2, 0] = rng.integers(0, 1 << 16, (kind == 2).sum(), dtype=np.uint64)
    full = O.rand_scalars(92, n)
    sc[kind == 3] = full[kind == 3]
    assert (ca.msm_bigint(curve, bases, sc) == normalised(G, G.msm(bases, sc, threads=16))).all()


def test_full_size_properties_2_20():
    """BASELINE config 2 size: closed form over known dlogs, linearity, split/merge — no oracle MSM needed."""
    G, curve = O.G1, ca.G1
    n = 1 << 20
    bases, k0, d = U.seq_bases(G, n, 7777, threads=64)
    s1 = O.rand_scalars(7779, n)
    s2 = O.rand_scalars(7780, n)
    db = ca.DeviceBases(curve, bases)
    r1 = db.msm_bigint(s1)
    assert U.jac_to_model(G, r1) == U.closed_form(G, s1, k0, d)
    # determinism: same normalised limbs on a second run (atomics reorder bucket contents, not the point)
    assert (db.msm_bigint(s1) == r1).all()
    # linearity: msm(s1) + msm(s2) == msm(s1 + s2 mod r)
    r2 = db.msm_bigint(s2)
    ssum = np.stack([O.int_to_limbs((O.limbs_to_int(a) + O.limbs_to_int(b)) % U.R, 4) for a, b in zip(s1[:4096], s2[:4096])])
    s12 = np.concatenate([ssum, s1[4096:]])         # only the first 4096 scalars are summed (python big ints), rest unchanged
    s2z = s2.copy(); s2z[4096:] = 0
    r2z = db.msm_bigint(s2z)
    lhs = G.add(r1, r2z)
    assert U.jac_to_model(G, lhs) == U.jac_to_model(G, db.msm_bigint(s12))
    # split/merge: msm over [0, n) == msm over [0, n/2) + msm over [n/2, n)
    ds = ca.DeviceScalars(s1)
    a = db.msm_resident(ds, n=n // 2)
    b = db.msm_resident(ds, n=n // 2, base_offset=n // 2, scalar_offset=n // 2)
    assert U.jac_to_model(G, G.add(a, b)) == U.jac_to_model(G, r1)
    assert U.jac_to_model(G, r2) == U.closed_form(G, s2, k0, d)


def test_g2_full_size_closed_form_2_16():
    G, curve = O.G2, ca.G2
    n = 1 << 16
    bases, k0, d = U.seq_bases(G, n, 4242, threads=64)
    sc = O.rand_scalars(4243, n)
    assert U.jac_to_model(G, ca.msm_bigint(curve, bases, sc)) == U.closed_form(G, sc, k0, d)


def test_g2_full_size_closed_form_2_20():
    """BASELINE config 3 size (G2, n = 2^20): closed form over known dlogs + split/merge through the resident handles"""
    G, curve = O.G2, ca.G2
    n = 1 << 20
    bases, k0, d = U.seq_bases(G, n, 5252, threads=64)
    sc = O.rand_scalars(5253, n)
    db = ca.DeviceBases(curve, bases); ds = ca.DeviceScalars(sc)
    r = db.msm_resident(ds)
    assert U.jac_to_model(G, r) == U.closed_form(G, sc, k0, d)
    a = db.msm_resident(ds, n=n // 4)
    b = db.msm_resident(ds, n=n - n // 4, base_offset=n // 4, scalar_offset=n // 4)
    assert U.jac_to_model(G, G.add(a, b)) == U.jac_to_model(G, r)


def test_g1_2_22_closed_form_split_and_window_independence(twin):
    """n = 2^22 (twice the per-GPU share of BASELINE config 5): bases k_i G from the fixed-base kernel with seeded k_i, so the result has a
    closed form; plus split/merge over resident handles and the same point for another window width."""
    from crypto_amd import fixed_base as fb
    n = 1 << 22
    ks = O.rand_scalars(9101, n); sc = O.rand_scalars(9102, n)
    with fb.WindowTable(ca.G1, O.G1.generator()) as t:
        db = t.multiply_many_to_bases(ks)
        to_int = lambda a: [int(w0) | (int(w1) << 64) | (int(w2) << 128) | (int(w3) << 192) for w0, w1, w2, w3 in a.tolist()]
        tot = sum(x * y for x, y in zip(to_int(ks), to_int(sc))) % U.R
        exp_xy, exp_inf = t.multiply(tot)
    ds = ca.DeviceScalars(sc)
    r = db.msm_resident(ds)
    assert not exp_inf and (r[:12] == exp_xy).all() and r[12:].any()
    a = db.msm_resident(ds, n=n // 2)
    b = db.msm_resident(ds, n=n - n // 2, base_offset=n // 2, scalar_offset=n // 2)
    assert (O.G1.to_affine(O.G1.add(a, b))[0] == r[:12]).all()
    assert lib().dgpu_set_window_bits(14) == 0
    try:
        assert (db.msm_resident(ds) == r).all()
    finally:
        lib().dgpu_set_window_bits(0)
    # the oracle agrees on the closed-form point (independent double-and-add)
    assert (O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(tot, 4)))[0] == exp_xy).all()


@pytest.mark.parametrize("path", ["tree", "buckets"])
def test_concurrent_callers_share_the_device(path):
    """the reference calls MSM from inside rayon workers (verifiable_encryption/src/tz_21/rdkgith.rs:140-147): several host
    threads in flight at once must each get their own correct result (per-call slots: stream + workspace) — on the tree path these sizes
    take by default and on the bucket pipeline (sort, accumulation, fix-up, reduction of six calls interleaved on the device)."""
    import contextlib
    with (U.bucket_pipeline() if path == "buckets" else contextlib.nullcontext()):
        _concurrent_callers()


def _concurrent_callers():
    import threading
    G, curve = O.G1, ca.G1
    sets = []
    for k in range(6):
        n = 3000 + 517 * k
        bases, _, _ = U.seq_bases(G, n, 600 + k)
        sc = O.rand_scalars(700 + k, n)
        sets.append((bases, sc, normalised(G, G.msm(bases, sc, threads=8))))
    errs = []

    def work(bases, sc, ref):
        try:
            for _ in range(4):
                if not (ca.msm_bigint(curve, bases, sc) == ref).all():
                    errs.append("mismatch")
        except Exception as e:          # noqa: BLE001
            errs.append(repr(e))
    ths = [threading.Thread(target=work, args=s) for s in sets]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("gname,logn", [("g1", 19), ("g2", 17)])
def test_witness_like_scalars_at_size_closed_form(gname, logn):
    """A Groth16-like witness at a size where its `1` bucket is long (2^16 .. 2^17 terms: thousands of chunks, several fold ranges) and the
    pair list is a fraction of n * W (the chunking is then decided on the device): closed form over known discrete logs, on the plain
    pipeline and on the table; the same with ALL scalars equal to 1 and to r - 1 (one bucket holds every term)."""
    G, curve = (O.G1, ca.G1) if gname == "g1" else (O.G2, ca.G2)
    n = 1 << logn
    bases, k0, d = U.seq_bases(G, n, 600 + logn, threads=64)
    rng = np.random.default_rng(logn)
    sc = O.rand_scalars(700 + logn, n)
    kind = rng.integers(0, 8, n)
    sc[kind <= 2] = 0                                           # 37.5 % zeros
    sc[kind == 3] = 0; sc[kind == 3, 0] = 1                     # 12.5 % ones
    m = (kind == 4) | (kind == 5); sc[m, 1:] = 0; sc[m, 0] &= np.uint64(0xFFFF)     # 25 % 16-bit
    ones = np.zeros((n, 4), np.uint64); ones[:, 0] = 1
    minus = np.tile(O.int_to_limbs(U.R - 1, 4), (n, 1))
    plain = ca.DeviceBases(curve, bases); tab = ca.DeviceBases(curve, bases).precompute(20)
    for s in (sc, ones, minus):
        want = U.closed_form(G, s, k0, d)
        assert U.jac_to_model(G, plain.msm_bigint(s)) == want
        assert U.jac_to_model(G, tab.msm_bigint(s)) == want
    plain.free(); tab.free()


@pytest.mark.parametrize("gname", ["g1", "g2"])
def test_identity_bases_from_2p17_terms_on_every_path(gname):
    """From 2^17 terms on the plain pipeline sorts with the two-level partition sort and leaves identity records to the accumulation; the table
    pipeline reads the table's byte-per-base identity flags.  One percent identity bases, duplicates and negated duplicates, a handle offset that
    is not a multiple of the sort's tile: one-shot call, plain handle, table handle against the oracle."""
    G, curve = (O.G1, ca.G1) if gname == "g1" else (O.G2, ca.G2)
    n = (1 << 17) + 77
    bases, _, _ = U.seq_bases(G, n, 9100, threads=64)
    rng = np.random.default_rng(17)
    bases[1000:2000] = bases[0:1000]                            # duplicates
    for i in range(2000, 2100):
        bases[i] = ops.neg(curve, bases[i - 2000])              # negated duplicates
    inf = (rng.integers(0, 100, n) == 0).astype(np.uint8)
    inf[0] = 1; inf[n - 1] = 1
    bases[inf == 1] = 0
    sc = O.rand_scalars(9200, n)
    want = G.to_affine(G.msm(bases, sc, inf, threads=64))
    off = 513
    want_off = G.to_affine(G.msm(bases[off:], sc[: n - off], inf[off:], threads=64))
    eq = lambda a, b: a[1] == b[1] and (a[1] or (a[0] == b[0]).all())
    assert eq(G.to_affine(ca.msm_bigint(curve, bases, sc, is_inf=inf)), want)
    db = ca.DeviceBases(curve, bases, inf)
    assert eq(G.to_affine(db.msm_bigint(sc)), want)
    assert eq(G.to_affine(db.msm_bigint(sc[: n - off], offset=off)), want_off)
    db.precompute(20)
    assert eq(G.to_affine(db.msm_bigint(sc)), want)
    assert eq(G.to_affine(db.msm_bigint(sc[: n - off], offset=off)), want_off)
    db.free()
