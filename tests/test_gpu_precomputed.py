"""GPU (-m gpu): precomputed-multiples tables for resident bases (dgpu_bases_precompute_*): every MSM on the converted handle must return
the limbs the plain pipeline returns (and the oracle's point) — all table window widths, offsets (`&query[1..]`), sub-ranges, Montgomery
scalars, identity bases, edge / skewed / all-equal scalars, G2, sharded handles, and BASELINE config 2's size through a closed form."""
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd._native import lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


@pytest.mark.parametrize("gname,n,c", [("G1", 1, 16), ("G1", 700, 16), ("G1", 700, 20), ("G1", 5003, 18), ("G1", 5003, 22), ("G1", (1 << 16) + 3, 0), ("G1", (1 << 16) + 3, 20),
                                       ("G2", 3001, 16), ("G2", 3001, 20)])
def test_table_msm_equals_plain_and_oracle(gname, n, c):
    curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
    bases, _, _ = U.seq_bases(G, n, 1200 + n, threads=16)
    sc = O.rand_scalars(1300 + n, n)
    inf = np.zeros(n, np.uint8)
    if n > 10:
        inf[3] = 1; bases[7] = 0                                # flagged and all-zero identity bases
        sc[0] = 0; sc[1] = O.int_to_limbs(U.R - 1, 4); sc[2] = O.int_to_limbs(1, 4)
        sc[20:60] = O.int_to_limbs(0xABCDEF, 4)                 # many equal scalars
    plain = ca.DeviceBases(curve, bases, inf)
    tab = ca.DeviceBases(curve, bases, inf).precompute(c)
    ref = plain.msm_bigint(sc)
    assert (tab.msm_bigint(sc) == ref).all()
    if n <= 5003:
        inf2 = inf.copy(); inf2[7:8] = 1 if n > 10 else inf2[7:8]
        assert U.jac_to_model(G, ref) == U.jac_to_model(G, G.msm(bases, sc, inf2, threads=16))
    assert (tab.msm_bigint(O.fr_to_mont(sc), montgomery=True) == ref).all()
    if n > 10:
        assert (tab.msm_bigint(sc[:n - 1], offset=1) == plain.msm_bigint(sc[:n - 1], offset=1)).all()        # &query[1..]
        ds = ca.DeviceScalars(sc)
        assert (tab.msm_resident(ds, n=n // 3, base_offset=5, scalar_offset=9) == plain.msm_resident(ds, n=n // 3, base_offset=5, scalar_offset=9)).all()
        assert (tab.msm_resident(ds) == ref).all()
        ds.free()
    tab.precompute(c)                                           # idempotent
    assert (tab.msm_bigint(sc) == ref).all()
    plain.free(); tab.free()


@pytest.mark.parametrize("gname,n,c", [("G1", 5003, 16), ("G1", (1 << 16) + 3, 20), ("G2", 3001, 20)])
def test_any_reduction_geometry_same_point(gname, n, c, twin):
    """dgpu_set_reduce_shift: the bucket reduction's serial share (2^shift buckets per lane, 64 lanes per group, <= 64 groups per
    pseudo-window, the rest on the host) is a tuning knob — every geometry must return the limbs of the automatic one (measured at
    2^20 terms: the automatic 8 buckets per lane is the fastest both for one call and for six in flight) — and so is the form of the
    last reduction kernel (one lane per point, or four members per point multiplying one operand pair per round)"""
    curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
    bases, _, _ = U.seq_bases(G, n, 1700 + n, threads=16)
    sc = O.rand_scalars(1800 + n, n)
    tab = ca.DeviceBases(curve, bases).precompute(c)
    plain = ca.DeviceBases(curve, bases)
    ref = tab.msm_bigint(sc)
    assert (plain.msm_bigint(sc) == ref).all()
    sparse = sc.copy(); sparse[5:] = 0; sparse[:5, 1:] = 0          # a handful of filled buckets: neighbouring suffix sums are EQUAL points (the doubling branch of the additions)
    ref_sparse = plain.msm_bigint(sparse)
    try:
        for lanes in (0, 2, 1, 4):                                     # dgpu_set_reduce_lanes: 0 = bit marginals (reduce_kernels.hip.h, the default; 2: their class folds with one lane per value); k_reduce_top / k_reduce_top_quad (four members per point)
            assert lib().dgpu_set_reduce_lanes(lanes) == 0
            for sh in (-1, 0, 1, 2, 3, 4, 5, 6):
                assert lib().dgpu_set_reduce_shift(sh) == 0
                assert (tab.msm_bigint(sc) == ref).all(), (lanes, sh)
            lib().dgpu_set_reduce_shift(-1)
            assert (plain.msm_bigint(sc) == ref).all(), lanes
            assert (tab.msm_bigint(sparse) == ref_sparse).all() and (plain.msm_bigint(sparse) == ref_sparse).all(), lanes
        assert lib().dgpu_set_reduce_shift(7) != 0 and lib().dgpu_set_reduce_lanes(3) != 0
    finally:
        lib().dgpu_set_reduce_shift(-1); lib().dgpu_set_reduce_lanes(0)


def test_skewed_and_degenerate_scalars_on_a_table():
    G, curve = O.G1, ca.G1
    n = 1 << 14
    bases, _, _ = U.seq_bases(G, n, 91, threads=16)
    plain = ca.DeviceBases(curve, bases); tab = ca.DeviceBases(curve, bases).precompute(20)
    rng = np.random.default_rng(5)
    sc = np.zeros((n, 4), np.uint64)
    kind = rng.integers(0, 4, n)
    sc[kind == 1, 0] = 1
    sc[kind == 2, 0] = rng.integers(0, 1 << 16, (kind == 2).sum(), dtype=np.uint64)
    full = O.rand_scalars(92, n); sc[kind == 3] = full[kind == 3]
    assert (tab.msm_bigint(sc) == plain.msm_bigint(sc)).all()
    eq = np.tile(O.int_to_limbs(0xDEADBEEFCAFEF00D1234567, 4), (n, 1))          # one bucket per window holds everything
    assert (tab.msm_bigint(eq) == plain.msm_bigint(eq)).all()
    assert (tab.msm_bigint(np.zeros_like(sc)) == plain.msm_bigint(np.zeros_like(sc))).all()
    hi = full.copy(); hi[:, 3] |= np.uint64(1 << 63)                             # scalars >= 2^255 are refused (include/dock_gpu.h: no width-independent meaning)
    with pytest.raises(ca.DockGpuError) as e:
        tab.msm_bigint(hi)
    assert e.value.code == -3
    assert (tab.msm_bigint(full) == plain.msm_bigint(full)).all()
    # everything cancels -> identity
    b2 = np.concatenate([bases[:50], bases[:50]]); b2[50:, 6:] = np.stack([U.fp_abi((-U.fp_int(y)) % U.P) for y in bases[:50, 6:]])
    t2 = ca.DeviceBases(curve, b2).precompute(16)
    assert not t2.msm_bigint(np.concatenate([full[:50], full[:50]]))[12:].any()


def test_table_2_20_closed_form_and_split():
    """BASELINE config 2 size on the table path (automatic width: c = 20, W = 13): closed form over known dlogs, split/merge, G2 at 2^18"""
    G, curve = O.G1, ca.G1
    n = 1 << 20
    bases, k0, d = U.seq_bases(G, n, 7777, threads=64)
    sc = O.rand_scalars(7779, n)
    tab = ca.DeviceBases(curve, bases).precompute()
    r = tab.msm_bigint(sc)
    assert U.jac_to_model(G, r) == U.closed_form(G, sc, k0, d)
    ds = ca.DeviceScalars(sc)
    a = tab.msm_resident(ds, n=n // 2)
    b = tab.msm_resident(ds, n=n - n // 2, base_offset=n // 2, scalar_offset=n // 2)
    assert U.jac_to_model(G, G.add(a, b)) == U.jac_to_model(G, r)
    assert (tab.msm_resident(ds) == r).all()
    tab.free(); ds.free()
    G, curve = O.G2, ca.G2
    n = 1 << 18
    bases, k0, d = U.seq_bases(G, n, 5252, threads=64)
    sc = O.rand_scalars(5253, n)
    tab = ca.DeviceBases(curve, bases).precompute(20)
    assert U.jac_to_model(G, tab.msm_bigint(sc)) == U.closed_form(G, sc, k0, d)


def test_sharded_handle_precomputed():
    ca.init_devices([0, 0])
    G, curve = O.G1, ca.G1
    n = 40000
    bases, _, _ = U.seq_bases(G, n, 55, threads=16)
    sc = O.rand_scalars(56, n)
    ref = ca.msm_bigint(curve, bases, sc)
    sh = ca.ShardedDeviceBases(curve, bases).precompute(18)
    assert (sh.msm_bigint(sc) == ref).all()
    ds = sh.upload_scalars(sc)
    assert (sh.msm_resident(ds) == ref).all()
    assert (sh.msm_bigint(sc[:25001]) == ca.msm_bigint(curve, bases[:25001], sc[:25001])).all()
    ds.free(); sh.free()
    lib().dgpu_set_device(0)


@pytest.mark.parametrize("n,c", [(700, 16), (5003, 18), ((1 << 15) + 5, 20), ((1 << 16) + 3, 0)])
def test_one_sort_shared_by_tables_of_one_shape(n, c):
    """dgpu_scalars_sort + dgpu_msm_g1/g2_sorted: the MSMs of a G1 table, of a second G1 table with identity rows and of a G2 table of the
    same shape over ONE sorted scalar list equal dgpu_msm_*_resident limb for limb — uniform and Groth16-like scalars (zeros, ones, hot
    buckets, all equal), row / scalar offsets (`&query[1..]`), several MSMs on the list at once; tables of another shape are refused."""
    from concurrent.futures import ThreadPoolExecutor
    b1, _, _ = U.seq_bases(O.G1, n, 2100 + n, threads=16); b1b, _, _ = U.seq_bases(O.G1, n, 2200 + n, threads=16); b2, _, _ = U.seq_bases(O.G2, n, 2300 + n, threads=16)
    inf_b = np.zeros(n, np.uint8); inf_b[::7] = 1; b1b[5] = 0; inf2 = np.zeros(n, np.uint8); inf2[3::11] = 1            # B queries have identity rows
    ta = ca.DeviceBases(ca.G1, b1).precompute(c); tb = ca.DeviceBases(ca.G1, b1b, inf_b).precompute(c); t2 = ca.DeviceBases(ca.G2, b2, inf2).precompute(c)
    assert ta.same_table_shape(tb) and ta.same_table_shape(t2)
    inf_l = np.zeros(n, np.uint8); inf_l[2::13] = 1
    tl = {k: ca.DeviceBases(ca.G1, b1b[k:], inf_l[k:]).precompute(c if c else ta.table_shape()[1]) for k in (1, 3, n // 5)}
    rng = np.random.default_rng(n)
    uniform = O.rand_scalars(2400 + n, n)
    groth = uniform.copy(); kind = rng.integers(0, 4, n); groth[kind <= 1] = 0; groth[kind == 1, 0] = 1; m = kind == 2; groth[m, 1:] = 0; groth[m, 0] &= np.uint64(0xFFFF)
    equal = np.repeat(uniform[:1], n, 0)
    for sc in (uniform, groth, equal):
        ds = ca.DeviceScalars(sc)
        for boff, soff, cnt in ((0, 0, n), (1, 0, n - 1), (1, 1, n - 2), (5, 9, n // 3)):
            srt = ca.SortedScalars(ta, ds, cnt, base_offset=boff, scalar_offset=soff)
            with ThreadPoolExecutor(3) as ex:
                got = list(ex.map(lambda t: t.msm_sorted(srt), (ta, tb, t2)))
            for t, g in zip((ta, tb, t2), got):
                assert (g == t.msm_resident(ds, n=cnt, base_offset=boff, scalar_offset=soff)).all(), (boff, soff, cnt)
            # a table k rows shorter that holds the points of rows k .. n - 1 (the l_query against the a_query's list): rows < k are passed over
            for k in (1, 3, n // 5):
                i0 = max(0, k - boff)
                if cnt - i0 > 0:
                    assert (tl[k].msm_sorted(srt, row_shift=k) == tl[k].msm_resident(ds, n=cnt - i0, base_offset=boff + i0 - k, scalar_offset=soff + i0)).all(), (k, boff, soff, cnt)
            srt.free()
        ds.free()
    # the oracle itself on one case
    ds = ca.DeviceScalars(groth); srt = ca.SortedScalars(tb, ds, n)
    if n <= 5003:
        infb = inf_b.copy(); infb[5] = 1
        assert U.jac_to_model(O.G1, tb.msm_sorted(srt)) == U.jac_to_model(O.G1, O.G1.msm(b1b, groth, infb, threads=16))
    # another shape: a different width or row count
    other = ca.DeviceBases(ca.G1, b1[:n - 1]).precompute(c)
    wide = ca.DeviceBases(ca.G1, b1).precompute(17 if c != 17 else 19)
    plain = ca.DeviceBases(ca.G1, b1)
    for bad in (other, wide, plain):
        assert not ta.same_table_shape(bad)
        with pytest.raises(ca.DockGpuError):
            bad.msm_sorted(srt)
    out = np.zeros(18, np.uint64)
    import ctypes as C
    assert lib().dgpu_msm_g1_sorted(ta.handle, ds.handle, 0, out.ctypes.data_as(C.c_void_p)) == -3          # a scalars handle is not a sorted list
    h = C.c_uint64(0)
    assert lib().dgpu_scalars_sort(plain.handle, 0, ds.handle, 0, n, C.byref(h)) == -3                    # not a table
    assert lib().dgpu_scalars_sort(ta.handle, 2, ds.handle, 0, n, C.byref(h)) == -3                       # rows past the end
    srt.free(); ds.free()
    with pytest.raises(ca.DockGpuError):
        ta.msm_sorted(ca.SortedScalars(ta, ca.DeviceScalars(uniform), n), row_shift=1)          # the table is not one row shorter
    for t in (ta, tb, t2, other, wide, plain) + tuple(tl.values()):
        t.free()


@pytest.mark.parametrize("gname,n,c", [("G1", 1 << 19, 0), ("G2", 1 << 17, 20), ("G1", (1 << 15) + 77, 16)])
def test_few_distinct_bases_collide_everywhere(gname, n, c):
    """The headline shape (n >= 320 000: 20-bit windows, 13 of them, ONE bucket set, bit-marginal reduction) on inputs made of collisions: the bases are
    k G for k in {1, -1, 2, -2, 3} repeated, so a bucket's run adds a point to itself (the doubling fix-up of the mixed addition), to its negative (an
    identity accumulator in mid-run), and the reduction's running sums and wave network meet equal and opposite partial sums; scalars uniform, all
    equal, +-1, and one value per base class.  Expected point: (sum s_i k_i) G by the CPU oracle's double-and-add — no MSM code on the checking side."""
    G, curve = (O.G1, ca.G1) if gname == "G1" else (O.G2, ca.G2)
    R = U.R
    kk = [1, R - 1, 2, R - 2, 3]
    gen = G.generator()
    pts = np.stack([G.to_affine(G.mul(gen, O.int_to_limbs(k, 4)))[0] for k in kk])
    rng = np.random.default_rng(n % 1000 + c)
    idx = rng.integers(0, len(kk), n)
    idx[: n // 8] = 0                                     # a long stretch of one and the same point
    bases = pts[idx]
    kvec = np.array(kk, dtype=object)[idx]
    tab = ca.DeviceBases(curve, bases).precompute(c)
    assert tab.table_shape()[1] == (c or 20)

    def check(sc):
        ints = [int(v[0]) | int(v[1]) << 64 | int(v[2]) << 128 | int(v[3]) << 192 for v in sc]
        tot = sum(s * int(k) for s, k in zip(ints, kvec)) % R
        exp = G.to_affine(G.mul(gen, O.int_to_limbs(tot, 4)))
        got = G.to_affine(tab.msm_bigint(sc))
        assert exp[1] == got[1] and (exp[1] or (exp[0] == got[0]).all())

    full = O.rand_scalars(4242 + n % 97, n)
    check(full)
    check(np.tile(full[3], (n, 1)))                                               # all equal: thirteen hot buckets hold everything
    pm = np.tile(O.int_to_limbs(1, 4), (n, 1)); pm[rng.integers(0, 2, n) == 1] = O.int_to_limbs(R - 1, 4)
    check(pm)                                                                     # +-1 on +-k G
    per = np.stack([full[j] for j in range(len(kk))])[idx]                        # one scalar per base class: five runs per bucket, each a chain of doublings
    check(per)
    sm = np.zeros((n, 4), np.uint64); sm[:, 0] = rng.integers(0, 4, n, dtype=np.uint64)      # digits 0 .. 3 only: four buckets, the rest of the set empty
    check(sm)
    tab.free()
