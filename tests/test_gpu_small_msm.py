"""GPU (-m gpu): the small-MSM path (crypto_amd/csrc/small_kernels.hip.h: 64 signed 4-bit windows, a table of eight multiples per base, one
tree per window; taken up to dgpu_set_small_msm_max = 8192 terms on plain bases) against the bucket pipeline on the same inputs (the knob
set to 0) and against the CPU oracle — one-shot calls, plain handles with fresh and with resident scalars, the caller's Affine structs, G1 and
G2, every block-count boundary (128 terms per block, 64 blocks), and the digit / point edge cases: zero and r - 1 scalars, scalars made of the
extreme digits (0x7, 0x8 nibbles: the carries of the signed recoding), all-equal scalars and bases (P + P inside the tree and the table),
P and -P, identity bases, a scalar with bit 255 set (refused).  A plain handle that meets the small path a second time (or is handed to
dgpu_bases_precompute_*) gets a table of the eight multiples of P, 2^64 P, 2^128 P, 2^192 P per base and runs 16 trees over 4 n leaves in ONE
launch: the same cases through that form, with offsets into the handle.  Bar: bit-exact."""
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd._native import lib
from test_gpu_msm import normalised

pytestmark = pytest.mark.gpu
CUR = {"G1": (ca.G1, O.G1), "G2": (ca.G2, O.G2)}
R = U.R


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available(), "GPU tests need a device"
    ca.init(0)
    lib().dgpu_set_min_gpu_n(1)
    yield
    lib().dgpu_set_small_msm_max(8192)
    lib().dgpu_set_min_gpu_n(0)


def both_paths(fn):
    """fn() through the tree path and through the bucket pipeline"""
    try:
        assert lib().dgpu_set_small_msm_max(8192) == 0
        a = fn()
        assert lib().dgpu_set_small_msm_max(0) == 0
        b = fn()
    finally:
        lib().dgpu_set_small_msm_max(8192)
    return a, b


@pytest.mark.parametrize("gname,n", [("G1", n) for n in (1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 257, 600, 1000, 2048, 4096, 5000, 8191, 8192)] +
                         [("G2", n) for n in (1, 2, 65, 128, 129, 600, 1500, 4096, 8192)])
def test_tree_path_equals_bucket_pipeline_and_oracle(gname, n):
    curve, G = CUR[gname]
    bases, _, _ = U.seq_bases(G, n, 3000 + n, threads=32)
    sc = O.rand_scalars(4000 + n, n)
    small, bucket = both_paths(lambda: ca.msm_bigint(curve, bases, sc))
    assert (small == bucket).all()
    if n <= 4096 or gname == "G1":
        assert (small == normalised(G, G.msm(bases, sc, threads=32))).all()
    # &[Fr] scalars (msm_unchecked), a plain handle with fresh scalars, the same with resident scalars and offsets
    assert (ca.msm_unchecked(curve, bases, O.fr_to_mont(sc)) == small).all()
    db = ca.DeviceBases(curve, bases)
    assert (db.msm_bigint(sc) == small).all()
    ds = ca.DeviceScalars(sc)
    assert (db.msm_resident(ds) == small).all()
    if n >= 3:
        part = db.msm_resident(ds, n=n - 2, base_offset=1, scalar_offset=1)
        assert (part == ca.msm_bigint(curve, bases[1:n - 1], sc[1:n - 1])).all()
    ds.free(); db.free()


@pytest.mark.parametrize("gname", ["G1", "G2"])
def test_digit_and_point_edge_cases(gname):
    curve, G = CUR[gname]
    n = 300
    bases, _, _ = U.seq_bases(G, n, 77, threads=16)
    lim = lambda v: O.int_to_limbs(v, 4)
    rnd = O.rand_scalars(78, n)
    cases = {}
    sc = rnd.copy(); sc[::3] = 0; sc[1::7] = lim(1); sc[2::11] = lim(R - 1); cases["zeros, ones, r - 1"] = (bases, sc, None)
    nib = lambda d: int(("%x" % d) * 63, 16)                                  # 252 bits of one nibble value
    sc = rnd.copy()
    for k, d in enumerate((7, 8, 0xF, 9, 1)):
        sc[k::5] = lim(nib(d) % (1 << 255))
    cases["extreme digits: carries through every window"] = (bases, sc, None)
    cases["all scalars equal"] = (bases, np.tile(rnd[5], (n, 1)), None)
    same = np.tile(bases[3], (n, 1)); cases["all bases equal (P + P at every level)"] = (same, rnd, None)
    neg = bases.copy(); h = G.AW // 2
    for i in range(1, n, 2):                                                 # P_{i} = -P_{i-1}: pairs cancel inside the tree when their scalars are equal
        neg[i] = bases[i - 1]
        for k in range(h // 6):
            y = U.fp_int(neg[i][h + 6 * k:h + 6 * k + 6]); neg[i][h + 6 * k:h + 6 * k + 6] = U.fp_abi((U.P - y) % U.P)
    sc = rnd.copy(); sc[1::2] = sc[0::2]
    cases["P and -P with equal scalars: the sum is the identity"] = (neg, sc, None)
    inf = np.zeros(n, np.uint8); inf[::4] = 1; zb = bases.copy(); zb[2::9] = 0
    cases["identity bases (flag and all-zero words)"] = (zb, rnd, inf)
    for what, (b, s, fl) in cases.items():
        small, bucket = both_paths(lambda: ca.msm_bigint(curve, b, s, fl))
        assert (small == bucket).all(), what
        fl2 = np.zeros(n, np.uint8) if fl is None else fl.copy()
        fl2 |= (~b.any(axis=1)).astype(np.uint8)
        assert (small == normalised(G, G.msm(b, s, fl2, threads=16))).all(), what
    assert not normalised(G, G.msm(neg, sc, threads=16))[G.AW:].any()         # (that case really is the identity)
    bad = rnd.copy(); bad[17, 3] |= np.uint64(1 << 63)
    with pytest.raises(ca.DockGpuError):
        ca.msm_bigint(curve, bases, bad)


@pytest.mark.parametrize("gname", ["G1", "G2"])
def test_resident_table_form(gname):
    """the table kept with a handle: built at the upload (or, for a handle made while the path was off, at its second small call), used with offsets and prefixes, the digit edge cases through
    its four sub-tables (window 16 s + v is leaf (i, s) of super-window v), identity bases inside it, a scalar >= 2^255 refused"""
    curve, G = CUR[gname]
    n = 700
    bases, _, _ = U.seq_bases(G, n, 91, threads=16)
    inf = np.zeros(n, np.uint8); inf[5::13] = 1
    lim = lambda v: O.int_to_limbs(v, 4)
    rnd = O.rand_scalars(92, n)
    nib = lambda d: int(("%x" % d) * 63, 16) % (1 << 255)
    edge = rnd.copy()
    pats = [0, 1, R - 1, nib(8), nib(8) + 1, nib(8) - 1, nib(7), nib(9), nib(0xF), (1 << 255) - 1, (1 << 64) - 1, 1 << 64, (1 << 128) - 1, 1 << 128, (1 << 192) - 1, 1 << 192,
            0x8 << 60, 0x9 << 60, (0x88888888 << 32) | 0x88888889, 0x88888888_88888888_88888888_88888888, 0x88888888_88888888_88888888_88888889 << 64]
    for k, v in enumerate(pats):
        edge[k::len(pats) + 3] = lim(v)
    want = {"rnd": normalised(G, G.msm(bases, rnd, inf, threads=16)), "edge": normalised(G, G.msm(bases, edge, inf, threads=16))}
    for how in ("upload", "precompute", "second call"):
        # the table comes with the upload (<= 8192 bases) ...; "precompute": asking again is a no-op; "second call": a handle uploaded while the small path
        # was switched off gets it lazily, at its second small MSM
        if how == "second call":
            assert lib().dgpu_set_small_msm_max(0) == 0
        (lib().dgpu_reserve_g1 if gname == "G1" else lib().dgpu_reserve_g2)(n)      # every slot's workspaces exist: what is allocated from here on belongs to the handle
        a0 = ca.device_alloc_count()
        db = ca.DeviceBases(curve, bases, inf)
        lib().dgpu_set_small_msm_max(8192)
        per_handle = ca.device_alloc_count() - a0               # the records, and the table unless it comes later
        if how == "precompute":
            db.precompute()                                   # 700 bases: no bucket table; the small path's table exists already
        a1 = ca.device_alloc_count()
        assert a1 == a0 + per_handle
        for rep in range(3):
            assert (db.msm_bigint(rnd) == want["rnd"]).all(), (how, rep)
            assert (db.msm_bigint(edge) == want["edge"]).all(), (how, rep)
        assert ca.device_alloc_count() == a1 + (1 if how == "second call" else 0), how      # the calls allocate nothing, except the one late table
        ds = ca.DeviceScalars(edge)
        assert (db.msm_resident(ds) == want["edge"]).all()
        for off, cnt in ((1, n - 1), (0, 64), (63, 130), (n - 1, 1), (300, 257)):
            got = db.msm_resident(ds, n=cnt, base_offset=off, scalar_offset=off)
            assert (got == normalised(G, G.msm(bases[off:off + cnt], edge[off:off + cnt], inf[off:off + cnt], threads=16))).all(), (off, cnt)
        small, bucket = both_paths(lambda: db.msm_resident(ds))     # the knob still switches the handle's calls to the bucket pipeline
        assert (small == bucket).all()
        bad = rnd.copy(); bad[n - 3, 3] |= np.uint64(1 << 63)
        with pytest.raises(ca.DockGpuError):
            db.msm_bigint(bad)
        assert (db.msm_bigint(rnd) == want["rnd"]).all()       # the refused call left the window counters clean
        ds.free(); db.free()


def test_concurrent_small_calls_do_not_share_state():
    """six host threads, different sizes: every call owns its slot's table, partials and window counters"""
    from concurrent.futures import ThreadPoolExecutor
    G, curve = O.G1, ca.G1
    jobs = []
    for k, n in enumerate((130, 600, 1000, 257, 4096, 64, 900, 2500)):
        b, _, _ = U.seq_bases(G, n, 500 + k, threads=16); s = O.rand_scalars(600 + k, n)
        jobs.append((b, s, normalised(G, G.msm(b, s, threads=16))))
    with ThreadPoolExecutor(6) as ex:
        for rep in range(3):
            got = list(ex.map(lambda j: ca.msm_bigint(curve, j[0], j[1]), jobs))
            assert all((g == j[2]).all() for g, j in zip(got, jobs))


def test_concurrent_calls_on_one_handle_build_its_table_once():
    """eight host threads meet a handle that has no table yet (uploaded while the path was off): one of them builds it, all of them return the right point,
    every later call runs on the table"""
    from concurrent.futures import ThreadPoolExecutor
    G, curve = O.G1, ca.G1
    n = 1500
    b, _, _ = U.seq_bases(G, n, 71, threads=16)
    scs = [O.rand_scalars(720 + k, n) for k in range(8)]
    refs = [normalised(G, G.msm(b, s, threads=16)) for s in scs]
    lib().dgpu_reserve_g1(n)
    assert lib().dgpu_set_small_msm_max(0) == 0
    db = ca.DeviceBases(curve, b)
    assert lib().dgpu_set_small_msm_max(8192) == 0
    a0 = ca.device_alloc_count()
    with ThreadPoolExecutor(8) as ex:
        for rep in range(3):
            got = list(ex.map(lambda k: db.msm_bigint(scs[k]), range(8)))
            assert all((g == r).all() for g, r in zip(got, refs)), rep
    assert ca.device_alloc_count() == a0 + 1          # the table, once
    db.free()


def test_no_device_allocation_in_steady_state():
    G, curve = O.G1, ca.G1
    n = 3000
    b, _, _ = U.seq_bases(G, n, 9, threads=16); s = O.rand_scalars(10, n)
    lib().dgpu_reserve_g1(n)
    ref = ca.msm_bigint(curve, b, s)
    a0 = ca.device_alloc_count()
    for _ in range(8):
        assert (ca.msm_bigint(curve, b, s) == ref).all()
    assert ca.device_alloc_count() == a0
