"""GPU (-m gpu): the RandomizedPairingChecker mirror behaves like the reference's
(/root/reference/utils/src/randomized_pairing_check.rs tests :234-420): it accepts batches of true pairing equations,
rejects a batch with one wrong target / swapped source, in lazy and eager mode, for the three ways of adding
equations; and the batched G1 scaling it relies on matches the oracle's mul_bigint."""
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd import pairing_check as pc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def g1(k):
    return O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(k % U.R, 4)))[0]


def g2(k):
    return O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(k % U.R, 4)))[0]


def gt(ps, qs):
    return O.final_exponentiation(O.multi_miller_loop(np.asarray(ps).reshape(-1, 12), np.asarray(qs).reshape(-1, 24)))


def test_g1_scale_batch_matches_mul_bigint():
    rng = np.random.default_rng(1)
    ks = [int(x) for x in rng.integers(1, 1 << 62, 40)]
    pts = np.stack([g1(k) for k in ks])
    m = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF12345678 % U.R
    out, inf = pc.g1_scale(pts, m)
    assert not inf.any()
    for k, o in zip(ks, out):
        assert (o == g1(k * m)).all()
    outn, _ = pc.g1_scale(pts[:5], m, negate=True)
    for k, o in zip(ks, outn):
        assert (o == g1(-(k * m))).all()
    # scalar 0 and r-1; identity input
    out0, inf0 = pc.g1_scale(pts[:3], 0)
    assert inf0.all() and not out0.any()
    outm, _ = pc.g1_scale(pts[:3], U.R - 1)
    for k, o in zip(ks, outm):
        assert (o == g1(-k)).all()
    z = np.zeros((2, 12), np.uint64)
    _, infz = pc.g1_scale(z, 5)
    assert infz.all()


@pytest.mark.parametrize("lazy", [False, True])
def test_checker_accepts_true_equations_and_rejects_false(lazy):
    rng = np.random.default_rng(3)
    r = int(rng.integers(1, 1 << 62)) * 0x9E3779B97F4A7C15 % U.R

    def fill(chk, corrupt=None):
        # (1) e(a1 G1, b1 G2) == out            add_sources_and_target
        a1, b1 = 11, 13
        out1 = gt(g1(a1), g2(b1))
        if corrupt == 1:
            out1 = gt(g1(a1 + 1), g2(b1))
        chk.add_sources_and_target(g1(a1), g2(b1), out1)
        # (2) e(a2 G1, b2 G2) == e(c2 G1, d2 G2) with a2 b2 == c2 d2    add_sources
        a2, b2, c2, d2 = 6, 35, 21, 10
        if corrupt == 2:
            d2 = 11
        chk.add_sources(g1(a2), g2(b2), g1(c2), g2(d2))
        # (3) prod e(a_i, b_i) == out            add_multiple_sources_and_target
        a3, b3 = [3, 5, 7], [2, 4, 6]
        out3 = gt(np.stack([g1(x) for x in a3]), np.stack([g2(x) for x in b3]))
        chk.add_multiple_sources_and_target(np.stack([g1(x) for x in a3]), np.stack([g2(x) for x in b3]), out3)
        # (4) prod e(a_i, b_i) == prod e(c_i, d_i)   add_multiple_sources   (2*9 + 4*3 == 5*6)
        a4, b4, c4, d4 = [2, 4], [9, 3], [5], [6]
        if corrupt == 4:
            a4 = [2, 5]
        chk.add_multiple_sources(np.stack([g1(x) for x in a4]), np.stack([g2(x) for x in b4]),
                                 np.stack([g1(x) for x in c4]), np.stack([g2(x) for x in d4]))

    chk = ca.RandomizedPairingChecker(r, lazy)
    fill(chk)
    assert chk.verify()
    for bad in (1, 2, 4):
        chk = ca.RandomizedPairingChecker(r, lazy)
        fill(chk, corrupt=bad)
        assert not chk.verify(), bad


def test_lazy_and_eager_agree_and_laziness_override():
    r = 0xDEADBEEF12345
    a, b = np.stack([g1(3), g1(5)]), np.stack([g2(7), g2(9)])
    out = gt(a, b)
    e = ca.RandomizedPairingChecker(r, False)
    l = ca.RandomizedPairingChecker(r, True)
    for c in (e, l):
        c.add_multiple_sources_and_target(a, b, out)
        c.add_multiple_sources_and_target(a, b, out, lazy=not c.lazy)      # *_with_laziness_choice
    assert e.verify() and l.verify()
    assert len(l.pending[0]) == 1 and len(e.pending[0]) == 1
    assert (e.right == l.right).all()


def test_g1_scale_glv_edge_scalars():
    """the scaling kernel splits k = k1 + k2 lambda (GLV): scalars around the split points, 0, 1, r - 1, values >= r, an identity point,
    per-point and shared scalars, with and without negation — all against the oracle's plain double-and-add"""
    lam = 0xac45a4010001a40200000000ffffffff
    R = U.R
    ks = [0, 1, 2, lam - 1, lam, lam + 1, 2 * lam, lam * lam % R, R - 1, R, R + 5, (1 << 128) - 1, 1 << 128, (1 << 255) - 19, (1 << 256) - 1, 0x1234567]
    pts = np.stack([g1(3 + 7 * i) for i in range(len(ks))])
    pts[7] = 0                                                  # identity point
    sc = np.stack([O.int_to_limbs(k, 4) for k in ks])
    out, inf = pc.g1_scale_each(pts, sc)
    outn, infn = pc.g1_scale_each(pts, sc, np.ones(len(ks), np.uint8))
    for i, k in enumerate(ks):
        e, einf = O.G1.to_affine(O.G1.mul(pts[i], O.int_to_limbs(k % R, 4), inf=not pts[i].any()))
        assert bool(inf[i]) == einf and bool(infn[i]) == einf, i
        if not einf:
            assert (out[i] == e).all(), i
            assert (outn[i][:6] == e[:6]).all() and not (outn[i][6:] == e[6:]).all(), i      # same x, other y
    same, sinf = pc.g1_scale(pts, lam + 12345)
    for i in (0, 3, 7, 15):
        e, einf = O.G1.to_affine(O.G1.mul(pts[i], O.int_to_limbs(lam + 12345, 4), inf=not pts[i].any()))
        assert bool(sinf[i]) == einf and (einf or (same[i] == e).all())


@pytest.mark.parametrize("lazy", [False, True])
def test_checker_with_prepared_g2_operands(lazy):
    """the reference queues `impl Into<E::G2Prepared>` (randomized_pairing_check.rs:61-77,119-138; its tests pass prepared values,
    :294-296): prepared, unprepared and mixed operands give the same verdict and the same `left`"""
    from crypto_amd import pairing
    r = 0xC0FFEE1234567
    a, b = np.stack([g1(3), g1(5), g1(8)]), np.stack([g2(7), g2(9), g2(2)])
    out = gt(a, b)
    bp = pairing.G2Prepared.from_affine(b)
    verdicts, lefts = [], []
    for bb in (b, bp):
        chk = ca.RandomizedPairingChecker(r, lazy)
        chk.add_multiple_sources_and_target(a, bb, out)
        chk.add_sources(g1(6), bb[:1] if bb is b else bb[0], g1(42), g2(1))        # e(6 G1, 7 G2) == e(42 G1, G2)
        chk.add_sources_and_target(g1(11), pairing.G2Prepared.from_affine(g2(13)) if bb is bp else g2(13), gt(g1(11), g2(13)))
        verdicts.append(chk.verify()); lefts.append(chk.left)
    assert verdicts == [True, True]
    if not lazy:
        assert (lefts[0] == lefts[1]).all()
    bad = ca.RandomizedPairingChecker(r, lazy)
    bad.add_multiple_sources_and_target(a, pairing.G2Prepared.from_affine(np.stack([g2(7), g2(9), g2(3)])), out)
    assert not bad.verify()


@pytest.mark.parametrize("n,n_prep", [(1, 0), (3, 2), (70, 0), (1024, 2), (5000, 0), (9000, 1)])
def test_scaled_miller_loop_is_scale_then_miller(n, n_prep):
    """dgpu_multi_miller_loop_scaled: prod e([m_i] P_i, Q_i) x prod e(P'_j, prepared_j), limb for limb what dgpu_g1_scale_batch followed by the Miller loop
    returns (the scalings run beside the chain of the Q_i: utils/src/randomized_pairing_check.rs:125-134 as one call) and what the CPU oracle computes from
    scaled points; zero scalars, identity points and skip flags drop their pair; one scalar for all pairs; n = 5000 scales with one quad per point inside the one-call form, n = 9000 takes the two-call form"""
    from crypto_amd import pairing
    rng = np.random.default_rng(900 + n)
    P = O.G1.gen_seq(O.rand_scalars(61, 1)[0], O.rand_scalars(62, 1)[0], n + n_prep, threads=16)
    Q = O.G2.gen_seq(O.rand_scalars(63, 1)[0], O.rand_scalars(64, 1)[0], n + n_prep, threads=16)
    sc = O.rand_scalars(65 + n, n)
    sc[rng.integers(0, 5, n) == 0, 2:] = 0                      # some short scalars
    if n > 2:
        sc[1] = 0                                               # a pair that drops out
        P[2] = 0                                                # an identity point
    skip = (rng.integers(0, 7, n) == 0).astype(np.uint8)
    prep = pairing.G2Prepared.from_affine(Q[n:]) if n_prep else None
    got = pairing.multi_miller_loop_scaled(P[:n], sc, Q[:n], skip, P[n:] if n_prep else None, prep)
    scaled, sinf = pc.g1_scale_each(P[:n], sc)
    sk = skip | sinf.astype(np.uint8)
    if n_prep:
        want = pairing.multi_miller_loop(np.concatenate([scaled, P[n:]]), [Q[:n], prep], np.concatenate([sk, np.zeros(n_prep, np.uint8)]))
    else:
        want = pairing.multi_miller_loop(scaled, Q[:n], sk)
    assert (got == want).all()
    if n <= 70:                                                 # the oracle on the same statement: scaled points by double-and-add, then its Miller loop
        ps = np.stack([O.G1.to_affine(O.G1.mul(P[i], sc[i]))[0] for i in range(n)] + [P[n + j] for j in range(n_prep)])
        osk = np.array([1 if (sk[i] or O.G1.to_affine(O.G1.mul(P[i], sc[i]))[1]) else 0 for i in range(n)] + [0] * n_prep, dtype=np.uint8)
        assert (got == O.multi_miller_loop(ps, Q, osk, threads=8)).all()
    # one scalar for every pair
    one = pairing.multi_miller_loop_scaled(P[:n], sc[0], Q[:n], skip)
    s1, i1 = pc.g1_scale(P[:n], int(sc[0][0]) | int(sc[0][1]) << 64 | int(sc[0][2]) << 128 | int(sc[0][3]) << 192)
    assert (one == pairing.multi_miller_loop(s1, Q[:n], skip | i1.astype(np.uint8))).all()


def test_scaled_miller_loop_from_many_threads():
    """six host threads call dgpu_multi_miller_loop_scaled at once (slots, side streams and the pipelined / two-call forms mix): every call returns what a lone call returns"""
    from concurrent.futures import ThreadPoolExecutor
    from crypto_amd import pairing
    n = 300
    P = O.G1.gen_seq(O.rand_scalars(71, 1)[0], O.rand_scalars(72, 1)[0], n, threads=16)
    Q = O.G2.gen_seq(O.rand_scalars(73, 1)[0], O.rand_scalars(74, 1)[0], n, threads=16)
    scs = [O.rand_scalars(80 + k, n) for k in range(4)]
    want = [pairing.multi_miller_loop_scaled(P, sc, Q) for sc in scs]
    with ThreadPoolExecutor(6) as ex:
        got = list(ex.map(lambda k: pairing.multi_miller_loop_scaled(P, scs[k % 4], Q), range(48)))
    assert all((got[k] == want[k % 4]).all() for k in range(48))
