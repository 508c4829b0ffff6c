"""GPU (-m gpu): the batched Miller loop through the C ABI vs the CPU oracle (raw Fp12 equality — the line
coefficients are arkworks' own, so the un-exponentiated Miller output must match limb for limb), the golden
fixtures, and bilinearity at BASELINE config 3's size (1024 pairs)."""
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def pts(a, b):
    g1, g2 = O.G1.generator(), O.G2.generator()
    ps = np.stack([O.G1.to_affine(O.G1.mul(g1, O.int_to_limbs(x, 4)))[0] for x in a])
    qs = np.stack([O.G2.to_affine(O.G2.mul(g2, O.int_to_limbs(x, 4)))[0] for x in b])
    return ps, qs


def test_golden_pairings():
    pr = U.load("pairing")
    g1, g2 = O.G1.generator(), O.G2.generator()
    f = ca.multi_miller_loop(g1.reshape(1, 12), g2.reshape(1, 24))
    assert U.f12_ints(f) == [int(v, 16) for v in pr["miller_g1_g2"]]
    assert U.f12_ints(ca.final_exponentiation(f)) == [int(v, 16) for v in pr["e_g1_g2"]]
    for case in pr["cases"]:
        ps = np.stack([U.g1_abi(U.dec_g1(p))[0] for p in case["p"]])
        qs = np.stack([U.g2_abi(U.dec_g2(q))[0] for q in case["q"]])
        f = ca.multi_miller_loop(ps, qs)
        assert U.f12_ints(f) == [int(v, 16) for v in case["miller"]]
        assert U.f12_ints(ca.multi_pairing(ps, qs)) == [int(v, 16) for v in case["gt"]]


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 63, 64, 65, 200])
def test_vs_oracle_raw_miller_output(n):
    rng = np.random.default_rng(n)
    a = [int(x) for x in rng.integers(1, 1 << 62, n)]
    b = [int(x) for x in rng.integers(1, 1 << 62, n)]
    ps, qs = pts(a, b) if n else (np.zeros((0, 12), np.uint64), np.zeros((0, 24), np.uint64))
    skip = (rng.integers(0, 5, n) == 0).astype(np.uint8) if n else None
    got = ca.multi_miller_loop(ps, qs, skip)
    ref = O.multi_miller_loop(ps, qs, skip, threads=16) if n else O.fp12_one()
    assert (got == ref).all()


@pytest.mark.parametrize("n", [255, 257, 1025, 2049, 4100, 8192, 8193])
def test_vs_oracle_across_kernel_boundaries(n):
    """batch sizes around the boundaries of the product kernels (groups of 64 slices, two tree levels, slice length 4 -> 8 at 2048 pairs)
    and of the line kernel (four lanes per pair up to 8192 pairs, two above): raw Miller-loop output limb for limb"""
    k0 = O.rand_scalars(31, 1)[0]; d = O.rand_scalars(32, 1)[0]
    ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
    ps[n // 3] = 0; qs[n // 2] = 0                      # identity members (all-zero words) are skipped
    skip = np.zeros(n, np.uint8); skip[n // 3] = 1; skip[n // 2] = 1
    assert (ca.multi_miller_loop(ps, qs) == O.multi_miller_loop(ps, qs, skip, threads=32)).all()


@pytest.mark.parametrize("n", [1, 3, 64, 130, 1024, 1100, 5000, 8192, 9000])
def test_every_form_of_the_miller_kernels_gives_the_same_value(n, twin):
    """dgpu_set_miller_pipeline: bit 0 cuts the 68-step chain of a call of up to 8192 pairs at bit 17 of |x| (two launches of the line
    kernel, evaluation at P moved into the product kernel, products and host share of the first 50 steps overlapped with the second
    launch), bit 1 runs the product tree with 18 lane pairs per node.  All four combinations: the same Fp12 value limb for limb, equal
    to the oracle's, identity members and skipped pairs included"""
    from crypto_amd._native import lib
    k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
    ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
    skip = np.zeros(n, np.uint8)
    if n >= 3:
        ps[n // 3] = 0; skip[n // 3] = 1; skip[n - 1] = 1
    got = {}
    try:
        for mode in range(32):
            assert lib().dgpu_set_miller_pipeline(mode) == 0
            got[mode] = ca.multi_miller_loop(ps, qs, skip)
        assert lib().dgpu_set_miller_pipeline(32) != 0
    finally:
        lib().dgpu_set_miller_pipeline(31)
    for mode in range(1, 32):
        assert (got[mode] == got[0]).all(), mode
    if n <= 1100:
        assert (got[31] == O.multi_miller_loop(ps, qs, skip, threads=32)).all()
    if n <= 1100:
        from crypto_amd import pairing
        inf = skip.copy()
        try:
            lib().dgpu_set_miller_pipeline(3); pc4 = pairing.G2Prepared.from_affine(qs, inf)
            lib().dgpu_set_miller_pipeline(7); pc16 = pairing.G2Prepared.from_affine(qs, inf)
            lib().dgpu_set_miller_pipeline(15); pcw = pairing.G2Prepared.from_affine(qs, inf)
        finally:
            lib().dgpu_set_miller_pipeline(31)
        assert (pc4.coeffs == pc16.coeffs).all() and (pc4.infinity == pc16.infinity).all()
        assert (pcw.coeffs == pc16.coeffs).all() and (pcw.infinity == pc16.infinity).all()
    # the verifier's call (dgpu_multi_miller_loop_mixed: some pairs affine, the others prepared) cuts the affine pairs' chain the same way
    if 3 <= n <= 1100:
        from crypto_amd import pairing
        pc = pairing.G2Prepared.from_affine(qs)
        cut = max(1, n // 3)
        try:
            for mode in range(32):
                assert lib().dgpu_set_miller_pipeline(mode) == 0
                assert (pairing.multi_miller_loop(ps, [qs[:cut], pc[cut:]], skip) == got[0]).all(), mode
        finally:
            lib().dgpu_set_miller_pipeline(31)


def test_identity_members_are_skipped_and_lengths_checked():
    ps, qs = pts([3, 5, 7], [2, 4, 6])
    ps2 = ps.copy(); ps2[1] = 0                       # all-zero words == identity
    skip = np.array([0, 1, 0], np.uint8)
    assert (ca.multi_miller_loop(ps2, qs) == O.multi_miller_loop(ps, qs, skip)).all()
    with pytest.raises(ca.DockGpuError):
        ca.multi_miller_loop(ps[:2], qs)
    assert ca.final_exponentiation(np.zeros(72, np.uint64)) is None      # arkworks: None


def test_batched_check_1024_pairs():
    """config 3: prod e(a_i G1, b_i G2) == e(G1, G2)^(sum a_i b_i); and the RandomizedPairingChecker shape:
    prod e(P_i, Q_i) * e(-sum..., ) == 1 (utils/src/randomized_pairing_check.rs:204-214)."""
    n = 1024
    rng = np.random.default_rng(7)
    a = [int(x) for x in rng.integers(1, 1 << 40, n)]
    b = [int(x) for x in rng.integers(1, 1 << 40, n)]
    ps, qs = pts(a, b)
    f = ca.multi_miller_loop(ps, qs)
    assert (f == O.multi_miller_loop(ps, qs, threads=64)).all()
    gt = ca.final_exponentiation(f)
    g1, g2 = O.G1.generator(), O.G2.generator()
    e = O.final_exponentiation(O.multi_miller_loop(g1.reshape(1, 12), g2.reshape(1, 24)))
    assert (gt == O.fp12_pow(e, sum(x * y for x, y in zip(a, b)) % U.R)).all()
    # append the pair (-(sum a_i b_i) G1, G2): the product must be one
    tot = sum(x * y for x, y in zip(a, b)) % U.R
    pn, qn = pts([U.R - tot], [1])
    f2 = ca.multi_miller_loop(np.concatenate([ps, pn]), np.concatenate([qs, qn]))
    assert (ca.final_exponentiation(f2) == O.fp12_one()).all()


# ---- G2Prepared: the form the reference's verifier / pairing checker hold (verifier.rs:69-76, randomized_pairing_check.rs:35) ----
@pytest.mark.parametrize("n", [1, 2, 7, 64, 300])
def test_g2_prepare_coefficients_equal_the_oracle(n):
    """dgpu_g2_prepare == the oracle's restatement of ark-ec G2Prepared::from, all 68 x 3 Fp2 coefficients, limb for limb"""
    from crypto_amd import pairing
    k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
    qs = O.G2.gen_seq(k0, d, n, threads=8)
    if n >= 7:
        qs[3] = 0                                         # identity: infinity flag, all-zero block (arkworks: empty ell_coeffs)
    pc = pairing.G2Prepared.from_affine(qs)
    assert pc.coeffs.shape == (n, 68 * 36)
    for i in range(n):
        if n >= 7 and i == 3:
            assert pc.infinity[i] == 1 and not pc.coeffs[i].any()
        else:
            assert pc.infinity[i] == 0 and (pc.coeffs[i] == O.g2_prepare(qs[i]).reshape(-1)).all(), i


@pytest.mark.parametrize("n", [1, 7, 300])
def test_g2_prepare_lane_pair_chain_writes_the_same_bytes(n, twin):
    """the lane-pair chain that converts its coefficients on the way (dgpu_set_miller_pipeline without bit 0) writes the same bytes as the default
    four-lane chain + conversion pass, and both equal the oracle's"""
    from crypto_amd import pairing
    from crypto_amd._native import lib
    k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
    qs = O.G2.gen_seq(k0, d, n, threads=8)
    if n >= 7:
        qs[3] = 0
    pc = pairing.G2Prepared.from_affine(qs)
    try:
        assert lib().dgpu_set_miller_pipeline(2) == 0
        old = pairing.G2Prepared.from_affine(qs)
    finally:
        lib().dgpu_set_miller_pipeline(31)
    assert (old.coeffs == pc.coeffs).all() and (old.infinity == pc.infinity).all()
    i = n - 1
    assert (pc.coeffs[i] == O.g2_prepare(qs[i]).reshape(-1)).all()


@pytest.mark.parametrize("n", [1, 3, 5, 200, 1024, 2500])
def test_prepared_equals_unprepared_raw_output(n):
    """multi_miller_loop on prepared operands == on the points themselves, raw Fp12 limbs (the reference asserts prepared == unprepared
    pairings, utils/src/msm.rs:261-276), with coefficients from the device AND from the oracle (= what a Rust host would hold)"""
    from crypto_amd import pairing
    k0 = O.rand_scalars(51, 1)[0]; d = O.rand_scalars(52, 1)[0]
    ps = O.G1.gen_seq(k0, d, n, threads=16); qs = O.G2.gen_seq(d, k0, n, threads=16)
    skip = None
    if n >= 5:
        ps[1] = 0; qs[4] = 0
        skip = np.zeros(n, np.uint8); skip[1] = 1; skip[4] = 1
    ref = O.multi_miller_loop(ps, qs, skip, threads=32)
    assert (ca.multi_miller_loop(ps, qs) == ref).all()
    pc = pairing.G2Prepared.from_affine(qs)
    assert (pairing.multi_miller_loop(ps, pc) == ref).all()
    if n <= 200:
        inf = np.array([0 if q.any() else 1 for q in qs], np.uint8)
        oc = np.stack([O.g2_prepare(q).reshape(-1) if q.any() else np.zeros(68 * 36, np.uint64) for q in qs])
        assert (pairing.multi_miller_loop(ps, pairing.G2Prepared(oc, inf)) == ref).all()
    # mixed operands, the verifier's shape: [b.into(), prepared, prepared]
    if n == 3:
        assert (pairing.multi_miller_loop(ps, [qs[:1], pc[1], pc[2]]) == ref).all()
    with pytest.raises(ca.DockGpuError):
        pairing.multi_miller_loop(ps, pc[:n - 1] if n > 1 else pairing.G2Prepared(np.zeros((0, 68 * 36), np.uint64), np.zeros(0, np.uint8)))


@pytest.mark.parametrize("n", [2, 7, 300, 9000])
def test_mixed_affine_and_prepared_operands_in_one_call(n):
    """dgpu_multi_miller_loop_mixed: any split of the pairs into affine and prepared operands gives the raw Fp12 output of the unprepared loop
    limb for limb (the product does not depend on the order of the pairs), identity members and skip flags on both sides, empty sides, and
    argument checks."""
    from crypto_amd import pairing
    k0 = O.rand_scalars(61, 1)[0]; d = O.rand_scalars(62, 1)[0]
    ps = O.G1.gen_seq(k0, d, n, threads=16); qs = O.G2.gen_seq(d, k0, n, threads=16)
    if n >= 7:
        ps[1] = 0; qs[5] = 0                         # identity members: one in each half of the later splits
    ref = ca.multi_miller_loop(ps, qs)
    if n <= 300:
        assert (ref == O.multi_miller_loop(ps, qs, np.array([0 if (p.any() and q.any()) else 1 for p, q in zip(ps, qs)], np.uint8), threads=16)).all()
    pc = pairing.G2Prepared.from_affine(qs)
    for cut in sorted({0, 1, n // 3, n - 1, n}):
        # first `cut` pairs affine, the rest prepared; and the other way round
        assert (pairing.multi_miller_loop(ps, [qs[:cut], pc[cut:]] if 0 < cut < n else ([pc] if cut == 0 else [qs])) == ref).all(), cut
        if 0 < cut < n:
            assert (pairing.multi_miller_loop(ps, [pc[:cut], qs[cut:]]) == ref).all(), cut
    # interleaved, the batch verifier's shape: affine, prepared, prepared, affine, ...
    if n >= 7:
        items = [qs[0:1], pc[1:3], qs[3:4], pc[4:6], qs[6:]]
        assert (pairing.multi_miller_loop(ps, items) == ref).all()
        skip = np.zeros(n, np.uint8); skip[0] = 1; skip[2] = 1
        sk_ref = ca.multi_miller_loop(ps, qs, skip)
        assert (pairing.multi_miller_loop(ps, items, skip) == sk_ref).all() and not (sk_ref == ref).all()
    with pytest.raises(ca.DockGpuError):
        pairing.multi_miller_loop(ps, [qs[:1], pc[2:]])                     # one operand short


@pytest.mark.parametrize("sizes", [[1, 1], [0, 3, 0, 1, 7], [5, 64, 65, 2, 256, 257, 1], [1, 2, 4, 8, 16, 32, 64, 128, 256, 512], [3000, 1, 0, 900], [9000, 4, 17],
                                   [2048, 2048, 1024, 1024, 1024, 1024], [8192, 513, 0, 700, 4097]])
def test_segmented_miller_loops_equal_the_single_calls(sizes):
    """dgpu_multi_miller_loop_segments: every segment's raw Fp12 output is limb for limb what dgpu_multi_miller_loop returns for that segment
    alone — empty segments (one), single pairs, sizes around the slice and group borders, identity members, skip flags, a segment long enough
    for the per-segment path, segments long enough for the second tree level (the commitments of a 1024-proof aggregation; 8192 + 4097: with and without the
    pieces), and the argument checks."""
    import ctypes as C
    from crypto_amd import pairing
    from crypto_amd._native import lib
    n = sum(sizes)
    k0 = O.rand_scalars(71, 1)[0]; d = O.rand_scalars(72, 1)[0]
    ps = O.G1.gen_seq(k0, d, n, threads=16); qs = O.G2.gen_seq(d, k0, n, threads=16)
    if n > 12:
        ps[3] = 0; qs[n - 2] = 0
    ends = np.cumsum(sizes); starts = ends - np.array(sizes)
    jobs = [(ps[a:b], qs[a:b]) for a, b in zip(starts, ends)]
    got = pairing.multi_miller_loops(jobs)
    assert len(got) == len(sizes)
    for (a, b), f in zip(zip(starts, ends), got):
        assert (f == ca.multi_miller_loop(ps[a:b], qs[a:b])).all(), (a, b)
    gts = pairing.multi_pairings(jobs)
    for f, g in zip(got, gts):
        assert (g == ca.final_exponentiation(f)).all()
    if n <= 1100:       # and the oracle itself on the largest segment
        a, b = max(zip(starts, ends), key=lambda ab: ab[1] - ab[0])
        sk = np.array([0 if (p.any() and q.any()) else 1 for p, q in zip(ps[a:b], qs[a:b])], np.uint8)
        assert (got[list(ends).index(b)] == O.multi_miller_loop(ps[a:b], qs[a:b], sk, threads=16)).all()
    p_ = lambda x: x.ctypes.data_as(C.c_void_p)
    skip = (np.arange(n) % 5 == 1).astype(np.uint8)
    e64 = ends.astype(np.uint64); out = np.zeros((len(sizes), 72), np.uint64)
    assert lib().dgpu_multi_miller_loop_segments(p_(ps), p_(qs), p_(skip), n, p_(e64), len(sizes), p_(out)) == 0
    for k, (a, b) in enumerate(zip(starts, ends)):
        assert (out[k] == ca.multi_miller_loop(ps[a:b], qs[a:b], skip[a:b])).all(), k
    bad = e64.copy(); bad[-1] = n - 1 if n else 1
    assert lib().dgpu_multi_miller_loop_segments(p_(ps), p_(qs), None, n, p_(bad), len(sizes), p_(out)) == -3       # last end != n
    if len(sizes) > 2 and sizes[1]:
        bad = e64.copy(); bad[0], bad[1] = bad[1], bad[0]
        assert lib().dgpu_multi_miller_loop_segments(p_(ps), p_(qs), None, n, p_(bad), len(sizes), p_(out)) == -3   # not ascending
    assert lib().dgpu_multi_miller_loop_segments(p_(ps), p_(qs), None, n, p_(e64), 0, p_(out)) == -3
