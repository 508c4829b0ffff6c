"""CPU: pins the C oracle (oracle/oracle.c) against the golden fixtures produced by the independent
big-integer model (tests/golden/gen_golden.py), and re-asserts the algebraic identities the reference's own
tests use for this path (utils/src/msm.rs:186-193 msm == sum of mul_bigint; :268-275 pairing equalities)."""
import numpy as np
import pytest
import oracle_c as O
import bls12_381_model as M
import util as U


@pytest.mark.parametrize("name,G", [("g1_msm", O.G1), ("g2_msm", O.G2)])
def test_msm_golden(name, G):
    for case in U.load(name):
        bases, inf, sc, exp = U.case_arrays(case)
        for th in (1, 4):
            got = U.jac_to_model(G, G.msm(bases, sc, inf, threads=th))
            assert got == exp, (name, case["n"], case.get("note"))


def test_digits_and_window_rule():
    for d in U.load("digits"):
        got = O.make_digits(O.int_to_limbs(int(d["scalar"], 16), 4), d["c"])
        assert list(got) == d["digits"]
        assert sum(int(x) << (d["c"] * i) for i, x in enumerate(got)) == int(d["scalar"], 16)
    for n, c in U.load("window_c"):
        assert O.window_c(n) == c


def test_msm_equals_sum_of_scalar_muls():
    # utils/src/msm.rs:186-193 — G1::msm([g1, g2], [e1, e2]) == g1*e1 + g2*e2
    for G in (O.G1, O.G2):
        bases, _, _ = U.seq_bases(G, 2, 42, threads=1)
        sc = O.rand_scalars(43, 2)
        lhs = G.msm(bases, sc)
        rhs = G.add(G.mul(bases[0], sc[0]), G.mul(bases[1], sc[1]))
        assert U.jac_to_model(G, lhs) == U.jac_to_model(G, rhs)


def test_msm_closed_form_medium():
    for G, n in ((O.G1, 3000), (O.G2, 300)):
        bases, k0, d = U.seq_bases(G, n, 7)
        sc = O.rand_scalars(9, n)
        assert U.jac_to_model(G, G.msm(bases, sc, threads=4)) == U.closed_form(G, sc, k0, d)


def test_fr_montgomery_roundtrip():
    sc = O.rand_scalars(3, 50)
    m = O.fr_to_mont(sc)
    assert (O.fr_from_mont(m) == sc).all()
    assert O.limbs_to_int(m[0]) == O.limbs_to_int(sc[0]) * M.FR_R % M.R


def test_pairing_golden():
    pr = U.load("pairing")
    g1, g2 = O.G1.generator(), O.G2.generator()
    ml = O.multi_miller_loop(g1.reshape(1, 12), g2.reshape(1, 24))
    assert U.f12_ints(ml) == [int(v, 16) for v in pr["miller_g1_g2"]]
    e = O.final_exponentiation(ml)
    assert U.f12_ints(e) == [int(v, 16) for v in pr["e_g1_g2"]]
    co = O.g2_prepare(g2)
    assert pr["n_coeffs"] == 68
    assert [U.fp_int(x) for x in co[0].reshape(6, 6)] == [int(v, 16) for v in pr["g2_prepared_gen_first"][0]]
    assert [U.fp_int(x) for x in co[-1].reshape(6, 6)] == [int(v, 16) for v in pr["g2_prepared_gen_last"][0]]
    for case in pr["cases"]:
        ps = np.stack([U.g1_abi(U.dec_g1(p))[0] for p in case["p"]])
        qs = np.stack([U.g2_abi(U.dec_g2(q))[0] for q in case["q"]])
        for th in (1, 3):
            f = O.multi_miller_loop(ps, qs, threads=th)
            assert U.f12_ints(f) == [int(v, 16) for v in case["miller"]]
        assert U.f12_ints(O.final_exponentiation(f)) == [int(v, 16) for v in case["gt"]]


def test_pairing_identities():
    g1, g2 = O.G1.generator(), O.G2.generator()
    e = O.final_exponentiation(O.multi_miller_loop(g1.reshape(1, 12), g2.reshape(1, 24)))
    # bilinearity with skip handling: prod e(a_i G1, b_i G2) == e(G1, G2)^(sum a_i b_i), skipped pair excluded
    a = [3, 5, 7, 11, 13]
    b = [2, 4, 6, 8, 10]
    ps = np.stack([O.G1.to_affine(O.G1.mul(g1, O.int_to_limbs(x, 4)))[0] for x in a])
    qs = np.stack([O.G2.to_affine(O.G2.mul(g2, O.int_to_limbs(x, 4)))[0] for x in b])
    skip = np.array([0, 0, 1, 0, 0], np.uint8)
    gt = O.final_exponentiation(O.multi_miller_loop(ps, qs, skip, threads=2))
    tot = sum(x * y for i, (x, y) in enumerate(zip(a, b)) if i != 2)
    assert (gt == O.fp12_pow(e, tot)).all()
    # e(P, Q) * e(-P, Q) == 1
    negp = ps[0].copy()
    negp[6:] = U.fp_abi((-U.fp_int(ps[0][6:])) % U.P)
    f = O.multi_miller_loop(np.stack([ps[0], negp]), np.stack([qs[0], qs[0]]))
    assert (O.final_exponentiation(f) == O.fp12_one()).all()
    # arkworks returns None for f == 0
    assert O.final_exponentiation(np.zeros(72, np.uint64)) is None
    with pytest.raises(AssertionError):
        O.multi_miller_loop(ps[:2], qs[:3])


def test_oracle_ntt_and_witness_map_vs_naive_bigint():
    """pins oracle.c's restatement of ark-poly's radix-2 domain and of r1cs_to_qap.rs:150-210 against plain big-integer
    DFT sums and polynomial division (tests/lego_setup.py)"""
    import ctypes as C
    import random
    import lego_setup as LS
    L = O.lib(); L.orc_witness_map.restype = C.c_int
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    R = LS.R
    random.seed(1)
    for logn in (1, 3, 6):
        n = 1 << logn
        v = [random.randrange(R) for _ in range(n)]
        om = pow(7, (R - 1) // n, R)
        arr = LS.scalars(v).copy(); L.orc_fr_ntt(p(arr), logn, 0, 0)
        assert [O.limbs_to_int(x) for x in arr] == [sum(v[i] * pow(om, i * k, R) for i in range(n)) % R for k in range(n)]
        L.orc_fr_ntt(p(arr), logn, 1, 0); assert [O.limbs_to_int(x) for x in arr] == v
        arr = LS.scalars(v).copy(); L.orc_fr_ntt(p(arr), logn, 0, 1)       # coset g = 7
        assert [O.limbs_to_int(x) for x in arr] == [sum(v[i] * pow(7 * pow(om, k, R) % R, i, R) for i in range(n)) % R for k in range(n)]
        L.orc_fr_ntt(p(arr), logn, 1, 1); assert [O.limbs_to_int(x) for x in arr] == v
    from crypto_amd.qap import csr
    for m in (1, 7, 40):
        cs = LS.circuit(m, 7)
        mats = [csr(cs[k]) for k in "ABC"]
        D = 1
        while D < cs["n_cons"] + cs["n_inst"]:
            D *= 2
        out = np.zeros((D, 4), np.uint64)
        args = []
        for rp, cl, vl in mats:
            args += [p(rp), p(cl), p(vl)]
        L.orc_witness_map(*args, p(LS.scalars(cs["z"])), C.c_size_t(len(cs["z"])), C.c_size_t(cs["n_inst"]), C.c_size_t(cs["n_cons"]), p(out))
        assert [O.limbs_to_int(x) for x in out] == LS.witness_map(cs)
        # the threaded form bench.py times as the CPU leg of the witness map / prover: the same coefficients on any thread count
        for T in (2, 3, 8):
            assert (O.witness_map(mats, LS.scalars(cs["z"]), cs["n_inst"], cs["n_cons"], threads=T) == out).all(), (m, T)


def test_oracle_mixed_miller_loop_equals_the_affine_one():
    """orc_multi_miller_loop_mixed (the verifier's call shape: some G2 members already prepared) == orc_multi_miller_loop on the same pairs"""
    pr = U.load("pairing")
    k = O.rand_scalars(41, 7)
    P = np.stack([O.G1.to_affine(O.G1.mul(O.G1.generator(), k[i]))[0] for i in range(7)])
    Q = np.stack([O.G2.to_affine(O.G2.mul(O.G2.generator(), k[6 - i]))[0] for i in range(7)])
    ref = O.multi_miller_loop(P, Q)
    for na in (0, 1, 3, 7):
        co = np.stack([O.g2_prepare(q).reshape(-1) for q in Q[na:]]) if na < 7 else np.zeros((0, 68 * 36), np.uint64)
        for T in (1, 2):
            assert (O.multi_miller_loop_mixed(P[:na], Q[:na], P[na:], co, threads=T) == ref).all(), (na, T)


def test_final_exponentiation_chain_equals_its_definition():
    """The addition chain of the final exponentiation (restated in both oracles and in the product's host code) against plain
    big-integer exponentiation: it raises to 3 (p^12 - 1) / r — the CUBE of the canonical exponent, which is why GT bytes differ from
    libraries that use (p^12 - 1) / r itself — and to neither 1x nor 2x that exponent.  Ties the stored e(G1, G2) to its definition."""
    import bls12_381_model as M
    pr = U.load("pairing")
    f = M.multi_miller_loop([M.G1_GEN], [M.G2_GEN])
    e = M.final_exponentiation(f)
    d, rem = divmod(M.P ** 12 - 1, M.R)
    assert rem == 0
    assert M.f12_pow(f, 3 * d) == e
    assert M.f12_pow(f, d) != e and M.f12_pow(f, 2 * d) != e
    assert M.f12_pow(e, M.R) == M.f12_pow(e, 0)                 # the result has order dividing r
    # the C oracle's value (what the GPU tests compare against) is the same element
    g1, g2 = O.G1.generator(), O.G2.generator()
    ec = O.final_exponentiation(O.multi_miller_loop(g1.reshape(1, 12), g2.reshape(1, 24)))
    assert U.f12_ints(ec) == [int(v, 16) for v in pr["e_g1_g2"]]


def test_threaded_batches_equal_the_single_calls():
    """orc_g1_scale_batch / orc_g2_prepare_batch / orc_fp12_multi_pow (bench.py's CPU legs beside the batched verifier and the aggregation) are the
    one-at-a-time oracle functions run over the host's cores: same values whatever the thread count"""
    n = 37
    k0 = O.rand_scalars(901, 1)[0]; d = O.rand_scalars(902, 1)[0]
    P = O.G1.gen_seq(k0, d, n); Q = O.G2.gen_seq(d, k0, n)
    sc = O.rand_scalars(903, n)
    neg = (np.arange(n) % 3 == 0).astype(np.uint8); inf = np.zeros(n, np.uint8); inf[5] = 1
    for thr in (1, 4, 64):
        out, oinf = O.g1_scale_batch(P, sc, negate=neg, is_inf=inf, threads=thr)
        for i in range(n):
            ref, rinf = O.G1.to_affine(O.G1.mul(P[i], sc[i], inf=bool(inf[i])))
            assert oinf[i] == rinf
            if not rinf:
                assert (out[i][:6] == ref[:6]).all()
                y = U.fp_int(ref[6:]); want = U.fp_abi((-y) % U.P if neg[i] else y)
                assert (out[i][6:] == want).all(), i
        same, _ = O.g1_scale_batch(P, sc[2], threads=thr)                                   # one scalar for every point
        assert (same[9] == O.G1.to_affine(O.G1.mul(P[9], sc[2]))[0]).all()
        pb = O.g2_prepare_batch(Q, threads=thr)
        assert all((pb[i] == O.g2_prepare(Q[i]).reshape(-1)).all() for i in (0, 7, n - 1))
    g = O.final_exponentiation(O.multi_miller_loop(P[:2], Q[:2]))
    fs = np.stack([g, O.fp12_mul(g, g), O.fp12_mul(O.fp12_mul(g, g), g)])
    want_same = O.fp12_one(); want_each = O.fp12_one()
    for i in range(3):
        want_same = O.fp12_mul(want_same, O.fp12_pow(g, O.limbs_to_int(sc[i])))
        want_each = O.fp12_mul(want_each, O.fp12_pow(fs[i], O.limbs_to_int(sc[i])))
    for thr in (1, 2, 8):
        assert (O.fp12_multi_pow(g, sc[:3], threads=thr) == want_same).all()
        assert (O.fp12_multi_pow(fs, sc[:3], threads=thr) == want_each).all()
