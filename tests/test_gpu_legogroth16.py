"""GPU (-m gpu): LegoGroth16 prove -> verify round trip on a real (small) proving key, the shape of the reference's own
tests (legogroth16/src/tests.rs:149-354: prove, verify, tamper, verify fails).  Prover MSMs, the verifier's MSM, the
three-pair Miller loop and the final exponentiation all run through the C ABI; the proof elements are also compared
bit for bit with the same equations evaluated by the CPU oracle."""
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import lego_setup as LS
import crypto_amd as ca
from crypto_amd import legogroth16 as LG

pytestmark = pytest.mark.gpu
R = LS.R


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def oracle_lincomb(G, pts, scs):
    acc = None
    for p, s in zip(pts, scs):
        inf = not np.asarray(p).any()
        j = G.mul(np.asarray(p, dtype=np.uint64), O.int_to_limbs(s % R, 4), inf=inf)
        acc = j if acc is None else G.add(acc, j)
    return acc


def aff(G, jac):
    a, inf = G.to_affine(jac)
    return np.zeros_like(a) if inf else a


@pytest.mark.parametrize("m,cw", [(20, 2), (117, 3)])
def test_prove_verify_roundtrip(m, cw):
    cs = LS.circuit(m, x0=7)
    key = LS.setup(cs, cw, seed=m)
    h = LS.witness_map(cs)
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], cw)
    pk = LG.ProvingKey(vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"], key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    z = cs["z"]
    inp, wit = LS.scalars(z[:cs["n_inst"]]), LS.scalars(z[cs["n_inst"]:])
    r, s, v = 0x1234567 * 0x9E3779B97F4A7C15 % R, 0xABCDEF01 * 0xBF58476D1CE4E5B9 % R, 0x55AA55 * 0x94D049BB133111EB % R
    proof = LG.create_proof(pk, r, s, v, LS.scalars(h), inp, wit)
    pvk = LG.prepare_verifying_key(vk)
    assert LG.verify_proof(pvk, proof, inp[1:])
    # same equations on the CPU oracle: A, B, C, D must match limb for limb
    zs = z[1:]
    A = oracle_lincomb(O.G1, [key["alpha_g1"], key["a_query"][0], key["delta_g1"]] + list(key["a_query"][1:]), [1, 1, r] + zs)
    B = oracle_lincomb(O.G2, [key["beta_g2"], key["b_g2_query"][0], key["delta_g2"]] + list(key["b_g2_query"][1:]), [1, 1, s] + zs)
    B1 = oracle_lincomb(O.G1, [key["beta_g1"], key["b_g1_query"][0], key["delta_g1"]] + list(key["b_g1_query"][1:]), [1, 1, s] + zs)
    n = cs["n_inst"] + cw
    wv = z[cs["n_inst"]:]
    Cc = oracle_lincomb(O.G1, [aff(O.G1, A), aff(O.G1, B1), key["delta_g1"], key["eta_delta_inv_g1"]] + list(key["l_query"]) + list(key["h_query"]),
                        [s, r, -(r * s), -v] + wv[cw:] + h[:len(key["h_query"])])
    Dd = oracle_lincomb(O.G1, list(key["gamma_abc_g1"][cs["n_inst"]:n]) + [key["eta_gamma_inv_g1"]], wv[:cw] + [v])
    assert (proof["a"] == aff(O.G1, A)).all() and (proof["b"] == aff(O.G2, B)).all()
    assert (proof["c"] == aff(O.G1, Cc)).all() and (proof["d"] == aff(O.G1, Dd)).all()
    # tampering: wrong public input, swapped proof elements (tests.rs: verification must fail)
    bad_inp = inp[1:].copy(); bad_inp[0][0] ^= np.uint64(1)
    assert not LG.verify_proof(pvk, proof, bad_inp)
    bad = dict(proof); bad["c"] = proof["a"]
    assert not LG.verify_proof(pvk, bad, inp[1:])
    # r == 0 skips the G1 copy of B (prover.rs:330)
    proof0 = LG.create_proof(pk, 0, s, v, LS.scalars(h), inp, wit)
    assert LG.verify_proof(pvk, proof0, inp[1:])
    # the verifier as ONE call of the C ABI (dgpu_legogroth16_verify: calculate_d on a host core under the device's chain): the same answers,
    # in every form of the Miller kernels, for &[Fr] inputs too, from six threads at once; the reference's error cases
    from crypto_amd._native import lib

    def abi_checks():
        assert LG.verify_proof_abi(pvk, proof, inp[1:]) and LG.verify_proof_abi(pvk, proof0, inp[1:])
        assert LG.verify_proof_abi(pvk, proof, O.fr_to_mont(inp[1:]), montgomery=True)
        assert not LG.verify_proof_abi(pvk, proof, bad_inp) and not LG.verify_proof_abi(pvk, bad, inp[1:])
    abi_checks()                                    # the product library (every knob at its default)
    with ca.twin():                                 # the forms of the Miller kernels are a development knob (include/dock_gpu_dev.h)
        try:
            for mode in (31, 15, 7, 3, 6, 0):
                assert lib().dgpu_set_miller_pipeline(mode) == 0
                abi_checks()
        finally:
            lib().dgpu_set_miller_pipeline(31)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(6) as ex:
        res = list(ex.map(lambda k: LG.verify_proof_abi(pvk, proof if k % 3 else bad, inp[1:]), range(24)))
    assert res == [bool(k % 3) for k in range(24)]
    with pytest.raises(ValueError):                                   # more inputs than the key has room for: MalformedVerifyingKey
        LG.verify_proof_abi(pvk, proof, np.concatenate([inp[1:]] * 8 + [inp[1:]]))
    ident = dict(proof); ident["a"] = np.zeros(12, np.uint64)        # A = identity: the pair is skipped like arkworks does; the equation no longer holds
    assert LG.verify_proof_abi(pvk, ident, inp[1:]) == LG.verify_proof(pvk, ident, inp[1:])


@pytest.mark.parametrize("m,cw", [(20, 2), (200, 3)])
def test_generator_matches_oracle_setup(m, cw):
    """generate_parameters (device fixed-base products) == the oracle's setup for the same toxic waste, element by element;
    the generated key then proves and verifies (generator.rs:245-442; tests.rs:149-180 uses the generated key the same way)."""
    from crypto_amd import qap
    cs = LS.circuit(m, x0=11)
    key = LS.setup(cs, cw, seed=1000 + m)
    w = key["_waste"]
    g1 = O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(w["k1"], 4)))[0]
    g2 = O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(w["k2"], 4)))[0]
    pk, n_inst = LG.generate_parameters(cs["A"], cs["B"], cs["C"], cs["n_inst"], cs["n_wit"], cw,
                                        w["alpha"], w["beta"], w["gamma"], w["delta"], w["eta"], w["t"], g1, g2)
    assert n_inst == cs["n_inst"]
    vk = pk.vk
    for name in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "eta_gamma_inv_g1"):
        assert (getattr(vk, name) == key[name]).all(), name
    assert (vk.gamma_abc_g1 == key["gamma_abc_g1"]).all()
    for name in ("beta_g1", "delta_g1", "eta_delta_inv_g1"):
        assert (getattr(pk, name) == key[name]).all(), name
    assert (pk.a0 == key["a_query"][0]).all() and (pk.b1_0 == key["b_g1_query"][0]).all() and (pk.b2_0 == key["b_g2_query"][0]).all()
    # resident queries: same group elements as the oracle's arrays (random linear combination of all entries)
    for name, cv in (("a_query", ca.G1), ("b_g1_query", ca.G1), ("b_g2_query", ca.G2), ("h_query", ca.G1), ("l_query", ca.G1)):
        db = getattr(pk, name)
        assert db.n == len(key[name]), name
        rs = O.rand_scalars(77, db.n)
        inf = np.array([0 if row.any() else 1 for row in key[name]], dtype=np.uint8)
        assert (db.msm_bigint(rs) == ca.msm_bigint(cv, key[name], rs, is_inf=inf)).all(), name
    z = cs["z"]
    inp, wit = LS.scalars(z[:cs["n_inst"]]), LS.scalars(z[cs["n_inst"]:])
    h = LS.scalars(LS.witness_map(cs))
    proof = LG.create_proof(pk, 12345, 67890, 13579, h, inp, wit)
    assert LG.verify_proof(LG.prepare_verifying_key(vk), proof, inp[1:])


def _real_key(m, cw, seed):
    cs = LS.circuit(m, x0=5)
    key = LS.setup(cs, cw, seed=seed)
    vk = LG.VerifyingKey(key["alpha_g1"], key["beta_g2"], key["gamma_g2"], key["delta_g2"], key["gamma_abc_g1"], key["eta_gamma_inv_g1"], cw)
    pk = LG.ProvingKey(vk, key["beta_g1"], key["delta_g1"], key["eta_delta_inv_g1"], key["a_query"], key["b_g1_query"], key["b_g2_query"], key["h_query"], key["l_query"])
    z = cs["z"]
    return cs, key, vk, pk, LS.scalars(z[:cs["n_inst"]]), LS.scalars(z[cs["n_inst"]:]), LS.scalars(LS.witness_map(cs))


def test_commitment_openings_and_rerandomisation():
    """verify_witness_commitment (prover.rs:434-467), rerandomize_proof / rerandomize_proof_1 (:478-549): the shape of the reference's tests
    (legogroth16/src/tests.rs:181-214 — rerandomised proofs verify, the commitment opens to the committed witnesses with the right randomness only)"""
    cs, key, vk, pk, inp, wit, h = _real_key(60, 3, 31)
    r, s, v = 11 ** 20 % R, 13 ** 19 % R, 17 ** 18 % R
    proof = LG.create_proof(pk, r, s, v, h, inp, wit)
    pvk = LG.prepare_verifying_key(vk)
    assert LG.verify_proof(pvk, proof, inp[1:])
    wv = cs["z"][cs["n_inst"]:]
    LG.verify_witness_commitment(vk, proof, cs["n_inst"] - 1, wv[:3], v)
    for bad in ((wv[:2] + [wv[2] + 1], v), (wv[:3], v + 1)):
        with pytest.raises(ValueError):
            LG.verify_witness_commitment(vk, proof, cs["n_inst"] - 1, bad[0], bad[1])
    with pytest.raises(ValueError):
        LG.verify_witness_commitment(vk, proof, cs["n_inst"] - 1, wv[:40], v)            # VectorLongerThanExpected
    p2 = LG.rerandomize_proof(proof, vk, 0xABCDEF123457, 0x1234567ABCDEF1)
    assert LG.verify_proof(pvk, p2, inp[1:]) and not (p2["a"] == proof["a"]).all()
    new_v = 99 ** 17 % R
    p3 = LG.rerandomize_proof_1(proof, v, new_v, vk, pk.eta_delta_inv_g1, 0x5555AAAA5555, 0x777766665555)
    assert LG.verify_proof(pvk, p3, inp[1:])
    LG.verify_witness_commitment(vk, p3, cs["n_inst"] - 1, wv[:3], new_v)
    with pytest.raises(ValueError):
        LG.verify_witness_commitment(vk, p3, cs["n_inst"] - 1, wv[:3], v)
    # the same equations on the oracle: A' = A / r1, D' = D + (new_v - old_v)(eta/gamma)
    r1 = 0x5555AAAA5555
    assert (p3["a"] == aff(O.G1, oracle_lincomb(O.G1, [proof["a"]], [pow(r1, R - 2, R)]))).all()
    assert (p3["d"] == aff(O.G1, oracle_lincomb(O.G1, [proof["d"], key["eta_gamma_inv_g1"]], [1, new_v - v]))).all()


def test_cp_link_prove_and_verify():
    """CP_link (prover.rs:183-234, generator.rs:166-204, link/snark.rs): proof.d and link_d commit to the same witnesses; the subspace SNARK
    verifies, and fails for another link_v, another witness or a swapped commitment (legogroth16/src/tests.rs:88-147 shape)"""
    from crypto_amd import link as LK
    cs, key, vk, pk, inp, wit, h = _real_key(40, 2, 41)
    cw = 2
    g = lambda k: O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(k % R, 4)))[0]
    g2 = lambda k: O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(k % R, 4)))[0]
    gens = np.stack([g(1001 + 7 * i) for i in range(cw + 1)])
    pp, ek, lvk, bases = LG.generate_link_keys(vk, cs["n_inst"], gens, g(5), g2(7), [123456789, 987654321], 555555)
    # the evaluation key against the oracle: column c of m^T k
    col0 = oracle_lincomb(O.G1, [gens[0], key["gamma_abc_g1"][cs["n_inst"]]], [123456789, 987654321])
    assert (ek["p"][0] == aff(O.G1, col0)).all()
    r, s, v, link_v = 3 ** 40 % R, 5 ** 30 % R, 7 ** 25 % R, 11 ** 21 % R
    pl = LG.create_proof_incl_cp_link(pk, pp, ek, bases, r, s, v, link_v, h, inp, wit)
    assert LG.verify_proof(LG.prepare_verifying_key(vk), pl["groth16_proof"], inp[1:])
    LG.verify_link_proof(pp, lvk, pl)
    wv = cs["z"][cs["n_inst"]:]
    LG.verify_commitments(vk, bases, pl, cs["n_inst"] - 1, wv[:cw], v, link_v)
    assert (pl["link_d"] == aff(O.G1, oracle_lincomb(O.G1, list(gens), wv[:cw] + [link_v]))).all()
    with pytest.raises(ValueError):
        LG.verify_link_commitment(bases, pl["link_d"], wv[:cw], link_v + 1)
    bad = dict(pl); bad["link_d"] = gens[0]
    with pytest.raises(LK.LinkError):
        LG.verify_link_proof(pp, lvk, bad)
    bad = dict(pl); bad["link_pi"] = gens[1]
    with pytest.raises(LK.LinkError):
        LG.verify_link_proof(pp, lvk, bad)
    with pytest.raises(LK.LinkError):
        LK.prove(pp, ek, [1, 2, 3, 4, 5])                       # more witnesses than columns


@pytest.mark.parametrize("n", [1, 2, 9, 300])
def test_batch_of_proofs_in_one_call(n):
    """dgpu_legogroth16_verify_batch: N proofs of one key through the merged batch check inside the library — accepts exactly when every proof verifies
    (against the one-proof call and the Python statement of the same check), whatever the batching scalar; one bad proof, one wrong public input, a
    swapped member, an identity member and a zero batching scalar are all answered as the one-proof path answers them"""
    R = U.R
    rng = np.random.default_rng(50 + n)
    ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1 for _ in range(k)]
    al, be, ga, de, g0, g1x = ints(6)
    av, bv, dv, xv = ints(n), ints(n), ints(n), ints(n)
    dinv = pow(de, R - 2, R)
    cv = [((a * b - al * be - (g0 + x * g1x + d) * ga) * dinv) % R for a, b, d, x in zip(av, bv, dv, xv)]
    lim = lambda vals: np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
    from crypto_amd import fixed_base as FB
    with FB.WindowTable(ca.G2, O.G2.generator()) as t2, FB.WindowTable(ca.G1, O.G1.generator()) as t1:
        A_, _ = t1.multiply_many(lim(av)); C_, _ = t1.multiply_many(lim(cv)); D_, _ = t1.multiply_many(lim(dv)); K_, _ = t1.multiply_many(lim([al, g0, g1x, 1]))
        B_, _ = t2.multiply_many(lim(bv)); V_, _ = t2.multiply_many(lim([be, ga, de]))
    vk = LG.VerifyingKey(K_[0], V_[0], V_[1], V_[2], K_[1:3], K_[3], 0)
    pvk = LG.prepare_verifying_key(vk)
    proofs = [{"a": A_[i], "b": B_[i], "c": C_[i], "d": D_[i]} for i in range(n)]
    pubs = [lim([x]) for x in xv]
    assert all(LG.verify_proof_abi(pvk, proofs[i], pubs[i]) for i in range(min(n, 3)))
    for rnd in (1, 2, 0x5EED0025, R - 1):
        assert LG.verify_proofs_batch_abi(pvk, proofs, pubs, rnd)
    assert LG.verify_proofs_batch_abi(pvk, proofs, [O.fr_to_mont(x) for x in pubs], 77, montgomery=True)
    assert LG.verify_proofs_batch_merged(pvk, proofs, pubs, 0x5EED0025)
    j = n // 2
    bad = list(proofs); bad[j] = dict(bad[j], c=proofs[j]["a"])
    assert not LG.verify_proofs_batch_abi(pvk, bad, pubs, 0x5EED0026) and not LG.verify_proof_abi(pvk, bad[j], pubs[j])
    wrong = list(pubs); wrong[j] = lim([(xv[j] + 1) % R])
    assert not LG.verify_proofs_batch_abi(pvk, proofs, wrong, 0x5EED0027)
    ident = list(proofs); ident[j] = dict(ident[j], a=np.zeros(12, np.uint64))
    assert not LG.verify_proofs_batch_abi(pvk, ident, pubs, 0x5EED0028)
    if n >= 2:
        sw = list(proofs); sw[0], sw[1] = dict(sw[0], d=proofs[1]["d"]), dict(sw[1], d=proofs[0]["d"])
        assert not LG.verify_proofs_batch_abi(pvk, sw, pubs, 0x5EED0029)
    with pytest.raises(ca.DockGpuError):
        LG.verify_proofs_batch_abi(pvk, proofs, pubs, 0)
    with pytest.raises(ca.DockGpuError):
        LG.verify_proofs_batch_abi(pvk, proofs, pubs, R)
    assert LG.verify_proofs_batch_abi(pvk, [], [], 5)
    with pytest.raises(ValueError):
        LG.verify_proofs_batch_abi(pvk, proofs, [lim([x, x, x]) for x in xv], 5)            # more inputs than the key covers: MalformedVerifyingKey
