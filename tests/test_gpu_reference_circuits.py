"""GPU (-m gpu): the reference's LegoGroth16 unit tests on its own three toy circuits (legogroth16/src/tests.rs:27-131, 149-566): the same
sequence of calls and the same accept / reject expectations, for every commit_witness_count the reference runs — including 0 and "all
witnesses committed" (an empty l_query), domains of 2 and 4 points, and one or two public inputs."""
import numpy as np
import pytest
import torch
import oracle_c as O
import lego_setup as LS
import crypto_amd as ca
from crypto_amd import qap, legogroth16 as LG, link as LK

pytestmark = pytest.mark.gpu
R = LS.R


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def _rnd(rng):
    return int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1


g1 = lambda k: O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(k % R, 4)))[0]
g2 = lambda k: O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(k % R, 4)))[0]


def silly(a, b):
    """MySillyCircuit (tests.rs:51-70): witnesses a, b; public c = a b; one constraint.  z = [1, c, a, b]"""
    return {"A": [[(1, 2)]], "B": [[(1, 3)]], "C": [[(1, 1)]], "z": [1, a * b % R, a, b], "n_inst": 2, "n_wit": 2, "n_cons": 1}


def less_silly(a, b, c, d):
    """MyLessSillyCircuit (tests.rs:72-110): witnesses a, b, c, d, e = ab, f = cd; public y = e + f.  z = [1, y, a, b, c, d, e, f]"""
    e, f = a * b % R, c * d % R
    return {"A": [[(1, 2)], [(1, 4)], [(1, 6), (1, 7)]], "B": [[(1, 3)], [(1, 5)], [(1, 0)]], "C": [[(1, 6)], [(1, 7)], [(1, 1)]],
            "z": [1, (e + f) % R, a, b, c, d, e, f], "n_inst": 2, "n_wit": 6, "n_cons": 3}


def less_silly_1(a, b, c, d):
    """MyLessSillyCircuit1 (tests.rs:112-130): witnesses a, b, c, d; public e = ab, f = cd.  z = [1, e, f, a, b, c, d]"""
    return {"A": [[(1, 3)], [(1, 5)]], "B": [[(1, 4)], [(1, 6)]], "C": [[(1, 1)], [(1, 2)]],
            "z": [1, a * b % R, c * d % R, a, b, c, d], "n_inst": 3, "n_wit": 4, "n_cons": 2}


def _params(cs, cw, rng, with_link=True):
    pk, n_inst = LG.generate_parameters(cs["A"], cs["B"], cs["C"], cs["n_inst"], cs["n_wit"], cw, *[_rnd(rng) for _ in range(6)], g1(_rnd(rng)), g2(_rnd(rng)))
    link = None
    if with_link:
        gens = np.stack([g1(_rnd(rng)) for _ in range(cw + 1)])
        link = LG.generate_link_keys(pk.vk, n_inst, gens, g1(_rnd(rng)), g2(_rnd(rng)), [_rnd(rng), _rnd(rng)], _rnd(rng))
    return pk, link


def _fails(fn, *a):
    try:
        fn(*a)
    except (ValueError, LK.LinkError):
        return True
    return False


def _prove(pk, cs, r, s, v):
    z = LS.scalars(cs["z"])
    circ = qap.DeviceR1cs(*[qap.csr(cs[k]) for k in "ABC"], len(cs["z"]), cs["n_inst"], cs["n_cons"])
    try:
        return LG.create_proof_with_reduction(pk, circ, r, s, v, z), z
    finally:
        circ.free()


@pytest.mark.parametrize("cw", [2, 1, 0])
def test_prove_and_verify(cw):
    """tests.rs:149-354"""
    rng = np.random.default_rng(cw)
    shape = silly(1, 1)
    with pytest.raises(ValueError, match="InsufficientWitnessesForCommitment"):
        LG.generate_parameters(shape["A"], shape["B"], shape["C"], 2, 2, 3, *[_rnd(rng) for _ in range(6)], g1(5), g2(7))
    pk, (pp, ek, lvk, bases) = _params(shape, cw, rng)
    vk, pvk = pk.vk, LG.prepare_verifying_key(pk.vk)
    for _ in range(3):
        a, b = _rnd(rng), _rnd(rng)
        cs = silly(a, b)
        pub = LS.scalars([cs["z"][1]]); none = np.zeros((0, 4), np.uint64)
        r, s, v, link_v = (_rnd(rng) for _ in range(4))
        proof, z = _prove(pk, cs, r, s, v)
        h = LS.scalars(LS.witness_map(cs))
        pl = LG.create_proof_incl_cp_link(pk, pp, ek, bases, r, s, v, link_v, h, z[:2], z[2:])
        assert all((pl["groth16_proof"][k] == proof[k]).all() for k in proof)
        wit = [a, b][:cw]
        LG.verify_commitments(vk, bases, pl, 1, wit, v, link_v)
        LG.verify_witness_commitment(vk, proof, 1, wit, v)
        if cw == 2:
            assert _fails(LG.verify_commitments, vk, bases, pl, 1, [a], v, link_v)
            assert _fails(LG.verify_commitments, vk, bases, pl, 2, [a, b], v, link_v)
            assert _fails(LG.verify_witness_commitment, vk, proof, 1, [b, a], v)
            assert _fails(LG.verify_witness_commitment, vk, proof, 1, [a], v)
            assert _fails(LG.verify_witness_commitment, vk, proof, 2, [], v)
        if cw == 1:
            assert _fails(LG.verify_witness_commitment, vk, proof, 1, [a, b], v)
            assert _fails(LG.verify_witness_commitment, vk, proof, 2, [a], v)
        if cw == 0:
            assert _fails(LG.verify_witness_commitment, vk, proof, 1, [a, b], v)
            assert _fails(LG.verify_witness_commitment, vk, proof, 1, [a], v)
            assert _fails(LG.verify_witness_commitment, vk, proof, 2, [], v)
        assert LG.verify_proof(pvk, pl["groth16_proof"], pub); LG.verify_link_proof(pp, lvk, pl)
        assert LG.verify_proof(pvk, proof, pub)
        assert not LG.verify_proof(pvk, proof, none)                       # (the reference: Err)
        p2 = LG.rerandomize_proof(proof, vk, _rnd(rng), _rnd(rng))
        assert LG.verify_proof(pvk, p2, pub)
        assert _fails(LG.verify_witness_commitment, vk, p2, 1, wit, v)     # rerandomize_proof does not keep D as a commitment
        new_v = _rnd(rng)
        p3 = LG.rerandomize_proof_1(proof, v, new_v, vk, pk.eta_delta_inv_g1, _rnd(rng), _rnd(rng))
        assert LG.verify_proof(pvk, p3, pub)
        LG.verify_witness_commitment(vk, p3, 1, wit, new_v)                 # rerandomize_proof_1 does
        assert _fails(LG.verify_witness_commitment, vk, p3, 1, wit, v)
        if cw == 2:
            assert _fails(LG.verify_witness_commitment, vk, p3, 1, [b, a], new_v)


def test_prove_and_verify_1():
    """tests.rs:356-460: MyLessSillyCircuit, all four input witnesses committed"""
    rng = np.random.default_rng(11)
    pk, (pp, ek, lvk, bases) = _params(less_silly(1, 1, 1, 1), 4, rng)
    vk, pvk = pk.vk, LG.prepare_verifying_key(pk.vk)
    for _ in range(3):
        a, b, c, d = (_rnd(rng) for _ in range(4))
        cs = less_silly(a, b, c, d)
        pub = LS.scalars([cs["z"][1]])
        r, s, v, link_v = (_rnd(rng) for _ in range(4))
        proof, z = _prove(pk, cs, r, s, v)
        pl = LG.create_proof_incl_cp_link(pk, pp, ek, bases, r, s, v, link_v, LS.scalars(LS.witness_map(cs)), z[:2], z[2:])
        LG.verify_commitments(vk, bases, pl, 1, [a, b, c, d], v, link_v)
        assert _fails(LG.verify_commitments, vk, bases, pl, 0, [a, b, c, d], v, link_v)
        assert _fails(LG.verify_commitments, vk, bases, pl, 1, [a, b, c], v, link_v)
        LG.verify_witness_commitment(vk, proof, 1, [a, b, c, d], v)
        assert _fails(LG.verify_witness_commitment, vk, proof, 0, [a, b, c, d], v)
        assert _fails(LG.verify_witness_commitment, vk, proof, 1, [a, b, c], v)
        assert LG.verify_proof(pvk, pl["groth16_proof"], pub); LG.verify_link_proof(pp, lvk, pl)
        assert LG.verify_proof(pvk, proof, pub) and not LG.verify_proof(pvk, proof, np.zeros((0, 4), np.uint64))
        assert LG.verify_proof(pvk, LG.rerandomize_proof(proof, vk, _rnd(rng), _rnd(rng)), pub)
        new_v = _rnd(rng)
        p3 = LG.rerandomize_proof_1(proof, v, new_v, vk, pk.eta_delta_inv_g1, _rnd(rng), _rnd(rng))
        assert LG.verify_proof(pvk, p3, pub)
        LG.verify_witness_commitment(vk, p3, 1, [a, b, c, d], new_v)


def test_prove_and_verify_2():
    """tests.rs:462-566: MyLessSillyCircuit1, two public inputs, all four witnesses committed (no l_query entry is left)"""
    rng = np.random.default_rng(12)
    pk, (pp, ek, lvk, bases) = _params(less_silly_1(1, 1, 1, 1), 4, rng)
    assert pk.l_query.n == 0
    vk, pvk = pk.vk, LG.prepare_verifying_key(pk.vk)
    for _ in range(3):
        a, b, c, d = (_rnd(rng) for _ in range(4))
        cs = less_silly_1(a, b, c, d)
        pub = LS.scalars(cs["z"][1:3])
        r, s, v, link_v = (_rnd(rng) for _ in range(4))
        proof, z = _prove(pk, cs, r, s, v)
        pl = LG.create_proof_incl_cp_link(pk, pp, ek, bases, r, s, v, link_v, LS.scalars(LS.witness_map(cs)), z[:3], z[3:])
        LG.verify_commitments(vk, bases, pl, 2, [a, b, c, d], v, link_v)
        assert _fails(LG.verify_commitments, vk, bases, pl, 1, [a, b, c, d], v, link_v)
        LG.verify_witness_commitment(vk, proof, 2, [a, b, c, d], v)
        assert _fails(LG.verify_witness_commitment, vk, proof, 1, [a, b, c, d], v)
        assert _fails(LG.verify_witness_commitment, vk, proof, 2, [a, b, c], v)
        assert LG.verify_proof(pvk, pl["groth16_proof"], pub); LG.verify_link_proof(pp, lvk, pl)
        assert LG.verify_proof(pvk, proof, pub)
        assert not LG.verify_proof(pvk, proof, pub[:1]) and not LG.verify_proof(pvk, proof, np.zeros((0, 4), np.uint64))
        assert LG.verify_proof(pvk, LG.rerandomize_proof(proof, vk, _rnd(rng), _rnd(rng)), pub)
        new_v = _rnd(rng)
        p3 = LG.rerandomize_proof_1(proof, v, new_v, vk, pk.eta_delta_inv_g1, _rnd(rng), _rnd(rng))
        assert LG.verify_proof(pvk, p3, pub)
        LG.verify_witness_commitment(vk, p3, 2, [a, b, c, d], new_v)


def test_batch_verification_through_the_pairing_checker():
    """verify_proofs_batch: many proofs of one circuit in one lazy RandomizedPairingChecker with the prepared verifying key's operands mixed in
    (proof_system/src/verifier.rs:1829-1835 shape): accepts the honest batch, rejects a batch with one swapped element / one wrong public input;
    calculate_d_batch equals calculate_d proof by proof."""
    rng = np.random.default_rng(21)
    pk, _ = _params(less_silly_1(1, 1, 1, 1), 2, rng, with_link=False)
    pvk = LG.prepare_verifying_key(pk.vk)
    proofs, pubs = [], []
    for _ in range(24):
        cs = less_silly_1(*(_rnd(rng) for _ in range(4)))
        proof, z = _prove(pk, cs, _rnd(rng), _rnd(rng), _rnd(rng))
        proofs.append(proof); pubs.append(z[1:3])
    ds = LG.calculate_d_batch(pvk, proofs, pubs)
    assert all((ds[i] == LG.calculate_d(pvk, proofs[i], pubs[i])).all() for i in range(len(proofs)))
    assert LG.verify_proofs_batch(pvk, proofs, pubs, _rnd(rng))
    bad = list(proofs); bad[7] = dict(bad[7]); bad[7]["c"] = proofs[8]["c"]
    assert not LG.verify_proofs_batch(pvk, bad, pubs, _rnd(rng))
    badp = list(pubs); badp[3] = pubs[4]
    assert not LG.verify_proofs_batch(pvk, proofs, badp, _rnd(rng))
    assert LG.verify_proofs_batch(pvk, proofs[:1], pubs[:1], _rnd(rng))


def test_merged_batch_verification_equals_the_checker():
    """verify_proofs_batch_merged (N + 2 pairs, two MSMs) accepts and rejects exactly like the 3 N-pair checker form"""
    rng = np.random.default_rng(22)
    pk, _ = _params(less_silly_1(1, 1, 1, 1), 2, rng, with_link=False)
    pvk = LG.prepare_verifying_key(pk.vk)
    proofs, pubs = [], []
    for _ in range(40):
        cs = less_silly_1(*(_rnd(rng) for _ in range(4)))
        proof, z = _prove(pk, cs, _rnd(rng), _rnd(rng), _rnd(rng))
        proofs.append(proof); pubs.append(z[1:3])
    assert LG.verify_proofs_batch_merged(pvk, proofs, pubs, _rnd(rng)) and LG.verify_proofs_batch(pvk, proofs, pubs, _rnd(rng))
    assert LG.verify_proofs_batch_merged(pvk, proofs[:1], pubs[:1], _rnd(rng)) and LG.verify_proofs_batch_merged(pvk, [], [], 5)
    for field in ("a", "b", "c", "d"):
        bad = list(proofs); bad[11] = dict(bad[11]); bad[11][field] = proofs[12][field]
        assert not LG.verify_proofs_batch_merged(pvk, bad, pubs, _rnd(rng)), field
    badp = list(pubs); badp[30] = pubs[31]
    assert not LG.verify_proofs_batch_merged(pvk, proofs, badp, _rnd(rng))
    with pytest.raises(ValueError):
        LG.verify_proofs_batch_merged(pvk, proofs, pubs[:-1], 7)
