"""GPU (-m gpu): the aggregation inside the library (dgpu_snarkpack_aggregate / dgpu_snarkpack_verify, crypto_amd/csrc/dock_aggregation.cpp) against
the Python statement of the same protocol (crypto_amd/aggregation/groth16.py, itself checked against the CPU oracle in
tests/test_gpu_aggregation.py): every element of the aggregate proof identical under the same Merlin transcript, each side's proof accepted
by the other side's verifier, and the reference's rejection cases (/root/reference/legogroth16/src/aggregation/tests.rs:117-330: wrong public
input, wrong transcript, tampered proof parts) through the C ABI."""
import copy
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd import legogroth16 as LG
from crypto_amd import aggregation as AG
from crypto_amd.aggregation import native as NA, groth16
from test_gpu_aggregation import make_statement, g1, g2

pytestmark = pytest.mark.gpu
R = U.R


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def same_proof(x, y):
    wx, wy = NA.proof_to_words(x), NA.proof_to_words(y)
    return len(wx) == len(wy) and bool((wx == wy).all())


def rejects(fn):
    try:
        fn()
    except AG.AggregationError:
        return True
    return False


@pytest.mark.parametrize("n", [2, 8, 64])
def test_native_aggregate_equals_python(n):
    vk, proofs, inputs, _ = make_statement(n, 2, seed=100 + n)
    pk, vsrs = AG.setup_fake_srs(0xA11CE5EED + n, 0xBE7A5EED, n, O.G1.generator(), O.G2.generator()).specialize(n)
    label = b"native-aggregation"
    py = AG.aggregate_proofs(pk, AG.MerlinTranscript(label), proofs)
    words = NA.aggregate_proofs_words(pk, AG.MerlinTranscript(label), proofs)
    assert len(words) == ca._native.lib().dgpu_snarkpack_proof_words(n, 0)
    nat = NA.proof_from_words(words)
    assert (NA.proof_to_words(nat) == words).all()                      # the word layout round-trips
    assert same_proof(py, nat)
    pvk = {"vk": vk}
    rnd = 0x5EED7654321
    # each verifier accepts each prover's proof
    NA.verify_aggregate_proof(vsrs, pvk, inputs, words, rnd, AG.MerlinTranscript(label))
    NA.verify_aggregate_proof(vsrs, pvk, inputs, py, rnd, AG.MerlinTranscript(label), validate_gt=True)
    AG.verify_aggregate_proof(vsrs, pvk, inputs, nat, rnd, AG.MerlinTranscript(label))
    # the reference's rejection cases, through the C ABI
    V = lambda proof=py, pub=inputs, lab=label, **kw: NA.verify_aggregate_proof(vsrs, pvk, pub, proof, rnd, AG.MerlinTranscript(lab), **kw)
    bad_inputs = copy.deepcopy(inputs); bad_inputs[0][0] = (bad_inputs[0][0] + 1) % R
    assert rejects(lambda: V(pub=bad_inputs))
    assert rejects(lambda: V(lab=b"another-transcript"))
    for path, value in ((("z_c",), g1(4711)), (("tmipp", "gipa", "final_a"), g1(12345)), (("tmipp", "gipa", "final_b"), g2(12345)), (("tmipp", "gipa", "final_c"), g1(3))):
        bad = copy.deepcopy(py)
        node = bad
        for k in path[:-1]:
            node = node[k]
        node[path[-1]] = value
        assert rejects(lambda: V(proof=bad)), path
    bad = copy.deepcopy(py); bad["tmipp"]["wkey_opening"] = (g1(5), bad["tmipp"]["wkey_opening"][1])
    assert rejects(lambda: V(proof=bad))
    if n > 2:
        bad = copy.deepcopy(py); bad["tmipp"]["vkey_opening"] = (bad["tmipp"]["vkey_opening"][1], bad["tmipp"]["vkey_opening"][0])
        assert rejects(lambda: V(proof=bad))
        bad = copy.deepcopy(py); l, r = bad["tmipp"]["gipa"]["comms_ab"][1]; bad["tmipp"]["gipa"]["comms_ab"][1] = (r, l)
        assert rejects(lambda: V(proof=bad))
    # parsing_check: a length that is not a power of two, a truncated proof, the wrong variant, a row count that is not nproofs
    w2 = words.copy(); w2[0] = n + 1
    assert rejects(lambda: V(proof=w2))
    assert rejects(lambda: V(proof=words[:-1].copy()))
    assert rejects(lambda: V(proof=words, with_d=True))
    assert rejects(lambda: V(pub=inputs[:-1]))
    # MalformedVerifyingKey: a Groth16 key must have exactly one more element than there are public inputs
    assert rejects(lambda: V(pub=[row[:1] for row in inputs]))
    # Validate::Yes: a GT element outside the order-r subgroup is refused before any pairing work
    raw = np.asarray(O.multi_miller_loop(g1(3).reshape(1, 12), g2(5).reshape(1, 24)), dtype=np.uint64).reshape(72)
    bad = copy.deepcopy(py); bad["tmipp"]["gipa"]["z_ab"][0] = (raw, bad["tmipp"]["gipa"]["z_ab"][0][1])
    assert rejects(lambda: V(proof=bad, validate_gt=True))
    # ... and so is a G1 / G2 member that is not a point of the prime-order subgroup (the other half of Validate::Yes): a point of E(Fp) outside G1
    # (the cofactor is ~2^126: any point found by solving y^2 = x^3 + 4 is outside), a pair of coordinates that is not on the curve at all
    P = U.P
    x = next(x for x in range(2, 100) if pow((x ** 3 + 4) % P, (P - 1) // 2, P) == 1)
    y = pow((x ** 3 + 4) % P, (P + 1) // 4, P)
    assert (y * y - x ** 3 - 4) % P == 0
    outside = np.concatenate([U.fp_abi(x), U.fp_abi(y)])
    off_curve = np.concatenate([U.fp_abi(x), U.fp_abi((y + 1) % P)])
    for pt in (outside, off_curve):
        bad = copy.deepcopy(py); bad["tmipp"]["gipa"]["final_a"] = pt
        assert rejects(lambda: V(proof=bad, validate_points=True))
    assert (V(validate_gt=True, validate_points=True), True)[1]         # the honest proof passes with both halves of the validation on
    # the pairing checker's batching scalar must not be zero mod r (every equation after the first would drop out of the product)
    for zero in (0, R):
        with pytest.raises(AG.AggregationError):
            NA.verify_aggregate_proof(vsrs, pvk, inputs, py, zero, AG.MerlinTranscript(label))
    # one wrong proof inside the batch
    wrong = copy.deepcopy(proofs); wrong[1]["c"] = g1(777)
    assert rejects(lambda: V(proof=NA.aggregate_proofs(pk, AG.MerlinTranscript(label), wrong)))


def test_native_transcript_in_c_gives_the_same_proof():
    """the caller's transcript as C callbacks (merlin_native.c: no interpreter inside the library call) — the same proof words as with the Python
    transcript called back, each verifier accepts the other's transcript form, a wrong label is rejected"""
    n = 16
    vk, proofs, inputs, _ = make_statement(n, 1, seed=21)
    pk, vsrs = AG.setup_fake_srs(7, 11, n, O.G1.generator(), O.G2.generator()).specialize(n)
    pvk = {"vk": vk}
    label = b"native-transcript"
    w_py = NA.aggregate_proofs_words(pk, AG.MerlinTranscript(label), proofs)
    w_c = NA.aggregate_proofs_words(pk, AG.NativeMerlinTranscript(label), proofs)
    assert (w_py == w_c).all()
    NA.verify_aggregate_proof(vsrs, pvk, inputs, w_c, 0x1234567, AG.NativeMerlinTranscript(label), validate_gt=True, validate_points=True)
    NA.verify_aggregate_proof(vsrs, pvk, inputs, w_c, 0x1234567, AG.MerlinTranscript(label))
    with pytest.raises(AG.AggregationError):
        NA.verify_aggregate_proof(vsrs, pvk, inputs, w_c, 0x1234567, AG.NativeMerlinTranscript(b"another"))


def test_native_argument_checks():
    vk, proofs, inputs, _ = make_statement(4, 1, seed=9)
    pk, _ = AG.setup_fake_srs(3, 5, 4, O.G1.generator(), O.G2.generator()).specialize(4)
    for bad in (proofs[:1], proofs[:3], proofs[:2]):                    # < 2, not a power of two, SRS specialised for 4
        with pytest.raises(AG.AggregationError):
            NA.aggregate_proofs(pk, AG.MerlinTranscript(b"t"), bad)
    # the same three straight at the ABI
    import ctypes as C
    L = ca._native.lib()
    assert L.dgpu_snarkpack_proof_words(3, 0) == 0 and L.dgpu_snarkpack_proof_words(1, 0) == 0 and L.dgpu_snarkpack_proof_words(4, 0) > 0
    assert L.dgpu_snarkpack_aggregate(None, None, None, None, None, 4, None, None, 0, None) == -3
    # an exception inside the caller's transcript comes back as that exception, not as a crash of the C call
    class Broken(AG.MerlinTranscript):
        def challenge_scalar(self, label):
            raise KeyError("transcript broke")
    with pytest.raises(KeyError):
        NA.aggregate_proofs(pk, Broken(b"t"), proofs)


def test_native_legogroth16_variants():
    """the LegoGroth16 aggregator (one more MIPP, for d) and the Groth16 aggregator with the d_i shipped alongside (using_groth16.rs), on proofs
    with arbitrary d: the prover side is compared with the Python aggregator, the verifier on consistent statements built from discrete logs"""
    n = 8
    rng = np.random.default_rng(77)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
    vk, proofs, inputs, dlogs = make_statement(n, 1, seed=5)
    pk, vsrs = AG.setup_fake_srs(rnd(), rnd(), n, O.G1.generator(), O.G2.generator()).specialize(n)
    for p in proofs:
        p["d"] = g1(rnd())
    label = b"native-lego"
    from crypto_amd.aggregation import legogroth16 as AL, using_groth16 as UG
    py = AL.aggregate_proofs(pk, AG.MerlinTranscript(label), proofs)
    nat = NA.aggregate_proofs(pk, AG.MerlinTranscript(label), proofs, with_d=True)
    assert "com_d" in nat and same_proof(py, nat)
    # (the statement is not a valid LegoGroth16 one — d is arbitrary —, so both verifiers must say no, and they must agree)
    pvk = {"vk": vk}
    assert rejects(lambda: AL.verify_aggregate_proof(vsrs, pvk, inputs, py, 5, AG.MerlinTranscript(label)))
    assert rejects(lambda: NA.verify_aggregate_proof(vsrs, pvk, inputs, nat, 5, AG.MerlinTranscript(label), with_d=True))
    py_g, ds = UG.aggregate_proofs(pk, AG.MerlinTranscript(label), proofs)
    nat_g = NA.aggregate_proofs(pk, AG.MerlinTranscript(label), proofs)
    assert same_proof(py_g, nat_g)
    assert rejects(lambda: NA.verify_aggregate_proof(vsrs, pvk, inputs, nat_g, 5, AG.MerlinTranscript(label), d=ds))


def test_native_real_legogroth16_proofs():
    """four LegoGroth16 proofs of one circuit (tests/test_gpu_aggregation.py::test_aggregate_real_legogroth16_proofs' statement) through both
    library variants: accepted; a wrong public input, a tampered z_d / final_d, a wrong d in the shipped list: rejected"""
    import lego_setup as LS
    m, cw, n = 20, 2, 4
    cs0 = LS.circuit(m, x0=3)
    rng = np.random.default_rng(5)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
    pk, _ = LG.generate_parameters(cs0["A"], cs0["B"], cs0["C"], cs0["n_inst"], cs0["n_wit"], cw, rnd(), rnd(), rnd(), rnd(), rnd(), rnd(),
                                   O.G1.generator(), O.G2.generator())
    pvk = LG.prepare_verifying_key(pk.vk)
    proofs, inputs = [], []
    for j in range(n):
        cs = LS.circuit(m, x0=100 + j)
        z = cs["z"]
        inp, wit = LS.scalars(z[:cs["n_inst"]]), LS.scalars(z[cs["n_inst"]:])
        proofs.append(LG.create_proof(pk, rnd(), rnd(), rnd(), LS.scalars(LS.witness_map(cs)), inp, wit)); inputs.append([z[1]])
    psrs, vsrs = AG.setup_fake_srs(rnd(), rnd(), n, O.G1.generator(), O.G2.generator()).specialize(n)
    agg = NA.aggregate_proofs(psrs, AG.MerlinTranscript(b"lego"), proofs, with_d=True)
    from crypto_amd.aggregation import legogroth16 as AL
    assert same_proof(agg, AL.aggregate_proofs(psrs, AG.MerlinTranscript(b"lego"), proofs))
    NA.verify_aggregate_proof(vsrs, pvk, inputs, agg, rnd(), AG.MerlinTranscript(b"lego"), with_d=True)
    AL.verify_aggregate_proof(vsrs, pvk, inputs, agg, rnd(), AG.MerlinTranscript(b"lego"))
    bad_inputs = copy.deepcopy(inputs); bad_inputs[2][0] = (bad_inputs[2][0] + 1) % R
    assert rejects(lambda: NA.verify_aggregate_proof(vsrs, pvk, bad_inputs, agg, rnd(), AG.MerlinTranscript(b"lego"), with_d=True))
    for key, where in (("z_d", None), ("final_d", "gipa")):
        bad = copy.deepcopy(agg)
        (bad["tmipp"]["gipa"] if where else bad)[key] = g1(99)
        assert rejects(lambda: NA.verify_aggregate_proof(vsrs, pvk, inputs, bad, rnd(), AG.MerlinTranscript(b"lego"), with_d=True)), key
    ds = np.stack([p["d"] for p in proofs])
    agg_g = NA.aggregate_proofs(psrs, AG.MerlinTranscript(b"lego-g16"), proofs)
    NA.verify_aggregate_proof(vsrs, pvk, inputs, agg_g, rnd(), AG.MerlinTranscript(b"lego-g16"), d=ds)
    bad_ds = ds.copy(); bad_ds[1] = g1(4242)
    assert rejects(lambda: NA.verify_aggregate_proof(vsrs, pvk, inputs, agg_g, rnd(), AG.MerlinTranscript(b"lego-g16"), d=bad_ds))
    assert rejects(lambda: NA.verify_aggregate_proof(vsrs, pvk, bad_inputs, agg_g, rnd(), AG.MerlinTranscript(b"lego-g16"), d=ds))
    # the Groth16 verifier must not accept a Lego aggregate (the variant is part of the proof words)
    assert rejects(lambda: NA.verify_aggregate_proof(vsrs, pvk, inputs, agg, rnd(), AG.MerlinTranscript(b"lego")))
