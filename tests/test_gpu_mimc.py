"""GPU (-m gpu): the reference's own LegoGroth16 integration test (legogroth16/tests/mimc.rs:149-290) on this backend — knowledge of a MiMC
preimage (322 rounds, 644 constraints, the image public, xl / xr committed): parameters with and without CP_link, proofs with the witness
map on the device, the prover's commitment openings, verification, both re-randomisations, and the failure cases.  The reference runs it
on BLS12-377 with `StdRng`; here the curve is BLS12-381 (the only curve of this backend) and the randomness is seeded integers."""
import numpy as np
import pytest
import torch
import oracle_c as O
import lego_setup as LS
import mimc_circuit as MC
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


def test_mimc_legogroth16():
    rng = np.random.default_rng(0)
    constants = [_rnd(rng) for _ in range(MC.MIMC_ROUNDS)]
    g1 = lambda k: O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(k % R, 4)))[0]
    g2 = lambda k: O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(k % R, 4)))[0]
    # parameters (generator.rs): the circuit's matrices do not depend on the witness
    shape = MC.circuit(1, 2, constants)
    assert shape["n_cons"] == 2 * MC.MIMC_ROUNDS and shape["n_wit"] == 2 * MC.MIMC_ROUNDS + 1
    waste = [_rnd(rng) for _ in range(6)]
    pk, n_inst = LG.generate_parameters(shape["A"], shape["B"], shape["C"], shape["n_inst"], shape["n_wit"], 2, *waste, g1(_rnd(rng)), g2(_rnd(rng)))
    vk, pvk = pk.vk, LG.prepare_verifying_key(pk.vk)
    # CP_link: 3 Pedersen bases (xl, xr, link_v) + the link generators (mimc.rs:168-178)
    gens = np.stack([g1(_rnd(rng)) for _ in range(3)])
    pp, ek, lvk, bases = LG.generate_link_keys(vk, n_inst, gens, g1(_rnd(rng)), g2(_rnd(rng)), [_rnd(rng), _rnd(rng)], _rnd(rng))
    mats = [qap.csr(shape[k]) for k in "ABC"]
    circ = qap.DeviceR1cs(*mats, len(shape["z"]), shape["n_inst"], shape["n_cons"])
    for sample in range(3):                       # (the reference loops 50 times for its timing printout)
        xl, xr = _rnd(rng), _rnd(rng)
        cs = MC.circuit(xl, xr, constants)
        image = MC.mimc(xl, xr, constants)
        assert cs["z"][1] == image
        pub, pub_bad = LS.scalars([image]), LS.scalars([(image + 1) % R])
        z = LS.scalars(cs["z"])
        inp, wit = z[:n_inst], z[n_inst:]
        r, s, v, link_v = _rnd(rng), _rnd(rng), _rnd(rng), _rnd(rng)
        # proof without CP_link: witness map + MSMs on the device (create_random_proof -> create_proof_with_reduction, prover.rs:153-180)
        proof = LG.create_proof_with_reduction(pk, circ, r, s, v, z)
        # the same proof from a host-side h (the oracle's witness map): every element equal
        h = LS.scalars(LS.witness_map(cs))
        proof_h = LG.create_proof(pk, r, s, v, h, inp, wit)
        assert all((proof[k] == proof_h[k]).all() for k in proof)
        # proof with CP_link
        pl = LG.create_proof_incl_cp_link(pk, pp, ek, bases, r, s, v, link_v, h, inp, wit)
        # the prover checks its own commitments (mimc.rs:235-238)
        LG.verify_commitments(vk, bases, pl, 1, [xl, xr], v, link_v)
        LG.verify_witness_commitment(vk, proof, 1, [xl, xr], v)
        # verification (mimc.rs:240-248)
        assert LG.verify_proof(pvk, pl["groth16_proof"], pub); LG.verify_link_proof(pp, lvk, pl)
        assert LG.verify_proof(pvk, proof, pub)
        assert not LG.verify_proof(pvk, proof, pub_bad)
        # re-randomisation (mimc.rs:250-270)
        p2 = LG.rerandomize_proof(proof, vk, _rnd(rng), _rnd(rng))
        assert LG.verify_proof(pvk, p2, pub)
        new_v = _rnd(rng)
        p3 = LG.rerandomize_proof_1(proof, v, new_v, vk, pk.eta_delta_inv_g1, _rnd(rng), _rnd(rng))
        assert LG.verify_proof(pvk, p3, pub)
        LG.verify_witness_commitment(vk, p3, 1, [xl, xr], new_v)
        with pytest.raises(ValueError):
            LG.verify_witness_commitment(vk, p3, 1, [xl, xr], v)
        with pytest.raises(ValueError):
            LG.verify_witness_commitment(vk, proof, 1, [xl, (xr + 1) % R], v)
        with pytest.raises(LK.LinkError):
            bad = dict(pl); bad["link_d"] = gens[0]; LG.verify_link_proof(pp, lvk, bad)
    circ.free()
