"""LegoGroth16 proofs aggregated with the Groth16 aggregator, the commitments `d` shipped alongside —
/root/reference/legogroth16/src/aggregation/legogroth16/using_groth16.rs:26-128.  The verifier folds the d_i itself
(sum_i r^i d_i, one MSM of n terms) into the gamma pairing instead of checking a MIPP proof for them."""
import numpy as np
from . import ops, groth16
from .ops import G1, R_MOD
from .srs import AggregationError
from ..pairing_check import RandomizedPairingChecker


def aggregate_proofs(srs, transcript, proofs):
    """using_groth16.rs:26-43: (Groth16 aggregate of the (a, b, c), the list of d)"""
    return groth16.aggregate_proofs(srs, transcript, proofs), np.stack([p["d"] for p in proofs])


def verify_aggregate_proof(ip_verifier_srs, pvk, public_inputs, proof, d, random, transcript, pairing_check=None):
    """using_groth16.rs:45-128"""
    vk = pvk["vk"]
    groth16.parsing_check(proof)
    for pub in public_inputs:
        if len(pub) + 1 > len(vk.gamma_abc_g1):
            raise AggregationError("MalformedVerifyingKey")
    n = proof["tmipp"]["gipa"]["nproofs"]
    if len(public_inputs) != n:
        raise AggregationError("public inputs len %d != number of proofs %d" % (len(public_inputs), n))
    transcript.append(b"AB-commitment", proof["com_ab"].to_bytes())
    transcript.append(b"C-commitment", proof["com_c"].to_bytes())
    r = transcript.challenge_scalar(b"r-random-fiatshamir")
    checker = pairing_check if pairing_check is not None else RandomizedPairingChecker(random, True)
    groth16.verify_tipp_mipp(ip_verifier_srs, proof, r, transcript, checker)
    r_powers = groth16.powers(r, n)
    r_sum = sum(r_powers) % R_MOD
    # (d_r + inp): both are MSMs over G1 — one call over the concatenated terms
    l = len(public_inputs[0])
    summed = [sum(public_inputs[j][i] * r_powers[j] for j in range(n)) % R_MOD for i in range(l)]
    d = ops.pts(G1, d)
    if len(d) != n:
        raise AggregationError("d len %d != number of proofs %d" % (len(d), n))
    mid = ops.msm(G1, np.concatenate([d, vk.gamma_abc_g1[:l + 1]]), r_powers + [r_sum] + summed)
    source1 = [ops.msm(G1, vk.alpha_g1.reshape(1, 12), [r_sum]), mid, proof["z_c"]]
    source2 = [vk.beta_g2, vk.gamma_g2, vk.delta_g2]
    checker.add_multiple_sources_and_target(np.stack(source1), np.stack(source2), proof["z_ab"])
    if not checker.verify():
        raise AggregationError("Proof Verification Failed due to pairing checks")
