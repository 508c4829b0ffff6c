"""Aggregation of LegoGroth16 proofs (a, b, c, d) — /root/reference/legogroth16/src/aggregation/legogroth16/prover.rs:38-127,
verifier.rs:34-96: the Groth16 protocol with one more MIPP instance for the commitment `d`, whose aggregate joins the gamma
pairing of the final check.  Shares the implementation of crypto_amd/aggregation/groth16.py (with_d=True)."""
from . import groth16


def aggregate_proofs(srs, transcript, proofs):
    return groth16.aggregate_proofs(srs, transcript, proofs, with_d=True)


def verify_aggregate_proof(ip_verifier_srs, pvk, public_inputs, proof, random, transcript, pairing_check=None, validate_gt=True):
    return groth16.verify_aggregate_proof(ip_verifier_srs, pvk, public_inputs, proof, random, transcript, pairing_check, with_d=True, validate_gt=validate_gt)
