"""Aggregation of Groth16 proofs with TIPP / MIPP — /root/reference/legogroth16/src/aggregation/groth16/prover.rs
(aggregate_proofs :47-147, prove_tipp_mipp :156-206, gipa_tipp_mipp :212-382), groth16/verifier.rs
(verify_aggregate_proof :36-100, verify_tipp_mipp :102-192, gipa_verify_tipp_mipp :194-400), groth16/proof.rs and
utils.rs (compress :34-49, inner_product_and_* :51-118, aggregate_public_inputs :120-158, final_verification_check :218-265).

Proofs are dicts {"a": G1, "b": G2, "c": G1} of affine ABI points; the prepared verifying key is the dict of
crypto_amd.legogroth16.prepare_verifying_key (only `vk` is used, as in the reference)."""
import numpy as np
from . import ops, kzg
from .ops import G1, G2, R_MOD, inv
from .srs import PairCommitment, Key, AggregationError, MAX_SRS_SIZE
from ..pairing_check import RandomizedPairingChecker


def powers(r, n):
    out, acc = [], 1
    for _ in range(n):
        out.append(acc); acc = acc * r % R_MOD
    return out


def compress(curve, vec, split, scalar):
    """utils.rs:34-49: vec[i] + scalar * vec[i + split], halving the vector"""
    return ops.mul_add(curve, vec[split:], int(scalar), vec[:split])


def _gipa(transcript, a, b, c, vkey, wkey, r_vec, ip_ab, agg_c):
    """gipa_tipp_mipp (prover.rs:212-382)"""
    m_a, m_b, m_c, m_r = ops.pts(G1, a), ops.pts(G2, b), ops.pts(G1, c), list(r_vec)
    comms_ab, comms_c, z_ab, z_c, challenges, challenges_inv = [], [], [], [], [], []
    transcript.append(b"inner-product-ab", ops.gt_bytes(ip_ab))
    transcript.append(b"comm-c", ops.g1_bytes(agg_c))
    c_inv = transcript.challenge_scalar(b"first-challenge")
    ch = inv(c_inv)
    i = 0
    while len(m_a) > 1:
        split = len(m_a) // 2
        a_left, a_right = m_a[:split], m_a[split:]
        b_left, b_right = m_b[:split], m_b[split:]
        c_left, c_right = m_c[:split], m_c[split:]
        r_left, r_right = m_r[:split], m_r[split:]
        vk_left, vk_right = vkey.split(split)
        wk_left, wk_right = wkey.split(split)
        # TIPP (utils.rs:83-118) and MIPP for C (utils.rs:51-81): ten multi-pairings and two MSMs, all independent
        tab_l, tab_r, zab_l, zab_r, zc_l, zc_r, tuc_l, tuc_r = ops.parallel([
            lambda: PairCommitment.double(vk_left, wk_right, a_right, b_left),
            lambda: PairCommitment.double(vk_right, wk_left, a_left, b_right),
            lambda: ops.multi_pairing(a_right, b_left),
            lambda: ops.multi_pairing(a_left, b_right),
            lambda: ops.msm(G1, c_right, r_left),
            lambda: ops.msm(G1, c_left, r_right),
            lambda: PairCommitment.single(vk_left, c_right),
            lambda: PairCommitment.single(vk_right, c_left)])
        if i > 0:
            transcript.append(b"c_inv", ops.fr_bytes(c_inv))
            transcript.append(b"zab_l", ops.gt_bytes(zab_l)); transcript.append(b"zab_r", ops.gt_bytes(zab_r))
            transcript.append(b"zc_l", ops.g1_bytes(zc_l)); transcript.append(b"zc_r", ops.g1_bytes(zc_r))
            transcript.append(b"tab_l", tab_l.to_bytes()); transcript.append(b"tab_r", tab_r.to_bytes())
            transcript.append(b"tuc_l", tuc_l.to_bytes()); transcript.append(b"tuc_r", tuc_r.to_bytes())
            c_inv = transcript.challenge_scalar(b"challenge_i")
            ch = inv(c_inv)
        # folding (prover.rs:328-351): A, C and both w vectors take the challenge, B and both v vectors its inverse —
        # one launch per group instead of `compress` x3 + Key::compress x2
        def fold_g1():
            r = ops.mul_add(G1, np.concatenate([a_right, c_right, wk_right.a, wk_right.b]), ch, np.concatenate([a_left, c_left, wk_left.a, wk_left.b]))
            return r[:split], r[split:2 * split], Key(G1, r[2 * split:3 * split], r[3 * split:])

        def fold_g2():
            r = ops.mul_add(G2, np.concatenate([b_right, vk_right.a, vk_right.b]), c_inv, np.concatenate([b_left, vk_left.a, vk_left.b]))
            return r[:split], Key(G2, r[split:2 * split], r[2 * split:])
        (m_a, m_c, wkey), (m_b, vkey) = ops.parallel([fold_g1, fold_g2])
        m_r = [(l + rr * c_inv) % R_MOD for l, rr in zip(r_left, r_right)]
        comms_ab.append((tab_l, tab_r)); comms_c.append((tuc_l, tuc_r))
        z_ab.append((zab_l, zab_r)); z_c.append((zc_l, zc_r))
        challenges.append(ch); challenges_inv.append(c_inv)
        i += 1
    assert len(m_a) == 1 and len(m_b) == 1 and len(m_c) == 1 and len(m_r) == 1 and len(vkey) == 1 and len(wkey) == 1
    gipa = {"nproofs": len(a), "comms_ab": comms_ab, "comms_c": comms_c, "z_ab": z_ab, "z_c": z_c,
            "final_a": m_a[0].copy(), "final_b": m_b[0].copy(), "final_c": m_c[0].copy(),
            "final_vkey": vkey.first(), "final_wkey": wkey.first()}
    return gipa, challenges, challenges_inv


def _kzg_challenge(transcript, first_challenge, gipa):
    transcript.append(b"kzg-challenge", ops.fr_bytes(first_challenge))
    transcript.append(b"vkey0", ops.g2_bytes(gipa["final_vkey"][0])); transcript.append(b"vkey1", ops.g2_bytes(gipa["final_vkey"][1]))
    transcript.append(b"wkey0", ops.g1_bytes(gipa["final_wkey"][0])); transcript.append(b"wkey1", ops.g1_bytes(gipa["final_wkey"][1]))
    return transcript.challenge_scalar(b"z-challenge")


def _prove_tipp_mipp(srs, transcript, a, b, c, wkey, r_vec, z_ab, z_c):
    """prover.rs:156-206"""
    r_shift = r_vec[1]
    gipa, challenges, challenges_inv = _gipa(transcript, a, b, c, srs.vkey, wkey, r_vec, z_ab, z_c)
    challenges.reverse(); challenges_inv.reverse()
    r_inverse = inv(r_shift)
    z = _kzg_challenge(transcript, challenges[0], gipa)
    vkey_opening = kzg.prove_commitment_v(srs.h_alpha_powers_table, srs.h_beta_powers_table, challenges_inv, z)
    wkey_opening = kzg.prove_commitment_w(srs.g_alpha_powers_table, srs.g_beta_powers_table, challenges, r_inverse, z)
    return {"gipa": gipa, "vkey_opening": vkey_opening, "wkey_opening": wkey_opening}


def aggregate_proofs(srs, transcript, proofs):
    """prover.rs:47-147"""
    n = len(proofs)
    if n < 2:
        raise AggregationError("invalid proof size < 2")
    if n & (n - 1):
        raise AggregationError("invalid proof size: not power of two")
    if not srs.has_correct_len(n):
        raise AggregationError("SRS len %d != proofs len %d" % (len(srs.vkey), n))
    a = np.stack([p["a"] for p in proofs]); b = np.stack([p["b"] for p in proofs]); c = np.stack([p["c"] for p in proofs])
    com_ab = PairCommitment.double(srs.vkey, srs.wkey, a, b)
    com_c = PairCommitment.single(srs.vkey, c)
    transcript.append(b"AB-commitment", com_ab.to_bytes())
    transcript.append(b"C-commitment", com_c.to_bytes())
    r = transcript.challenge_scalar(b"r-random-fiatshamir")
    r_vec = powers(r, n)
    r_inv = [inv(x) for x in r_vec]
    b_r = ops.mul_add(G2, b, r_vec)                           # B^{r^i}   (:107-112)
    z_ab = ops.multi_pairing(a, b_r)                          # :115
    z_c = ops.msm(G1, c, r_vec)                               # :117
    wkey_r_inv = srs.wkey.scale(r_inv)                        # :120
    tmipp = _prove_tipp_mipp(srs, transcript, a, b_r, c, wkey_r_inv, r_vec, z_ab, z_c)
    return {"com_ab": com_ab, "com_c": com_c, "z_ab": z_ab, "z_c": z_c, "tmipp": tmipp}


# ---- verifier -------------------------------------------------------------------------------------------------------------
def parsing_check(proof):
    """proof.rs:29-58"""
    gipa = proof["tmipp"]["gipa"]
    n = gipa["nproofs"]
    if n < 2 or n > MAX_SRS_SIZE:
        raise AggregationError("Proof length out of bounds")
    if n & (n - 1):
        raise AggregationError("Proof length not a power of two")
    ref_len = n.bit_length() - 1
    if not (ref_len == len(gipa["comms_ab"]) == len(gipa["comms_c"]) == len(gipa["z_ab"]) == len(gipa["z_c"])):
        raise AggregationError("Proof vectors unequal sizes")


def _gipa_verify(proof, r_shift, transcript):
    """gipa_verify_tipp_mipp (verifier.rs:194-400): replay the challenges, fold T, U, Z with them"""
    gipa = proof["tmipp"]["gipa"]
    challenges, challenges_inv = [], []
    transcript.append(b"inner-product-ab", ops.gt_bytes(proof["z_ab"]))
    transcript.append(b"comm-c", ops.g1_bytes(proof["z_c"]))
    c_inv = transcript.challenge_scalar(b"first-challenge")
    ch = inv(c_inv)
    for i, ((tab_l, tab_r), (zab_l, zab_r), (tuc_l, tuc_r), (zc_l, zc_r)) in enumerate(zip(gipa["comms_ab"], gipa["z_ab"], gipa["comms_c"], gipa["z_c"])):
        if i > 0:
            transcript.append(b"c_inv", ops.fr_bytes(c_inv))
            transcript.append(b"zab_l", ops.gt_bytes(zab_l)); transcript.append(b"zab_r", ops.gt_bytes(zab_r))
            transcript.append(b"zc_l", ops.g1_bytes(zc_l)); transcript.append(b"zc_r", ops.g1_bytes(zc_r))
            transcript.append(b"tab_l", tab_l.to_bytes()); transcript.append(b"tab_r", tab_r.to_bytes())
            transcript.append(b"tuc_l", tuc_l.to_bytes()); transcript.append(b"tuc_r", tuc_r.to_bytes())
            c_inv = transcript.challenge_scalar(b"challenge_i")
            ch = inv(c_inv)
        challenges.append(ch); challenges_inv.append(c_inv)
    res = {"tab": proof["com_ab"].t, "uab": proof["com_ab"].u, "zab": proof["z_ab"], "tc": proof["com_c"].t, "uc": proof["com_c"].u}
    # zc = z_c + sum (c zc_l + c^-1 zc_r)      (:262-270)
    zpts = [proof["z_c"]] + [p for pair in gipa["z_c"] for p in pair]
    zsc = [1] + [s for pair in zip(challenges, challenges_inv) for s in pair]
    res["zc"] = ops.msm(G1, np.stack(zpts), zsc)
    # T, U, Z folded with the challenges (:272-370): left entries to the challenge, right entries to its inverse
    exps = [1] + [s for pair in zip(challenges, challenges_inv) for s in pair]
    sel = {"tab": lambda ab, z, c: (ab[0].t, ab[1].t), "uab": lambda ab, z, c: (ab[0].u, ab[1].u), "zab": lambda ab, z, c: z,
           "tc": lambda ab, z, c: (c[0].t, c[1].t), "uc": lambda ab, z, c: (c[0].u, c[1].u)}
    for key, pick in sel.items():
        bases = [res[key]]
        for ab, z, cc in zip(gipa["comms_ab"], gipa["z_ab"], gipa["comms_c"]):
            bases += list(pick(ab, z, cc))
        res[key] = ops.gt_multi_pow(bases, exps)
    challenges.reverse(); challenges_inv.reverse()
    final_r = kzg.polynomial_evaluation_product_form_from_transcript(challenges_inv, r_shift, 1)
    return res, final_r, challenges, challenges_inv


def verify_tipp_mipp(v_srs, proof, r_shift, transcript, checker):
    """verifier.rs:102-192"""
    final_res, final_r, challenges, challenges_inv = _gipa_verify(proof, r_shift, transcript)
    gipa = proof["tmipp"]["gipa"]
    z = _kzg_challenge(transcript, challenges[0], gipa)
    kzg.verify_kzg_v(v_srs, gipa["final_vkey"], proof["tmipp"]["vkey_opening"], challenges_inv, z, checker)
    kzg.verify_kzg_w(v_srs, gipa["final_wkey"], proof["tmipp"]["wkey_opening"], challenges, inv(r_shift), z, checker)
    fa, fb, fc = gipa["final_a"], gipa["final_b"], gipa["final_c"]
    v0, v1 = gipa["final_vkey"]; w0, w1 = gipa["final_wkey"]
    checker.add_multiple_sources_and_target(fa.reshape(1, 12), fb.reshape(1, 24), final_res["zab"])
    checker.add_multiple_sources_and_target(np.stack([fa, w0]), np.stack([v0, fb]), final_res["tab"])
    checker.add_multiple_sources_and_target(np.stack([fa, w1]), np.stack([v1, fb]), final_res["uab"])
    final_zc = ops.msm(G1, fc.reshape(1, 12), [final_r])
    checker.add_multiple_sources_and_target(fc.reshape(1, 12), v0.reshape(1, 24), final_res["tc"])
    checker.add_multiple_sources_and_target(fc.reshape(1, 12), v1.reshape(1, 24), final_res["uc"])
    if not (final_zc == final_res["zc"]).all():
        raise AggregationError("tipp verify: INVALID final_z check for C")


def aggregate_public_inputs(public_inputs, r_powers, r_sum, gamma_abc_g1):
    """utils.rs:120-158"""
    l = len(public_inputs[0])
    summed = [sum(public_inputs[j][i] * r_powers[j] for j in range(len(public_inputs))) % R_MOD for i in range(l)]
    return ops.msm(G1, gamma_abc_g1[:l + 1], [r_sum] + summed)


def final_verification_check(source1, source2, z_c, z_ab, r, public_inputs, vk, checker):
    """utils.rs:218-265"""
    n = len(public_inputs)
    r_powers = powers(r, n)
    r_sum = sum(r_powers) % R_MOD
    source1 = list(source1) + [ops.msm(G1, vk.alpha_g1.reshape(1, 12), [r_sum]), aggregate_public_inputs(public_inputs, r_powers, r_sum, vk.gamma_abc_g1), z_c]
    source2 = list(source2) + [vk.beta_g2, vk.gamma_g2, vk.delta_g2]
    checker.add_multiple_sources_and_target(np.stack(source1), np.stack(source2), z_ab)
    if not checker.verify():
        raise AggregationError("Proof Verification Failed due to pairing checks")


def verify_aggregate_proof(ip_verifier_srs, pvk, public_inputs, proof, random, transcript, pairing_check=None):
    """verifier.rs:36-100.  public_inputs: one list of ints per proof; `random`: the checker's batching scalar
    (RandomizedPairingChecker::new_using_rng draws it from `rng`).  Raises AggregationError on an invalid proof."""
    vk = pvk["vk"]
    parsing_check(proof)
    for pub in public_inputs:
        if len(pub) + 1 != len(vk.gamma_abc_g1):
            raise AggregationError("MalformedVerifyingKey")
    if len(public_inputs) != proof["tmipp"]["gipa"]["nproofs"]:
        raise AggregationError("public inputs len %d != number of proofs %d" % (len(public_inputs), proof["tmipp"]["gipa"]["nproofs"]))
    transcript.append(b"AB-commitment", proof["com_ab"].to_bytes())
    transcript.append(b"C-commitment", proof["com_c"].to_bytes())
    r = transcript.challenge_scalar(b"r-random-fiatshamir")
    checker = pairing_check if pairing_check is not None else RandomizedPairingChecker(random, True)
    verify_tipp_mipp(ip_verifier_srs, proof, r, transcript, checker)
    final_verification_check([], [], proof["z_c"], proof["z_ab"], r, public_inputs, vk, checker)
