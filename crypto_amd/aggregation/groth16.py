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


def _gipa(transcript, a, b, mipp, vkey, wkey, r_vec, ip_ab, agg):
    """gipa_tipp_mipp (groth16/prover.rs:212-382; legogroth16/prover.rs:176-350 runs one more MIPP, for D).
    mipp: ordered {"c": C} or {"c": C, "d": D}; agg: the matching sum_i r^i C_i (and D)."""
    names = list(mipp)
    m_a, m_b, m_r = ops.pts(G1, a), ops.pts(G2, b), list(r_vec)
    m_v = {k: ops.pts(G1, mipp[k]) for k in names}
    comms_ab, z_ab, challenges, challenges_inv = [], [], [], []
    comms = {k: [] for k in names}; zs = {k: [] for k in names}
    transcript.append(b"inner-product-ab", ops.gt_bytes(ip_ab))
    for k in names:
        transcript.append(b"comm-" + k.encode(), ops.g1_bytes(agg[k]))
    c_inv = transcript.challenge_scalar(b"first-challenge")
    ch = inv(c_inv)
    i = 0
    while len(m_a) > 1:
        split = len(m_a) // 2
        a_left, a_right = m_a[:split], m_a[split:]
        b_left, b_right = m_b[:split], m_b[split:]
        v_left = {k: m_v[k][:split] for k in names}; v_right = {k: m_v[k][split:] for k in names}
        r_left, r_right = m_r[:split], m_r[split:]
        vk_left, vk_right = vkey.split(split)
        wk_left, wk_right = wkey.split(split)
        # TIPP (utils.rs:83-118) and MIPP for C (utils.rs:51-81): ten multi-pairings and two MSMs, all independent — the pairings in
        # one segmented call (dgpu_multi_miller_loop_segments), the MSMs beside it
        jobs = PairCommitment.double_jobs(vk_left, wk_right, a_right, b_left) + PairCommitment.double_jobs(vk_right, wk_left, a_left, b_right) \
            + [(a_right, b_left), (a_left, b_right)]
        thunks = [lambda: ops.multi_pairings(jobs)]
        for k in names:
            jobs += PairCommitment.single_jobs(vk_left, v_right[k]) + PairCommitment.single_jobs(vk_right, v_left[k])
            thunks += [lambda k=k: ops.msm(G1, v_right[k], r_left), lambda k=k: ops.msm(G1, v_left[k], r_right)]
        res = ops.parallel(thunks)
        gts = res[0]
        tab_l, tab_r, zab_l, zab_r = PairCommitment(gts[0], gts[1]), PairCommitment(gts[2], gts[3]), gts[4], gts[5]
        z_lr = {k: (res[1 + 2 * j], res[2 + 2 * j]) for j, k in enumerate(names)}
        tu_lr = {k: (PairCommitment(gts[6 + 4 * j], gts[7 + 4 * j]), PairCommitment(gts[8 + 4 * j], gts[9 + 4 * j])) for j, k in enumerate(names)}
        if i > 0:
            transcript.append(b"c_inv", ops.fr_bytes(c_inv))
            transcript.append(b"zab_l", ops.gt_bytes(zab_l)); transcript.append(b"zab_r", ops.gt_bytes(zab_r))
            for k in names:
                transcript.append(b"z%s_l" % k.encode(), ops.g1_bytes(z_lr[k][0])); transcript.append(b"z%s_r" % k.encode(), ops.g1_bytes(z_lr[k][1]))
            transcript.append(b"tab_l", tab_l.to_bytes()); transcript.append(b"tab_r", tab_r.to_bytes())
            for k in names:
                transcript.append(b"tu%s_l" % k.encode(), tu_lr[k][0].to_bytes()); transcript.append(b"tu%s_r" % k.encode(), tu_lr[k][1].to_bytes())
            c_inv = transcript.challenge_scalar(b"challenge_i")
            ch = inv(c_inv)
        # folding (prover.rs:328-351): A, C and both w vectors take the challenge, B and both v vectors its inverse —
        # one launch per group instead of `compress` x3 + Key::compress x2
        def fold_g1():
            r = ops.mul_add(G1, np.concatenate([a_right, wk_right.a, wk_right.b] + [v_right[k] for k in names]), ch,
                            np.concatenate([a_left, wk_left.a, wk_left.b] + [v_left[k] for k in names]))
            return r[:split], Key(G1, r[split:2 * split], r[2 * split:3 * split]), {k: r[(3 + j) * split:(4 + j) * split] for j, k in enumerate(names)}

        def fold_g2():
            r = ops.mul_add(G2, np.concatenate([b_right, vk_right.a, vk_right.b]), c_inv, np.concatenate([b_left, vk_left.a, vk_left.b]))
            return r[:split], Key(G2, r[split:2 * split], r[2 * split:])
        (m_a, wkey, m_v), (m_b, vkey) = ops.parallel([fold_g1, fold_g2])
        m_r = [(l + rr * c_inv) % R_MOD for l, rr in zip(r_left, r_right)]
        comms_ab.append((tab_l, tab_r)); z_ab.append((zab_l, zab_r))
        for k in names:
            comms[k].append(tu_lr[k]); zs[k].append(z_lr[k])
        challenges.append(ch); challenges_inv.append(c_inv)
        i += 1
    assert len(m_a) == 1 and len(m_b) == 1 and len(m_r) == 1 and len(vkey) == 1 and len(wkey) == 1 and all(len(m_v[k]) == 1 for k in names)
    gipa = {"nproofs": len(a), "comms_ab": comms_ab, "z_ab": z_ab, "final_a": m_a[0].copy(), "final_b": m_b[0].copy(),
            "final_vkey": vkey.first(), "final_wkey": wkey.first()}
    for k in names:
        gipa["comms_" + k], gipa["z_" + k], gipa["final_" + k] = comms[k], zs[k], m_v[k][0].copy()
    return gipa, challenges, challenges_inv


def _kzg_challenge(transcript, first_challenge, gipa):
    transcript.append(b"kzg-challenge", ops.fr_bytes(first_challenge))
    transcript.append(b"vkey0", ops.g2_bytes(gipa["final_vkey"][0])); transcript.append(b"vkey1", ops.g2_bytes(gipa["final_vkey"][1]))
    transcript.append(b"wkey0", ops.g1_bytes(gipa["final_wkey"][0])); transcript.append(b"wkey1", ops.g1_bytes(gipa["final_wkey"][1]))
    return transcript.challenge_scalar(b"z-challenge")


def _prove_tipp_mipp(srs, transcript, a, b, mipp, wkey, r_vec, z_ab, agg):
    """prover.rs:156-206"""
    r_shift = r_vec[1]
    gipa, challenges, challenges_inv = _gipa(transcript, a, b, mipp, srs.vkey, wkey, r_vec, z_ab, agg)
    challenges.reverse(); challenges_inv.reverse()
    r_inverse = inv(r_shift)
    z = _kzg_challenge(transcript, challenges[0], gipa)
    vkey_opening = kzg.prove_commitment_v(srs.h_alpha_powers_table, srs.h_beta_powers_table, challenges_inv, z)
    wkey_opening = kzg.prove_commitment_w(srs.g_alpha_powers_table, srs.g_beta_powers_table, challenges, r_inverse, z)
    return {"gipa": gipa, "vkey_opening": vkey_opening, "wkey_opening": wkey_opening}


def aggregate_proofs(srs, transcript, proofs, with_d=False):
    """groth16/prover.rs:47-147; with_d: legogroth16/prover.rs:38-127 (the proofs carry the commitment `d`, which gets its
    own single commitment, aggregate z_d and MIPP)"""
    names = ("c", "d") if with_d else ("c",)
    n = len(proofs)
    if n < 2:
        raise AggregationError("invalid proof size < 2")
    if n & (n - 1):
        raise AggregationError("invalid proof size: not power of two")
    if not srs.has_correct_len(n):
        raise AggregationError("SRS len %d != proofs len %d" % (len(srs.vkey), n))
    a = np.stack([p["a"] for p in proofs]); b = np.stack([p["b"] for p in proofs])
    mipp = {k: np.stack([p[k] for p in proofs]) for k in names}
    jobs = PairCommitment.double_jobs(srs.vkey, srs.wkey, a, b)
    for k in names:
        jobs += PairCommitment.single_jobs(srs.vkey, mipp[k])
    gts = ops.multi_pairings(jobs)                            # :77-88, one segmented call
    com_ab = PairCommitment(gts[0], gts[1])
    com = {k: PairCommitment(gts[2 + 2 * j], gts[3 + 2 * j]) for j, k in enumerate(names)}
    transcript.append(b"AB-commitment", com_ab.to_bytes())
    for k in names:
        transcript.append(k.upper().encode() + b"-commitment", com[k].to_bytes())
    r = transcript.challenge_scalar(b"r-random-fiatshamir")
    r_vec = powers(r, n)
    r_inv = powers(inv(r), n)                                 # 1, r^-1, r^-2, ... (the reference batch-inverts r_vec, :101-103)
    def scaled_b():
        b_r = ops.mul_add(G2, b, r_vec)                       # B^{r^i}   (:107-112)
        return b_r, ops.multi_pairing(a, b_r)                 # :115
    res = ops.parallel([scaled_b, lambda: srs.wkey.scale(r_inv)] + [lambda k=k: ops.msm(G1, mipp[k], r_vec) for k in names])   # :107-120, independent
    (b_r, z_ab), wkey_r_inv = res[0], res[1]
    agg = {k: res[2 + j] for j, k in enumerate(names)}
    tmipp = _prove_tipp_mipp(srs, transcript, a, b_r, mipp, wkey_r_inv, r_vec, z_ab, agg)
    out = {"com_ab": com_ab, "z_ab": z_ab, "tmipp": tmipp}
    for k in names:
        out["com_" + k], out["z_" + k] = com[k], agg[k]
    return out


# ---- verifier -------------------------------------------------------------------------------------------------------------
def parsing_check(proof, names=("c",)):
    """proof.rs:29-58; every vector the variant `names` implies must be present (a proof that lacks comms_d / z_d is malformed, not a KeyError)"""
    for key in ["com_ab", "z_ab", "tmipp"] + ["com_" + k for k in names] + ["z_" + k for k in names]:
        if key not in proof:
            raise AggregationError("Proof is missing " + key)
    if "gipa" not in proof["tmipp"]:
        raise AggregationError("Proof is missing tmipp.gipa")
    gipa = proof["tmipp"]["gipa"]
    for key in ["nproofs", "comms_ab", "z_ab"] + [p + k for k in names for p in ("comms_", "z_")]:
        if key not in gipa:
            raise AggregationError("Proof is missing gipa." + key)
    n = gipa["nproofs"]
    if n < 2 or n > MAX_SRS_SIZE:
        raise AggregationError("Proof length out of bounds")
    if n & (n - 1):
        raise AggregationError("Proof length not a power of two")
    ref_len = n.bit_length() - 1
    lens = [len(gipa["comms_ab"]), len(gipa["z_ab"])] + [len(gipa[p + k]) for k in names for p in ("comms_", "z_")]
    if any(x != ref_len for x in lens):
        raise AggregationError("Proof vectors unequal sizes")


def _gipa_verify(proof, r_shift, transcript, names):
    """gipa_verify_tipp_mipp (groth16/verifier.rs:194-400, legogroth16/verifier.rs:189-470): replay the challenges, fold T, U, Z with them"""
    gipa = proof["tmipp"]["gipa"]
    challenges, challenges_inv = [], []
    transcript.append(b"inner-product-ab", ops.gt_bytes(proof["z_ab"]))
    for k in names:
        transcript.append(b"comm-" + k.encode(), ops.g1_bytes(proof["z_" + k]))
    c_inv = transcript.challenge_scalar(b"first-challenge")
    ch = inv(c_inv)
    for i in range(len(gipa["comms_ab"])):
        if i > 0:
            (tab_l, tab_r), (zab_l, zab_r) = gipa["comms_ab"][i], gipa["z_ab"][i]
            transcript.append(b"c_inv", ops.fr_bytes(c_inv))
            transcript.append(b"zab_l", ops.gt_bytes(zab_l)); transcript.append(b"zab_r", ops.gt_bytes(zab_r))
            for k in names:
                zl, zr = gipa["z_" + k][i]
                transcript.append(b"z%s_l" % k.encode(), ops.g1_bytes(zl)); transcript.append(b"z%s_r" % k.encode(), ops.g1_bytes(zr))
            transcript.append(b"tab_l", tab_l.to_bytes()); transcript.append(b"tab_r", tab_r.to_bytes())
            for k in names:
                tl, tr_ = gipa["comms_" + k][i]
                transcript.append(b"tu%s_l" % k.encode(), tl.to_bytes()); transcript.append(b"tu%s_r" % k.encode(), tr_.to_bytes())
            c_inv = transcript.challenge_scalar(b"challenge_i")
            ch = inv(c_inv)
        challenges.append(ch); challenges_inv.append(c_inv)
    exps = [1] + [s for pair in zip(challenges, challenges_inv) for s in pair]
    res = {}
    # z_k = z_k + sum (c z_l + c^-1 z_r)      (:262-270)
    for k in names:
        zpts = [proof["z_" + k]] + [p for pair in gipa["z_" + k] for p in pair]
        res["z" + k] = ops.msm(G1, np.stack(zpts), exps)
    # T, U, Z folded with the challenges (:272-370): left entries to the challenge, right entries to its inverse
    jobs = {"tab": (proof["com_ab"].t, [(l.t, r.t) for l, r in gipa["comms_ab"]]), "uab": (proof["com_ab"].u, [(l.u, r.u) for l, r in gipa["comms_ab"]]),
            "zab": (proof["z_ab"], gipa["z_ab"])}
    for k in names:
        jobs["t" + k] = (proof["com_" + k].t, [(l.t, r.t) for l, r in gipa["comms_" + k]])
        jobs["u" + k] = (proof["com_" + k].u, [(l.u, r.u) for l, r in gipa["comms_" + k]])
    keys = list(jobs)
    outs = ops.parallel([lambda k=k: ops.gt_multi_pow([jobs[k][0]] + [x for pair in jobs[k][1] for x in pair], exps) for k in keys], host=True)
    for k, o in zip(keys, outs):
        res[k] = o
    challenges.reverse(); challenges_inv.reverse()
    final_r = kzg.polynomial_evaluation_product_form_from_transcript(challenges_inv, r_shift, 1)
    return res, final_r, challenges, challenges_inv


def verify_tipp_mipp(v_srs, proof, r_shift, transcript, checker, names=("c",)):
    """groth16/verifier.rs:102-192, legogroth16/verifier.rs:98-187"""
    final_res, final_r, challenges, challenges_inv = _gipa_verify(proof, r_shift, transcript, names)
    gipa = proof["tmipp"]["gipa"]
    z = _kzg_challenge(transcript, challenges[0], gipa)
    kzg.verify_kzg_v(v_srs, gipa["final_vkey"], proof["tmipp"]["vkey_opening"], challenges_inv, z, checker)
    kzg.verify_kzg_w(v_srs, gipa["final_wkey"], proof["tmipp"]["wkey_opening"], challenges, inv(r_shift), z, checker)
    fa, fb = gipa["final_a"], gipa["final_b"]
    v0, v1 = gipa["final_vkey"]; w0, w1 = gipa["final_wkey"]
    checker.add_multiple_sources_and_target(fa.reshape(1, 12), fb.reshape(1, 24), final_res["zab"])
    checker.add_multiple_sources_and_target(np.stack([fa, w0]), np.stack([v0, fb]), final_res["tab"])
    checker.add_multiple_sources_and_target(np.stack([fa, w1]), np.stack([v1, fb]), final_res["uab"])
    bad = None
    for k in names:                                           # MIPP: Z == final^final_r, T = e(final, v1), U = e(final, v2)
        fk = gipa["final_" + k]
        final_z = ops.msm(G1, fk.reshape(1, 12), [final_r])
        checker.add_multiple_sources_and_target(fk.reshape(1, 12), v0.reshape(1, 24), final_res["t" + k])
        checker.add_multiple_sources_and_target(fk.reshape(1, 12), v1.reshape(1, 24), final_res["u" + k])
        if bad is None and not (final_z == final_res["z" + k]).all():
            bad = k
    if bad is not None:
        raise AggregationError("tipp verify: INVALID final_z check for " + bad.upper())


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


def gt_elements(proof):
    """every target-group element an aggregate proof carries (commitment outputs are pairs)"""
    gipa = proof["tmipp"]["gipa"]
    out = [proof["com_ab"].t, proof["com_ab"].u, proof["z_ab"]]
    for key in ("com_c", "com_d"):
        if key in proof:
            out += [proof[key].t, proof[key].u]
    for lr in gipa["comms_ab"]:
        for o in lr:
            out += [o.t, o.u]
    for lr in gipa["z_ab"]:
        out += list(lr)
    for key in ("comms_c", "comms_d"):
        for lr in gipa.get(key, []):
            for o in lr:
                out += [o.t, o.u]
    return out


def verify_aggregate_proof(ip_verifier_srs, pvk, public_inputs, proof, random, transcript, pairing_check=None, with_d=False, validate_gt=True):
    """groth16/verifier.rs:36-100; with_d: legogroth16/verifier.rs:34-96 (z_d joins the gamma pairing of the final check).
    public_inputs: one list of ints per proof; `random`: the checker's batching scalar (RandomizedPairingChecker::new_using_rng
    draws it from `rng`).  Raises AggregationError on an invalid proof.  validate_gt: what `CanonicalDeserialize` with `Validate::Yes` does
    for a proof that arrives as bytes — every GT element must lie in the order-r subgroup (f^r == 1; host arithmetic, ~1 ms per element,
    2 + 6 log2(n) .. elements) — for proofs held in memory by an untrusted producer."""
    vk = pvk["vk"]
    names = ("c", "d") if with_d else ("c",)
    parsing_check(proof, names)
    if validate_gt and not all(ops.parallel([(lambda f=f: ops.gt_in_subgroup(f)) for f in gt_elements(proof)], host=True)):
        raise AggregationError("a target-group element of the proof is outside the order-r subgroup")
    if not public_inputs or any(len(pub) != len(public_inputs[0]) for pub in public_inputs):
        raise AggregationError("public inputs of unequal length")            # (aggregate_public_inputs indexes them as a rectangle)
    for pub in public_inputs:
        if (len(pub) + 1 > len(vk.gamma_abc_g1)) if with_d else (len(pub) + 1 != len(vk.gamma_abc_g1)):
            raise AggregationError("MalformedVerifyingKey")
    if len(public_inputs) != proof["tmipp"]["gipa"]["nproofs"]:
        raise AggregationError("public inputs len %d != number of proofs %d" % (len(public_inputs), proof["tmipp"]["gipa"]["nproofs"]))
    transcript.append(b"AB-commitment", proof["com_ab"].to_bytes())
    for k in names:
        transcript.append(k.upper().encode() + b"-commitment", proof["com_" + k].to_bytes())
    r = transcript.challenge_scalar(b"r-random-fiatshamir")
    checker = pairing_check if pairing_check is not None else RandomizedPairingChecker(random, True)
    verify_tipp_mipp(ip_verifier_srs, proof, r, transcript, checker, names)
    source1, source2 = ([proof["z_d"]], [vk.gamma_g2]) if with_d else ([], [])
    final_verification_check(source1, source2, proof["z_c"], proof["z_ab"], r, public_inputs, vk, checker)
