"""KZG openings of the final commitment keys — /root/reference/legogroth16/src/aggregation/kzg.rs
(verify_kzg_v :30-76, verify_kzg_w :78-125, kzg_check_v/w :127-180, create_kzg_opening :182-236,
polynomial_evaluation_product_form_from_transcript :238-256, polynomial_coefficients_from_transcript :258-292,
prove_commitment_v/w :294-343)."""
import numpy as np
from . import ops
from .ops import G1, G2, R_MOD, inv


def polynomial_evaluation_product_form_from_transcript(transcript, z, r_shift):
    power_zr = z * r_shift % R_MOD
    res = (1 + transcript[0] * power_zr) % R_MOD
    for x in transcript[1:]:
        power_zr = power_zr * power_zr % R_MOD
        res = res * (1 + x * power_zr) % R_MOD
    return res


def polynomial_coefficients_from_transcript(transcript, r_shift):
    coefficients = [1]
    power_2_r = r_shift % R_MOD
    for i, x in enumerate(transcript):
        if i > 0:
            power_2_r = power_2_r * power_2_r % R_MOD
        k = x * power_2_r % R_MOD
        coefficients += [c * k % R_MOD for c in coefficients]
    return coefficients


def _quotient_by_linear(coeffs, z):
    """quotient of poly(X) / (X - z) (the remainder — poly(z) — is dropped, as DensePolynomial `/` does)"""
    n = len(coeffs)
    q = [0] * (n - 1)
    carry = 0
    for i in range(n - 1, 0, -1):
        carry = (coeffs[i] + carry * z) % R_MOD
        q[i - 1] = carry
    return q


def create_kzg_opening(curve, srs_powers_alpha_table, srs_powers_beta_table, poly, eval_poly, kzg_challenge):
    if len(poly) != len(srs_powers_alpha_table):
        raise ValueError("SRS len %d != coefficients len %d" % (len(srs_powers_alpha_table), len(poly)))
    p = list(poly)
    p[0] = (p[0] - eval_poly) % R_MOD
    q = _quotient_by_linear(p, kzg_challenge % R_MOD)
    q += [0] * (len(srs_powers_alpha_table) - len(q))
    return ops.msm(curve, srs_powers_alpha_table, q), ops.msm(curve, srs_powers_beta_table, q)


def prove_commitment_v(h_alpha_table, h_beta_table, transcript, kzg_challenge):
    poly = polynomial_coefficients_from_transcript(transcript, 1)
    ev = polynomial_evaluation_product_form_from_transcript(transcript, kzg_challenge, 1)
    return create_kzg_opening(G2, h_alpha_table, h_beta_table, poly, ev, kzg_challenge)


def prove_commitment_w(g_alpha_table, g_beta_table, transcript, r_shift, kzg_challenge):
    n = len(g_alpha_table)
    f = polynomial_coefficients_from_transcript(transcript, r_shift)
    fw = [0] * len(f) + f
    fz = polynomial_evaluation_product_form_from_transcript(transcript, kzg_challenge, r_shift)
    fwz = fz * pow(kzg_challenge, n, R_MOD) % R_MOD          # (kzg.rs:329: n = table length; only the dropped remainder depends on it)
    return create_kzg_opening(G1, g_alpha_table, g_beta_table, fw, fwz, kzg_challenge)


def verify_kzg_v(v_srs, final_vkey, vkey_opening, challenges, kzg_challenge, checker):
    y = polynomial_evaluation_product_form_from_transcript(challenges, kzg_challenge, 1)
    ng = ops.neg(G1, v_srs.g)
    items = ((final_vkey[0], v_srs.g_alpha, vkey_opening[0]), (final_vkey[1], v_srs.g_beta, vkey_opening[1]))
    # the four two-term combinations are independent: issued together (the library keeps 4 calls in flight)
    bs_cs = ops.parallel([f for cf, vk, _ in items for f in (lambda cf=cf: ops.msm(G2, np.stack([cf, v_srs.h]), [1, -y]),                 # C_f - y h
                                                             lambda vk=vk: ops.msm(G1, np.stack([vk, v_srs.g]), [1, -kzg_challenge]))])      # vk - x g
    for k, (_, _, pi) in enumerate(items):
        b, c = bs_cs[2 * k], bs_cs[2 * k + 1]
        checker.add_multiple_sources_and_target(np.stack([ng, c]), np.stack([b, pi]), ops.fp12_one())


def verify_kzg_w(v_srs, final_wkey, wkey_opening, challenges, r_shift, kzg_challenge, checker):
    fz = polynomial_evaluation_product_form_from_transcript(challenges, kzg_challenge, r_shift)
    fwz = fz * pow(kzg_challenge, v_srs.n, R_MOD) % R_MOD
    nh = ops.neg(G2, v_srs.h)
    items = ((final_wkey[0], v_srs.h_alpha, wkey_opening[0]), (final_wkey[1], v_srs.h_beta, wkey_opening[1]))
    as_ds = ops.parallel([f for cf, wk, _ in items for f in (lambda cf=cf: ops.msm(G1, np.stack([cf, v_srs.g]), [1, -fwz]),               # C_f - y g
                                                             lambda wk=wk: ops.msm(G2, np.stack([wk, v_srs.h]), [1, -kzg_challenge]))])     # wk - x h
    for k, (_, _, pi) in enumerate(items):
        a, d = as_ds[2 * k], as_ds[2 * k + 1]
        checker.add_multiple_sources_and_target(np.stack([a, pi]), np.stack([nh, d]), ops.fp12_one())
