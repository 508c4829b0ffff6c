"""LegoGroth16 prover / verifier group-side arithmetic over the C ABI — mirror of
/root/reference/legogroth16/src/prover.rs:267-383 (`create_proof_and_committed_witnesses_with_assignment`,
`calculate_coeff` :585-594) and /root/reference/legogroth16/src/verifier.rs:29-109 (`prepare_inputs`, `calculate_d`,
`verify_qap_proof`).  Every large MSM goes through device-resident proving-key handles (`&query[1..]` is the handle
offset), the handful of O(1) scalar multiplications are tiny MSMs, sums are `dgpu_fold_*`, the verifier's three-pair
check is `dgpu_multi_miller_loop` + `dgpu_final_exponentiation`.

The witness map (h coefficients; r1cs_to_qap.rs:150-210) is SURVEY 8f-1 "next": `h` is an input here, exactly as it
is an input of the reference function mirrored.  Scalars are canonical (`into_bigint`) 4x64 numpy rows; points are
ABI-layout numpy arrays (affine; identity = all-zero words).
"""
import numpy as np
import importlib
M = importlib.import_module(__package__ + ".msm")   # (the package re-exports a function called `msm`, which shadows the submodule attribute)
from . import sharded
from . import pairing

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


def _sc(v):
    v %= R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _affine(curve, jac):
    """normalised Jacobian triple (what the ABI returns) -> affine ABI point (identity -> zero words)"""
    jac = np.asarray(jac, dtype=np.uint64)
    if not jac[curve.AW:].any():
        return np.zeros(curve.AW, dtype=np.uint64)
    return jac[:curve.AW].copy()


def _neg_affine(curve, pt):
    """-P for an affine ABI point: negate every Fq limb group of y (p - y on the Montgomery representative)"""
    pt = np.array(pt, dtype=np.uint64)
    if not pt.any():
        return pt
    h = curve.AW // 2
    for k in range(h // 6):
        y = sum(int(x) << (64 * i) for i, x in enumerate(pt[h + 6 * k:h + 6 * k + 6]))
        y = (P_MOD - y) % P_MOD
        pt[h + 6 * k:h + 6 * k + 6] = [(y >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]
    return pt


def lincomb(curve, points, scalars):
    """sum scalars[i] * points[i] (a tiny MSM through the same entry point), normalised Jacobian"""
    pts = np.stack([np.asarray(p, dtype=np.uint64) for p in points])
    sc = np.stack([_sc(s) for s in scalars])
    return M.msm_bigint(curve, pts, sc)


class ProvingKey:
    """ProvingKeyCommon + VerifyingKey (legogroth16/src/data_structures.rs:55-70,151-168); queries live on the device."""

    def __init__(self, vk, beta_g1, delta_g1, eta_delta_inv_g1, a_query, b_g1_query, b_g2_query, h_query, l_query):
        self.vk = vk
        self.beta_g1, self.delta_g1, self.eta_delta_inv_g1 = beta_g1, delta_g1, eta_delta_inv_g1
        self.a0, self.b1_0, self.b2_0 = a_query[0].copy(), b_g1_query[0].copy(), b_g2_query[0].copy()
        self.a_query = M.DeviceBases(M.G1, a_query)
        self.b_g1_query = M.DeviceBases(M.G1, b_g1_query)
        self.b_g2_query = M.DeviceBases(M.G2, b_g2_query)
        self.h_query = M.DeviceBases(M.G1, h_query)
        self.l_query = M.DeviceBases(M.G1, l_query)


class VerifyingKey:
    def __init__(self, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, eta_gamma_inv_g1, commit_witness_count):
        self.alpha_g1, self.beta_g2, self.gamma_g2, self.delta_g2 = alpha_g1, beta_g2, gamma_g2, delta_g2
        self.gamma_abc_g1 = np.asarray(gamma_abc_g1, dtype=np.uint64).reshape(-1, 12)
        self.eta_gamma_inv_g1 = eta_gamma_inv_g1
        self.commit_witness_count = commit_witness_count


def _calculate_coeff(curve, initial_point, initial_scalar, query_handle, query0, vk_param, assignment):
    # prover.rs:585-594:  initial + query[0] + msm(query[1..], assignment) + vk_param
    acc = query_handle.msm_bigint(assignment, offset=1)
    rest = lincomb(curve, [initial_point, query0, vk_param], [initial_scalar, 1, 1])
    return sharded.fold(curve, np.stack([acc, rest]))


def create_proof(pk, r, s, v, h, input_assignment_with_one, witness_assignment):
    """prover.rs:267-383.  Returns the proof (a, b, c, d) as affine ABI points."""
    vk = pk.vk
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 4)
    wit = np.ascontiguousarray(witness_assignment, dtype=np.uint64).reshape(-1, 4)
    inp = np.ascontiguousarray(input_assignment_with_one, dtype=np.uint64).reshape(-1, 4)
    cw = vk.commit_witness_count
    h_acc = pk.h_query.msm_bigint(h)                               # :286  (h_query has D-1 points: truncation)
    committed, uncommitted = wit[:cw], wit[cw:]
    l_aux_acc = pk.l_query.msm_bigint(uncommitted)                 # :299
    assignment = np.concatenate([inp[1:], wit])                    # :319-321
    g_a = _calculate_coeff(M.G1, pk.delta_g1, r, pk.a_query, pk.a0, vk.alpha_g1, assignment)           # :325-326
    if r % R_MOD != 0:
        g1_b = _calculate_coeff(M.G1, pk.delta_g1, s, pk.b_g1_query, pk.b1_0, pk.beta_g1, assignment)  # :330-336
    else:
        g1_b = np.zeros(18, dtype=np.uint64); g1_b[:12] = 0
    g2_b = _calculate_coeff(M.G2, vk.delta_g2, s, pk.b_g2_query, pk.b2_0, vk.beta_g2, assignment)      # :343-344
    # g_c = s g_a + r g1_b - rs delta + l_aux + h_acc - v (eta/delta)    :350-355
    small = lincomb(M.G1, [_affine(M.G1, g_a), _affine(M.G1, g1_b), pk.delta_g1, pk.eta_delta_inv_g1], [s, r, -(r * s), -v])
    g_c = sharded.fold(M.G1, np.stack([small, l_aux_acc, h_acc]))
    # g_d = msm(gamma_abc[len(inputs) .. + cw], committed) + v (eta/gamma)    :361-368
    src = vk.gamma_abc_g1[len(inp):len(inp) + cw]
    pts = np.concatenate([src, vk.eta_gamma_inv_g1.reshape(1, 12)])
    g_d = M.msm_bigint(M.G1, pts, np.concatenate([committed, _sc(v).reshape(1, 4)]))
    return {"a": _affine(M.G1, g_a), "b": _affine(M.G2, g2_b), "c": _affine(M.G1, g_c), "d": _affine(M.G1, g_d)}


def prepare_verifying_key(vk):
    """verifier.rs:17-25"""
    return {"vk": vk, "alpha_g1_beta_g2": pairing.multi_pairing(vk.alpha_g1.reshape(1, 12), vk.beta_g2.reshape(1, 24)),
            "gamma_g2_neg": _neg_affine(M.G2, vk.gamma_g2), "delta_g2_neg": _neg_affine(M.G2, vk.delta_g2)}


def calculate_d(pvk, proof, public_inputs):
    """verifier.rs:29-50,101-109: gamma_abc[0] + sum x_j gamma_abc[1+j] + proof.d"""
    vk = pvk["vk"]
    pub = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    if len(pub) + 1 > len(vk.gamma_abc_g1):
        raise ValueError("MalformedVerifyingKey")
    pts = np.concatenate([vk.gamma_abc_g1[:1 + len(pub)], proof["d"].reshape(1, 12)])
    sc = np.concatenate([_sc(1).reshape(1, 4), pub, _sc(1).reshape(1, 4)])
    return _affine(M.G1, M.msm_bigint(M.G1, pts, sc))


def verify_proof(pvk, proof, public_inputs):
    """verifier.rs:62-99: e(A, B) e(C, -delta) e(d, -gamma) == e(alpha, beta)"""
    d = calculate_d(pvk, proof, public_inputs)
    ps = np.stack([proof["a"], proof["c"], d])
    qs = np.stack([proof["b"], pvk["delta_g2_neg"], pvk["gamma_g2_neg"]])
    gt = pairing.multi_pairing(ps, qs)
    if gt is None:
        raise ValueError("UnexpectedIdentity")
    return bool((gt == pvk["alpha_g1_beta_g2"]).all())
