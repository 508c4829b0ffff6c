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
import ctypes as C
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
    """sum scalars[i] * points[i] for the handful of points around the MSMs (the reference: `mul_bigint` on the CPU, prover.rs:309-313,350-355).
    Up to 16 terms: dgpu_lincomb_* (host arithmetic inside the library, no device launch); more: the MSM entry point.  Normalised Jacobian."""
    pts = np.ascontiguousarray(np.stack([np.asarray(p, dtype=np.uint64) for p in points]))
    sc = np.ascontiguousarray(np.stack([_sc(s) for s in scalars]))
    if len(pts) > 16:
        return M.msm_bigint(curve, pts, sc)
    inf = np.ascontiguousarray((~pts.any(axis=1)).astype(np.uint8))          # the ABI's affine identity: all-zero words
    out = np.zeros(curve.JW, dtype=np.uint64)
    rc = curve.fn("dgpu_lincomb_%s")(pts.ctypes.data_as(C.c_void_p), inf.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), len(pts), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise M.DockGpuError(rc, "dgpu_lincomb")
    return out


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

    @classmethod
    def from_device(cls, vk, beta_g1, delta_g1, eta_delta_inv_g1, a0, b1_0, b2_0, a_query, b_g1_query, b_g2_query, h_query, l_query):
        """queries already resident (DeviceBases produced by the generator); a0 / b1_0 / b2_0 = query[0] on the host"""
        self = cls.__new__(cls)
        self.vk = vk
        self.beta_g1, self.delta_g1, self.eta_delta_inv_g1 = beta_g1, delta_g1, eta_delta_inv_g1
        self.a0, self.b1_0, self.b2_0 = a0, b1_0, b2_0
        self.a_query, self.b_g1_query, self.b_g2_query, self.h_query, self.l_query = a_query, b_g1_query, b_g2_query, h_query, l_query
        return self


class VerifyingKey:
    def __init__(self, alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1, eta_gamma_inv_g1, commit_witness_count):
        self.alpha_g1, self.beta_g2, self.gamma_g2, self.delta_g2 = alpha_g1, beta_g2, gamma_g2, delta_g2
        self.gamma_abc_g1 = np.asarray(gamma_abc_g1, dtype=np.uint64).reshape(-1, 12)
        self.eta_gamma_inv_g1 = eta_gamma_inv_g1
        self.commit_witness_count = commit_witness_count


def evaluate_all_lagrange_coefficients(D, omega, t):
    """ark-poly Radix2EvaluationDomain::evaluate_all_lagrange_coefficients for t outside the domain:
    u_i = Z(t) w^i / (D (t - w^i)),  one batch inversion"""
    zt = (pow(t, D, R_MOD) - 1) % R_MOD
    ws, w = [], 1
    for _ in range(D):
        ws.append(w); w = w * omega % R_MOD
    den = [(t - x) % R_MOD for x in ws]
    pre, acc = [], 1
    for x in den:
        pre.append(acc); acc = acc * x % R_MOD
    inv = pow(acc, R_MOD - 2, R_MOD)
    k = zt * pow(D, R_MOD - 2, R_MOD) % R_MOD
    u = [0] * D
    for i in range(D - 1, -1, -1):
        u[i] = inv * pre[i] % R_MOD * ws[i] % R_MOD * k % R_MOD
        inv = inv * den[i] % R_MOD
    return u, zt


def instance_map_with_evaluation(A, B, C, num_instance_variables, num_witness_variables, t):
    """LibsnarkReduction::instance_map_with_evaluation (legogroth16/src/r1cs_to_qap.rs:105-147).
    A, B, C: one list of (coeff, variable index) per constraint.  Returns (a, b, c, zt, qap_num_variables, domain_size)."""
    num_constraints = len(A)
    D = 1
    while D < num_constraints + num_instance_variables:
        D *= 2
    omega = pow(7, (R_MOD - 1) // D, R_MOD)     # Fr::GENERATOR = 7
    u, zt = evaluate_all_lagrange_coefficients(D, omega, t)
    V = (num_instance_variables - 1) + num_witness_variables
    a, b, c = [0] * (V + 1), [0] * (V + 1), [0] * (V + 1)
    for j in range(num_instance_variables):
        a[j] = u[num_constraints + j]
    for i in range(num_constraints):
        ui = u[i]
        for co, idx in A[i]: a[idx] = (a[idx] + ui * co) % R_MOD
        for co, idx in B[i]: b[idx] = (b[idx] + ui * co) % R_MOD
        for co, idx in C[i]: c[idx] = (c[idx] + ui * co) % R_MOD
    return a, b, c, zt, V, D


def generate_parameters(A, B, C, num_instance_variables, num_witness_variables, commit_witness_count,
                        alpha, beta, gamma, delta, eta, t, g1_generator, g2_generator):
    """generate_parameters_and_extra_info_with_qap (legogroth16/src/generator.rs:245-442) with the toxic waste and the
    evaluation point `t` passed in (the reference draws them from `rng`, :220-232, :283).  Scalars are host integers as in
    the reference (rayon field arithmetic); every FixedBase::msm + normalize_batch (:335-399,424-431) is one device
    WindowTable product whose output stays in HBM as the prover's bases handle."""
    from . import fixed_base as FB
    if num_witness_variables < commit_witness_count:
        raise ValueError("InsufficientWitnessesForCommitment(%d, %d)" % (num_witness_variables, commit_witness_count))   # generator.rs:289-294
    a, b, c, zt, V, D = instance_map_with_evaluation(A, B, C, num_instance_variables, num_witness_variables, t)
    n = num_instance_variables + commit_witness_count
    gi, di = pow(gamma, R_MOD - 2, R_MOD), pow(delta, R_MOD - 2, R_MOD)
    mix = [(beta * x + alpha * y + z) % R_MOD for x, y, z in zip(a, b, c)]
    gamma_abc = [m * gi % R_MOD for m in mix[:n]]                       # :316-320
    l = [m * di % R_MOD for m in mix]                                    # :322-326
    hq, k = [], zt * di % R_MOD                                          # r1cs_to_qap.rs:212-223, max_power = m_raw - 1
    for _ in range(D - 1):
        hq.append(k); k = k * t % R_MOD
    with FB.WindowTable(M.G2, g2_generator) as t2, FB.WindowTable(M.G1, g1_generator) as t1:
        b_g2_query = t2.multiply_many_to_bases(b)                        # :337-339
        small2, _ = t2.multiply_many([beta, delta, gamma, b[0]])         # :351,353,405 + query[0] for calculate_coeff
        a_query = t1.multiply_many_to_bases(a)                           # :357
        b_g1_query = t1.multiply_many_to_bases(b)                        # :363
        h_query = t1.multiply_many_to_bases(hq)                          # :369-376
        l_query = t1.multiply_many_to_bases(l[n:])                       # :382
        gamma_abc_g1, _ = t1.multiply_many(gamma_abc)                    # :406
        small1, _ = t1.multiply_many([alpha, beta, delta, eta * gi % R_MOD, eta * di % R_MOD, a[0], b[0]])   # :349-352,411,433
    vk = VerifyingKey(small1[0], small2[0], small2[2], small2[1], gamma_abc_g1, small1[3], commit_witness_count)
    pk = ProvingKey.from_device(vk, small1[1], small1[2], small1[4], small1[5], small1[6], small2[3],
                                a_query, b_g1_query, b_g2_query, h_query, l_query)
    return pk, num_instance_variables


_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(12)
    return _POOL


def _calculate_coeff(curve, initial_point, initial_scalar, query_handle, query0, vk_param, assignment, soff=0):
    # prover.rs:585-594:  initial + query[0] + msm(query[1..], assignment) + vk_param     (assignment: resident scalars from offset `soff`)
    acc = query_handle.msm_resident(assignment, n=min(assignment.n - soff, query_handle.n - 1), base_offset=1, scalar_offset=soff)
    rest = lincomb(curve, [initial_point, query0, vk_param], [initial_scalar, 1, 1])
    return sharded.fold(curve, np.stack([acc, rest]))


def create_proof(pk, r, s, v, h, input_assignment_with_one, witness_assignment, resident_z=None, share_sort=True):
    """prover.rs:267-383.  Returns the proof (a, b, c, d) as affine ABI points.  `h`: canonical limbs or the DeviceScalars
    the witness map left in HBM (qap.witness_map(..., resident=True)).  `resident_z`: the whole assignment z = instance ++ witness already on
    the device (then `assignment` = z[1..] is that handle at scalar offset 1 and nothing is uploaded here).  `share_sort`: one partition sort for
    the A / B-in-G1 / B-in-G2 MSMs when their queries are tables of one shape (False: every MSM sorts for itself, for measurements)."""
    vk = pk.vk
    own_h = []

    def resolve_h():
        """h as resident scalars: a DeviceScalars, host limbs, or a callable that produces either (the witness map, run inside the h job so
        that it overlaps the assignment upload and the four MSMs that do not depend on it)"""
        x = h() if callable(h) else h
        if isinstance(x, M.DeviceScalars):
            if callable(h):
                own_h.append(x)
            return x
        x = M.DeviceScalars(np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4)); own_h.append(x)
        return x
    wit = np.ascontiguousarray(witness_assignment, dtype=np.uint64).reshape(-1, 4)
    inp = np.ascontiguousarray(input_assignment_with_one, dtype=np.uint64).reshape(-1, 4)
    cw = vk.commit_witness_count
    committed = wit[:cw]
    # one upload serves four MSMs: `assignment` = inputs[1..] ++ witnesses (:319-321) and `aux` of :299 is its suffix
    pool = _pool()
    futs, state = [], {"shared": None, "assignment": None}

    def submit(fn):
        f = pool.submit(fn); futs.append(f); return f
    try:
        return _create_proof_body(pk, vk, r, s, v, inp, wit, cw, committed, resident_z, share_sort, resolve_h, submit, state)
    finally:
        # every job has ended (whatever it raised) before anything it reads is released: a failing MSM must not leave its siblings running on
        # a freed sort / assignment, nor leak them
        for f in futs:
            try:
                f.result()
            except Exception:
                pass
        if state["shared"] is not None:
            state["shared"].free()
        if resident_z is None and state["assignment"] is not None:
            state["assignment"].free()
        for x in own_h:
            x.free()


def _create_proof_body(pk, vk, r, s, v, inp, wit, cw, committed, resident_z, share_sort, resolve_h, submit, state):
    import types
    pool = types.SimpleNamespace(submit=submit)      # (the body below was written against an executor: same calls, every future is tracked by the caller)
    def h_job():
        hs = resolve_h()
        return pk.h_query.msm_resident(hs, n=min(pk.h_query.n, hs.n))                               # :286  (h_query has D-1 points: truncation)
    f_h = pool.submit(h_job)                                                                        # runs while the assignment uploads
    if resident_z is None:
        assignment, a0 = M.DeviceScalars.from_parts([inp[1:], wit]), 0
        state["assignment"] = assignment
    else:
        assignment, a0 = resident_z, 1
    n_aux, aux_at = len(wit) - cw, a0 + len(inp) - 1 + cw
    # the five large MSMs are independent (the reference runs each under rayon, one after the other): issue them from host
    # threads so that the latency-bound tail of one overlaps the bulk of the next (the library keeps 6 calls in flight per device)
    # g_d = msm(gamma_abc[len(inputs) .. + cw], committed) + v (eta/gamma)    :361-368  (independent of the large MSMs: issued with them)
    src = vk.gamma_abc_g1[len(inp):len(inp) + cw]
    d_pts = np.concatenate([src, vk.eta_gamma_inv_g1.reshape(1, 12)])
    d_sc = np.concatenate([committed, _sc(v).reshape(1, 4)])
    def coeff_msm(query):            # msm(query[1..], assignment) of calculate_coeff (:592)
        return lambda: query.msm_resident(assignment, n=min(assignment.n - a0, query.n - 1), base_offset=1, scalar_offset=a0)
    with_b1 = r % R_MOD != 0                                                                          # :330-336
    # A, B in G1 and B in G2 multiply the same assignment by three queries of equal length: held as tables of one shape they share ONE
    # partition sort (dgpu_scalars_sort), each MSM then starts at its accumulation
    shared = None
    if share_sort and pk.a_query.same_table_shape(pk.b_g2_query) and (not with_b1 or pk.a_query.same_table_shape(pk.b_g1_query)):
        shared = state["shared"] = M.SortedScalars(pk.a_query, assignment, min(assignment.n - a0, pk.a_query.n - 1), base_offset=1, scalar_offset=a0)
        coeff_msm = lambda query: (lambda: query.msm_sorted(shared))
    # ... and the l_query MSM (the assignment minus its first len(inp) - 1 + cw entries against a table that many + 1 rows shorter) joins when
    # its table has the same window geometry: its rows are the a_query's rows from the (len(inp) + cw)-th on
    l_shift = len(inp) + cw
    sa, sl_ = pk.a_query.table_shape(), pk.l_query.table_shape() if isinstance(pk.l_query, M.DeviceBases) else None
    l_shares = shared is not None and sl_ is not None and sl_[1:] == sa[1:] and sl_[0] + l_shift == sa[0] and n_aux == pk.l_query.n and shared.n == pk.a_query.n - 1
    # Issue order = the order in which the accumulations get the chip.  The G2 MSM is the longest call and ends in ~3 ms of latency-bound
    # kernels (fix-up, bucket reduction): first in, its tail runs under the G1 MSMs instead of after them.
    f_b2 = pool.submit(coeff_msm(pk.b_g2_query))                                                      # :343-344
    f_a = pool.submit(coeff_msm(pk.a_query))                                                          # :325-326
    f_b1 = pool.submit(coeff_msm(pk.b_g1_query)) if with_b1 else None
    f_l = pool.submit((lambda: pk.l_query.msm_sorted(shared, row_shift=l_shift)) if l_shares else
                      (lambda: pk.l_query.msm_resident(assignment, n=min(pk.l_query.n, n_aux), scalar_offset=aux_at)))   # :299
    # The O(1)-sized pieces that depend on no MSM result (one job, one call in flight: they hide behind the large MSMs instead of forming
    # a ~2 ms chain of tiny launches after the last of them): the constant parts of calculate_coeff, of g_c, and g_d.
    def constants():
        return (lincomb(M.G1, [pk.delta_g1, pk.a0, vk.alpha_g1], [r, 1, 1]),
                lincomb(M.G2, [vk.delta_g2, pk.b2_0, vk.beta_g2], [s, 1, 1]),
                lincomb(M.G1, [pk.delta_g1, pk.b1_0, pk.beta_g1], [s, 1, 1]) if with_b1 else None,
                lincomb(M.G1, [pk.delta_g1, pk.eta_delta_inv_g1], [-(r * s), -v]),
                M.msm_bigint(M.G1, d_pts, d_sc))
    rest_a, rest_b2, rest_b1, rest_c, g_d = pool.submit(constants).result()
    g_a = sharded.fold(M.G1, np.stack([f_a.result(), rest_a]))
    g1_b = sharded.fold(M.G1, np.stack([f_b1.result(), rest_b1])) if with_b1 else np.zeros(18, dtype=np.uint64)
    # s g_a + r g1_b needs only the two G1 results: computed while the G2 MSM is still in flight
    sa_rb = lincomb(M.G1, [_affine(M.G1, g_a), _affine(M.G1, g1_b)], [s, r])
    g2_b = sharded.fold(M.G2, np.stack([f_b2.result(), rest_b2]))
    l_aux_acc, h_acc = f_l.result(), f_h.result()
    # g_c = s g_a + r g1_b - rs delta + l_aux + h_acc - v (eta/delta)    :350-355
    g_c = sharded.fold(M.G1, np.stack([sa_rb, rest_c, l_aux_acc, h_acc]))
    return {"a": _affine(M.G1, g_a), "b": _affine(M.G2, g2_b), "c": _affine(M.G1, g_c), "d": _affine(M.G1, g_d)}


def _lego_pk_struct(pk):
    """the dgpu_lego_pk of a ProvingKey (+ the arrays it points into, which must outlive the call)"""
    from ._native import LegoPk
    vk = pk.vk
    keep = {k: np.ascontiguousarray(v, dtype=np.uint64) for k, v in (
        ("alpha_g1", vk.alpha_g1), ("beta_g1", pk.beta_g1), ("delta_g1", pk.delta_g1), ("eta_delta_inv_g1", pk.eta_delta_inv_g1),
        ("eta_gamma_inv_g1", vk.eta_gamma_inv_g1), ("beta_g2", vk.beta_g2), ("delta_g2", vk.delta_g2), ("a0", pk.a0), ("b1_0", pk.b1_0), ("b2_0", pk.b2_0),
        ("gamma_abc_g1", vk.gamma_abc_g1))}
    st = LegoPk()
    st.a_query, st.b_g1_query, st.b_g2_query, st.h_query, st.l_query = (q.handle for q in (pk.a_query, pk.b_g1_query, pk.b_g2_query, pk.h_query, pk.l_query))
    for k, a in keep.items():
        setattr(st, k, a.ctypes.data)
    st.gamma_abc_len = len(vk.gamma_abc_g1)
    st.commit_witness_count = vk.commit_witness_count
    return st, keep


def prove_abi(pk, r, s, v, assignment_with_one, n_inst, circuit=None, h=None, montgomery=False):
    """dgpu_legogroth16_prove: prover.rs:267-383 (and the witness map in front of it when `circuit` is a resident qap.DeviceR1cs) as ONE call of
    the C ABI — what a Rust shim binds.  `h`: a DeviceScalars holding the h coefficients when no circuit is given."""
    z = np.ascontiguousarray(assignment_with_one, dtype=np.uint64).reshape(-1, 4)
    st, keep = _lego_pk_struct(pk)
    a, b, c, d = np.zeros(12, np.uint64), np.zeros(24, np.uint64), np.zeros(12, np.uint64), np.zeros(12, np.uint64)
    inf = np.zeros(4, np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    raw = lambda x: np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)      # (any value below 2^256: reduced inside)
    rc = M.lib().dgpu_legogroth16_prove(C.byref(st), circuit.handle if circuit is not None else 0, h.handle if h is not None else 0, p(z), len(z), n_inst,
                                        int(montgomery), p(raw(r)), p(raw(s)), p(raw(v)), p(a), p(b), p(c), p(d), p(inf))
    del keep
    if rc:
        raise M.DockGpuError(rc, "dgpu_legogroth16_prove")
    return {"a": a, "b": b, "c": c, "d": d}


class HostProvingKey:
    """A proving key the way the reference holds it (legogroth16/src/data_structures.rs `ProvingKeyCommon`): the five queries as arrays of ark-ec Affine
    structs in HOST memory (msm.to_affine_structs), nothing uploaded by the caller.  dgpu_legogroth16_prove_host resolves them through the library's
    resident-bases cache."""

    def __init__(self, vk, beta_g1, delta_g1, eta_delta_inv_g1, a_query, b_g1_query, b_g2_query, h_query, l_query, a0=None, b1_0=None, b2_0=None):
        self.vk, self.beta_g1, self.delta_g1, self.eta_delta_inv_g1 = vk, beta_g1, delta_g1, eta_delta_inv_g1
        st = lambda curve, q: q if q.dtype.names else M.to_affine_structs(curve, q)
        self.a_query, self.b_g1_query, self.h_query, self.l_query = (st(M.G1, q) for q in (a_query, b_g1_query, h_query, l_query))
        self.b_g2_query = st(M.G2, b_g2_query)
        unpack = lambda q: np.concatenate([q["x"][0], q["y"][0]])
        # query[0] of a / b_g1 / b_g2 (calculate_coeff's `el`, prover.rs:591); given explicitly by a synthetic key whose row 0 is not the element it adds (bench.py)
        self.a0 = unpack(self.a_query) if a0 is None else a0
        self.b1_0 = unpack(self.b_g1_query) if b1_0 is None else b1_0
        self.b2_0 = unpack(self.b_g2_query) if b2_0 is None else b2_0


def prove_host(hpk, r, s, v, h, instance_with_one, witness, montgomery=False, h_montgomery=False, circuit=None):
    """dgpu_legogroth16_prove_host: create_proof_and_committed_witnesses_with_assignment (prover.rs:267-383) for a host-held key; h = the coefficients
    QAP::witness_map returned (host array), or None with `circuit` a resident qap.DeviceR1cs (the witness map runs inside the call)."""
    from ._native import LegoPkHost
    vk = hpk.vk
    keep = {k: np.ascontiguousarray(a, dtype=np.uint64) for k, a in (
        ("alpha_g1", vk.alpha_g1), ("beta_g1", hpk.beta_g1), ("delta_g1", hpk.delta_g1), ("eta_delta_inv_g1", hpk.eta_delta_inv_g1),
        ("eta_gamma_inv_g1", vk.eta_gamma_inv_g1), ("beta_g2", vk.beta_g2), ("delta_g2", vk.delta_g2), ("a0", hpk.a0), ("b1_0", hpk.b1_0), ("b2_0", hpk.b2_0),
        ("gamma_abc_g1", vk.gamma_abc_g1))}
    st = LegoPkHost()
    for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
        q = getattr(hpk, name); f = q.dtype.fields; view = getattr(st, name)
        view.p, view.stride, view.x_off, view.y_off, view.inf_off, view.n = q.ctypes.data, q.dtype.itemsize, f["x"][1], f["y"][1], f["infinity"][1], len(q)
    for k, a in keep.items():
        setattr(st, k, a.ctypes.data)
    st.gamma_abc_len = len(vk.gamma_abc_g1)
    st.commit_witness_count = vk.commit_witness_count
    h = None if h is None else np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 4)
    inst = np.ascontiguousarray(instance_with_one, dtype=np.uint64).reshape(-1, 4)
    wit = np.ascontiguousarray(witness, dtype=np.uint64).reshape(-1, 4)
    a, b, c, d = np.zeros(12, np.uint64), np.zeros(24, np.uint64), np.zeros(12, np.uint64), np.zeros(12, np.uint64)
    inf = np.zeros(4, np.uint8)
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    raw = lambda x: np.array([(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    rc = M.lib().dgpu_legogroth16_prove_host(C.byref(st), circuit.handle if circuit is not None else 0, None if h is None else p(h), 0 if h is None else len(h), int(h_montgomery), p(inst), len(inst), p(wit), len(wit), int(montgomery),
                                             p(raw(r)), p(raw(s)), p(raw(v)), p(a), p(b), p(c), p(d), p(inf))
    del keep
    if rc:
        raise M.DockGpuError(rc, "dgpu_legogroth16_prove_host")
    return {"a": a, "b": b, "c": c, "d": d}


def create_proof_with_reduction(pk, circuit, r, s, v, assignment_with_one, share_sort=True, via_abi=True):
    if via_abi and share_sort:
        return prove_abi(pk, r, s, v, assignment_with_one, circuit.num_inputs, circuit=circuit)
    return create_proof_with_reduction_py(pk, circuit, r, s, v, assignment_with_one, share_sort)


def create_proof_with_reduction_py(pk, circuit, r, s, v, assignment_with_one, share_sort=True):
    """create_proof_with_reduction (prover.rs:153-180): h = QAP::witness_map(cs) then create_proof_with_assignment.  `circuit` is the resident
    R1CS (qap.DeviceR1cs: the matrices of the synthesised constraint system), `assignment_with_one` the full assignment z = instance ++ witness.
    z is uploaded ONCE: the witness map reads it in place (dgpu_witness_map_r1cs_resident), the four assignment MSMs use the same handle at
    scalar offset 1, and h never leaves HBM; the witness map runs inside the h job."""
    z = np.ascontiguousarray(assignment_with_one, dtype=np.uint64).reshape(-1, 4)
    n_inst = circuit.num_inputs
    dz = M.DeviceScalars(z)

    def h():
        _, dh = circuit.witness_map(dz, to_host=False, resident=True)
        return dh
    try:
        return create_proof(pk, r, s, v, h, z[:n_inst], z[n_inst:], resident_z=dz, share_sort=share_sort)
    finally:
        dz.free()


# ---- CP_link, commitment openings, re-randomisation: the small callers on the same path ----------------------------------------------------
def generate_link_keys(vk, num_instance_variables, pedersen_gens, link_g1, link_g2, k, a):
    """generate_parameters_incl_cp_link_with_qap's subspace-SNARK part (legogroth16/src/generator.rs:166-204): two rows — the Pedersen bases of
    CP_link and gamma_abc_g1[n_inst .. n_inst + cw] | eta_gamma_inv_g1 of proof.d — with the trapdoor (k, a) passed in.
    Returns (link_pp, link_ek, link_vk, link_bases)."""
    from . import link as LK
    cw = vk.commit_witness_count
    gens = np.ascontiguousarray(pedersen_gens, dtype=np.uint64).reshape(-1, 12)
    pp = LK.PP(2, cw + 2, link_g1, link_g2)
    m = LK.SparseMatrix(2, cw + 2)
    m.insert_row_slice(0, 0, list(gens))
    m.insert_row_slice(1, 0, list(vk.gamma_abc_g1[num_instance_variables:num_instance_variables + cw]))
    m.insert_row_slice(1, cw + 1, [vk.eta_gamma_inv_g1])
    ek, lvk = LK.keygen(pp, m, k, a)
    return pp, ek, lvk, gens


def create_proof_incl_cp_link(pk, link_pp, link_ek, link_bases, r, s, v, link_v, h, input_assignment_with_one, witness_assignment):
    """create_proof_incl_cp_link_with_assignment (prover.rs:183-234): the proof, link_d = msm(link_bases, committed ++ [link_v]) (:215)
    and link_pi = PESubspaceSnark::prove(committed ++ [link_v, v]) (:222)"""
    from . import link as LK
    proof = create_proof(pk, r, s, v, h, input_assignment_with_one, witness_assignment)
    wit = np.ascontiguousarray(witness_assignment, dtype=np.uint64).reshape(-1, 4)
    cw = pk.vk.commit_witness_count
    comm = [sum(int(x) << (64 * i) for i, x in enumerate(row)) for row in wit[:cw]]
    bases = np.ascontiguousarray(link_bases, dtype=np.uint64).reshape(-1, 12)
    link_d = _affine(M.G1, M.msm_bigint(M.G1, bases, np.stack([_sc(x) for x in comm + [link_v]])))
    link_pi = LK.prove(link_pp, link_ek, comm + [link_v, v])
    return {"groth16_proof": proof, "link_d": link_d, "link_pi": link_pi}


def verify_link_proof(link_pp, link_vk, proof_with_link):
    """verifier.rs:53-60: the subspace SNARK on [link_d, groth16_proof.d]; raises link.LinkError on failure"""
    from . import link as LK
    LK.verify(link_pp, link_vk, np.stack([proof_with_link["link_d"], proof_with_link["groth16_proof"]["d"]]), proof_with_link["link_pi"])


def verify_link_commitment(cp_link_bases, link_d, witnesses_expected_in_commitment, link_v):
    """prover.rs:385-407"""
    bases = np.ascontiguousarray(cp_link_bases, dtype=np.uint64).reshape(-1, 12)
    if len(witnesses_expected_in_commitment) + 1 > len(bases):
        raise ValueError("VectorLongerThanExpected(%d, %d)" % (len(witnesses_expected_in_commitment) + 1, len(bases)))
    k = len(witnesses_expected_in_commitment) + 1
    if not (np.asarray(link_d) == _affine(M.G1, lincomb(M.G1, list(bases[:k]), list(witnesses_expected_in_commitment) + [link_v]))).all():
        raise ValueError("InvalidLinkCommitment")


def verify_witness_commitment(vk, proof, public_inputs_count, witnesses_expected_in_commitment, v):
    """prover.rs:434-467: proof.d == msm(gamma_abc_g1[1 + inputs ..], committed) + v (eta/gamma)"""
    k = len(witnesses_expected_in_commitment)
    if public_inputs_count + k + 1 > len(vk.gamma_abc_g1):
        raise ValueError("VectorLongerThanExpected(%d, %d)" % (public_inputs_count + k + 1, len(vk.gamma_abc_g1)))
    pts = np.concatenate([vk.gamma_abc_g1[1 + public_inputs_count:1 + public_inputs_count + k], np.asarray(vk.eta_gamma_inv_g1).reshape(1, 12)])
    d = _affine(M.G1, lincomb(M.G1, list(pts), list(witnesses_expected_in_commitment) + [v]))
    if not (np.asarray(proof["d"]) == d).all():
        raise ValueError("InvalidWitnessCommitment")


def verify_commitments(vk, link_bases, proof_with_link, public_inputs_count, witnesses_expected_in_commitment, v, link_v):
    """prover.rs:412-431"""
    verify_link_commitment(link_bases, proof_with_link["link_d"], witnesses_expected_in_commitment, link_v)
    verify_witness_commitment(vk, proof_with_link["groth16_proof"], public_inputs_count, witnesses_expected_in_commitment, v)


def rerandomize_proof(proof, vk, r1, r2):
    """prover.rs:478-510 with the factors passed in (nonzero):  A' = A / r1,  B' = r1 B + r1 r2 (delta + gamma),  C' = C + r2 A,  D' = D + r2 A"""
    if r1 % R_MOD == 0 or r2 % R_MOD == 0:
        raise ValueError("rerandomisation factors must be nonzero")
    r1i = pow(r1, R_MOD - 2, R_MOD)
    a_r2 = _affine(M.G1, lincomb(M.G1, [proof["a"]], [r2]))
    return {"a": _affine(M.G1, lincomb(M.G1, [proof["a"]], [r1i])),
            "b": _affine(M.G2, lincomb(M.G2, [proof["b"], vk.delta_g2, vk.gamma_g2], [r1, r1 * r2, r1 * r2])),
            "c": _affine(M.G1, lincomb(M.G1, [proof["c"], a_r2], [1, 1])),
            "d": _affine(M.G1, lincomb(M.G1, [proof["d"], a_r2], [1, 1]))}


def rerandomize_proof_1(proof, old_v, new_v, vk, eta_delta_inv_g1, r1, r2):
    """prover.rs:514-549: keeps proof.d a commitment to the witnesses (with randomness new_v):
    A' = A / r1,  B' = r1 B + r1 r2 delta,  C' = C + r2 A + (old_v - new_v)(eta/delta),  D' = D + (new_v - old_v)(eta/gamma)"""
    if r1 % R_MOD == 0 or r2 % R_MOD == 0:
        raise ValueError("rerandomisation factors must be nonzero")
    r1i = pow(r1, R_MOD - 2, R_MOD)
    return {"a": _affine(M.G1, lincomb(M.G1, [proof["a"]], [r1i])),
            "b": _affine(M.G2, lincomb(M.G2, [proof["b"], vk.delta_g2], [r1, r1 * r2])),
            "c": _affine(M.G1, lincomb(M.G1, [proof["c"], proof["a"], eta_delta_inv_g1], [1, r2, old_v - new_v])),
            "d": _affine(M.G1, lincomb(M.G1, [proof["d"], vk.eta_gamma_inv_g1], [1, new_v - old_v]))}


# ---- multi-GPU prover: every large MSM chunked over the ranks (SURVEY.md 8e "LegoGroth16 prove") ----------------------------------------
class ShardedProvingKey:
    """This rank's chunk of every proving-key query, resident on its GPU.  Chunks are contiguous and balanced (sharded.chunk_bounds):
    a / b_g1 / b_g2 over query[1..], h over the D - 1 points, l over its own length; query[0] and the O(1) elements stay on the host."""

    def __init__(self, vk, beta_g1, delta_g1, eta_delta_inv_g1, a_query, b_g1_query, b_g2_query, h_query, l_query, world, rank):
        self.vk, self.world, self.rank = vk, world, rank
        self.beta_g1, self.delta_g1, self.eta_delta_inv_g1 = beta_g1, delta_g1, eta_delta_inv_g1
        self.a0, self.b1_0, self.b2_0 = a_query[0].copy(), b_g1_query[0].copy(), b_g2_query[0].copy()
        self.V, self.H, self.L = len(a_query) - 1, len(h_query), len(l_query)
        cb = lambda n: sharded.chunk_bounds(n, world, rank)
        (lo, hi), (hlo, hhi), (llo, lhi) = cb(self.V), cb(self.H), cb(self.L)
        self.a_query = M.DeviceBases(M.G1, a_query[1 + lo:1 + hi])
        self.b_g1_query = M.DeviceBases(M.G1, b_g1_query[1 + lo:1 + hi])
        self.b_g2_query = M.DeviceBases(M.G2, b_g2_query[1 + lo:1 + hi])
        self.h_query = M.DeviceBases(M.G1, h_query[hlo:hhi])
        self.l_query = M.DeviceBases(M.G1, l_query[llo:lhi])


def create_proof_sharded(spk, r, s, v, h, input_assignment_with_one, witness_assignment, device=None):
    """create_proof with the five large MSMs chunked over the ranks: each rank runs them on its chunk of (bases, scalars), ONE all_gather
    moves the five partial points (864 B per rank) and every rank folds them and finishes the proof identically.  The witness map (NTT) is
    not sharded: `h` is the full coefficient vector on every rank."""
    import torch
    import torch.distributed as dist
    vk = spk.vk
    h = np.ascontiguousarray(h, dtype=np.uint64).reshape(-1, 4)
    wit = np.ascontiguousarray(witness_assignment, dtype=np.uint64).reshape(-1, 4)
    inp = np.ascontiguousarray(input_assignment_with_one, dtype=np.uint64).reshape(-1, 4)
    cw = vk.commit_witness_count
    assignment = np.concatenate([inp[1:], wit])
    aux = wit[cw:]
    cb = lambda n: sharded.chunk_bounds(n, spk.world, spk.rank)
    (lo, hi), (hlo, hhi), (llo, lhi) = cb(spk.V), cb(spk.H), cb(spk.L)
    pool = _pool()
    jobs = [lambda: spk.h_query.msm_bigint(h[hlo:min(hhi, len(h))]),
            lambda: spk.l_query.msm_bigint(aux[llo:lhi]),
            lambda: spk.a_query.msm_bigint(assignment[lo:hi]),
            (lambda: spk.b_g1_query.msm_bigint(assignment[lo:hi])) if r % R_MOD != 0 else (lambda: np.zeros(18, dtype=np.uint64)),
            lambda: spk.b_g2_query.msm_bigint(assignment[lo:hi])]
    parts = [f.result() for f in [pool.submit(j) for j in jobs]]
    local = np.concatenate(parts)                                   # 4 x 18 + 36 limbs
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.from_numpy(local.view(np.int64).copy())
        if device is not None:
            t = t.to(device)
        buf = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(buf, t)
        allp = np.stack([b.cpu().numpy().view(np.uint64) for b in buf])
    else:
        allp = local.reshape(1, -1)
    h_acc, l_aux_acc, acc_a, acc_b1 = (sharded.fold(M.G1, allp[:, 18 * k:18 * k + 18]) for k in range(4))
    acc_b2 = sharded.fold(M.G2, allp[:, 72:108])
    coeff = lambda curve, init, k, q0, vkp, acc: sharded.fold(curve, np.stack([acc, lincomb(curve, [init, q0, vkp], [k, 1, 1])]))   # prover.rs:585-594
    g_a = coeff(M.G1, spk.delta_g1, r, spk.a0, vk.alpha_g1, acc_a)
    g1_b = coeff(M.G1, spk.delta_g1, s, spk.b1_0, spk.beta_g1, acc_b1) if r % R_MOD != 0 else np.zeros(18, dtype=np.uint64)
    g2_b = coeff(M.G2, vk.delta_g2, s, spk.b2_0, vk.beta_g2, acc_b2)
    small = lincomb(M.G1, [_affine(M.G1, g_a), _affine(M.G1, g1_b), spk.delta_g1, spk.eta_delta_inv_g1], [s, r, -(r * s), -v])
    g_c = sharded.fold(M.G1, np.stack([small, l_aux_acc, h_acc]))
    src = vk.gamma_abc_g1[len(inp):len(inp) + cw]
    pts = np.concatenate([src, vk.eta_gamma_inv_g1.reshape(1, 12)])
    g_d = M.msm_bigint(M.G1, pts, np.concatenate([wit[:cw], _sc(v).reshape(1, 4)]))
    return {"a": _affine(M.G1, g_a), "b": _affine(M.G2, g2_b), "c": _affine(M.G1, g_c), "d": _affine(M.G1, g_d)}


def prepare_verifying_key(vk):
    """verifier.rs:17-25"""
    neg_pc = pairing.G2Prepared.from_affine(np.stack([_neg_affine(M.G2, vk.gamma_g2), _neg_affine(M.G2, vk.delta_g2)]))   # :22-23: held PREPARED
    return {"vk": vk, "alpha_g1_beta_g2": pairing.multi_pairing(vk.alpha_g1.reshape(1, 12), vk.beta_g2.reshape(1, 24)),
            "gamma_g2_neg_pc": neg_pc[0], "delta_g2_neg_pc": neg_pc[1]}


def O_limbs_to_int(l):
    return int(l[0]) | (int(l[1]) << 64) | (int(l[2]) << 128) | (int(l[3]) << 192)


def calculate_d(pvk, proof, public_inputs):
    """verifier.rs:29-50,101-109: gamma_abc[0] + sum x_j gamma_abc[1+j] + proof.d"""
    vk = pvk["vk"]
    pub = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    if len(pub) + 1 > len(vk.gamma_abc_g1):
        raise ValueError("MalformedVerifyingKey")
    pts = np.concatenate([vk.gamma_abc_g1[:1 + len(pub)], proof["d"].reshape(1, 12)])
    sc = np.concatenate([_sc(1).reshape(1, 4), pub, _sc(1).reshape(1, 4)])
    if len(pts) <= 16:                  # a handful of public inputs: host arithmetic (the reference: a short CPU msm), no device round trip
        return _affine(M.G1, lincomb(M.G1, list(pts), [O_limbs_to_int(x) for x in sc]))
    return _affine(M.G1, M.msm_bigint(M.G1, pts, sc))


def calculate_d_batch(pvk, proofs, public_inputs):
    """calculate_d for many proofs of one circuit: gamma_abc[0] + sum_j x_ij gamma_abc[1 + j] + proof_i.d for every i, as 1 + k batched
    device calls (one same-base `mul_add` per public input) instead of one host scalar multiplication per proof and input.  (n, 12) affine."""
    from .aggregation import ops
    vk = pvk["vk"]
    n = len(proofs)
    pubs = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in public_inputs]
    k = len(pubs[0]) if n else 0
    if any(len(x) != k for x in pubs):
        raise ValueError("public inputs of unequal length")
    if k + 1 > len(vk.gamma_abc_g1):
        raise ValueError("MalformedVerifyingKey")
    d = ops.mul_add(M.G1, np.tile(vk.gamma_abc_g1[0], (n, 1)), 1, np.stack([p["d"] for p in proofs]))
    for j in range(k):
        d = ops.mul_add(M.G1, np.tile(vk.gamma_abc_g1[1 + j], (n, 1)), [O_limbs_to_int(x[j]) for x in pubs], d)
    return d


def verify_proofs_batch(pvk, proofs, public_inputs, random):
    """Many LegoGroth16 proofs of one circuit through ONE lazy RandomizedPairingChecker (what proof_system/src/verifier.rs:1829-1835 does with
    the statements of a composite proof): three pairs per proof — (A_i, B_i), (C_i, -delta), (d_i, -gamma), the last two on the prepared
    key — and the target e(alpha, beta) once per proof, all folded by the checker's powers of `random`.  True iff every proof verifies
    (up to the checker's 2^-255 soundness error)."""
    from .pairing_check import RandomizedPairingChecker
    chk = RandomizedPairingChecker(random, True)
    ds = calculate_d_batch(pvk, proofs, public_inputs)
    for p, d in zip(proofs, ds):
        chk.add_multiple_sources_and_target(np.stack([p["a"], p["c"], d]), [p["b"].reshape(1, 24), pvk["delta_g2_neg_pc"], pvk["gamma_g2_neg_pc"]], pvk["alpha_g1_beta_g2"])
    return chk.verify()


def verify_proofs_batch_merged(pvk, proofs, public_inputs, random):
    """The same batch check with the pairs that share a G2 operand merged BEFORE the pairing (the classical Groth16 batch verifier; the
    reference's checker keeps all 3 N pairs): with m_i = random^i,
        prod_i e(m_i A_i, B_i) * e(sum_i m_i C_i, -delta) * e(sum_i m_i (gamma_abc_0 + sum_j x_ij gamma_abc_j + d_i), -gamma) == e(alpha, beta)^(sum m_i)
    i.e. N scalings, two variable-base MSMs of N (+ k + 1) terms — the hot path of this library — and ONE Miller loop over N + 2 pairs, two of
    them on the prepared key.  Accepts exactly what verify_proofs_batch accepts (up to the 2^-255 soundness error of the random combination)."""
    from .pairing_check import g1_scale_each, fp12_pow
    vk = pvk["vk"]
    n = len(proofs)
    if n == 0:
        return True
    pubs = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in public_inputs]
    k = len(pubs[0])
    if len(pubs) != n or any(len(x) != k for x in pubs):
        raise ValueError("public inputs of unequal length")
    if k + 1 > len(vk.gamma_abc_g1):
        raise ValueError("MalformedVerifyingKey")
    rnd = random % R_MOD
    ms, m = [], 1
    for _ in range(n):
        ms.append(m); m = m * rnd % R_MOD
    m_limbs = np.stack([_sc(x) for x in ms])
    a_scaled, a_inf = g1_scale_each(np.stack([p["a"] for p in proofs]), m_limbs, None)
    c_sum = _affine(M.G1, M.msm_bigint(M.G1, np.stack([p["c"] for p in proofs]), m_limbs))
    xs = [[O_limbs_to_int(x[j]) for x in pubs] for j in range(k)]
    d_pts = np.concatenate([vk.gamma_abc_g1[:1 + k], np.stack([p["d"] for p in proofs])])
    d_sc = np.stack([_sc(sum(ms))] + [_sc(sum(mi * xi for mi, xi in zip(ms, col))) for col in xs] + [m_limbs[i] for i in range(n)])
    d_sum = _affine(M.G1, M.msm_bigint(M.G1, d_pts, d_sc))
    ps = np.concatenate([a_scaled, c_sum.reshape(1, 12), d_sum.reshape(1, 12)])
    qs = [np.stack([p["b"] for p in proofs]), pvk["delta_g2_neg_pc"], pvk["gamma_g2_neg_pc"]]
    skip = np.concatenate([np.asarray(a_inf, dtype=np.uint8), np.zeros(2, np.uint8)])
    gt = pairing.final_exponentiation(pairing.multi_miller_loop(ps, qs, skip))
    if gt is None:
        raise ValueError("UnexpectedIdentity")
    return bool((gt == fp12_pow(pvk["alpha_g1_beta_g2"], sum(ms))).all())


def verify_proof_abi(pvk, proof, public_inputs, montgomery=False):
    """verifier.rs:62-99 as ONE call of the C ABI (dgpu_legogroth16_verify): calculate_d on a host core inside the library while the device
    runs the chain of (A, B); same answer as verify_proof below"""
    import ctypes as C
    from ._native import lib, DockGpuError
    vk = pvk["vk"]
    pub = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    gabc = np.ascontiguousarray(vk.gamma_abc_g1, dtype=np.uint64).reshape(-1, 12)
    p_ = lambda a: np.ascontiguousarray(a, dtype=np.uint64).ctypes.data_as(C.c_void_p)
    keep = [np.ascontiguousarray(proof[k], dtype=np.uint64) for k in "abcd"]
    dn, gn = np.ascontiguousarray(pvk["delta_g2_neg_pc"].coeffs.reshape(-1)), np.ascontiguousarray(pvk["gamma_g2_neg_pc"].coeffs.reshape(-1))
    ab = np.ascontiguousarray(pvk["alpha_g1_beta_g2"], dtype=np.uint64)
    ok = C.c_int32(-1)
    rc = lib().dgpu_legogroth16_verify(p_(ab), p_(dn), p_(gn), p_(gabc), len(gabc), p_(keep[0]), p_(keep[1]), p_(keep[2]), p_(keep[3]), None,
                                       pub.ctypes.data_as(C.c_void_p), len(pub), int(montgomery), C.byref(ok))
    if rc == -3 and len(pub) + 1 > len(gabc):
        raise ValueError("MalformedVerifyingKey")
    if rc == -5:
        raise ValueError("UnexpectedIdentity")
    if rc:
        raise DockGpuError(rc, "dgpu_legogroth16_verify")
    return ok.value == 1


def pack_proofs(proofs, public_inputs):
    """the column form dgpu_legogroth16_verify_batch takes: (a: n x 12, b: n x 24, c, d: n x 12, public inputs: n x k x 4) — what a Rust host's
    `&[Proof]` costs microseconds to produce and a list of Python dictionaries a millisecond per thousand proofs"""
    n = len(proofs)
    cols = [np.ascontiguousarray(np.stack([pr[key] for pr in proofs]), dtype=np.uint64) if n else np.zeros((0, w), np.uint64) for key, w in (("a", 12), ("b", 24), ("c", 12), ("d", 12))]
    pubs = np.ascontiguousarray(np.stack([np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in public_inputs]) if n else np.zeros((0, 0, 4), np.uint64))
    if len(pubs) != n:
        raise ValueError("public inputs of unequal length")
    return cols[0], cols[1], cols[2], cols[3], pubs


def verify_proofs_batch_abi(pvk, proofs, public_inputs, random, montgomery=False, packed=None):
    """N proofs of one verifying key through ONE call of the C ABI (dgpu_legogroth16_verify_batch): the merged batch check of
    verify_proofs_batch_merged with its scalings, its two MSMs and its GT power side by side inside the library and no interpreter between the
    pieces.  `random`: the batching scalar (an int, drawn after the proofs are fixed).  packed: pack_proofs(proofs, public_inputs) made ahead."""
    import ctypes as C
    from ._native import lib, DockGpuError
    vk = pvk["vk"]
    a, b, c, d, pubs = packed if packed is not None else pack_proofs(proofs, public_inputs)
    n = len(a)
    k = pubs.shape[1] if n else 0
    gabc = np.ascontiguousarray(vk.gamma_abc_g1, dtype=np.uint64).reshape(-1, 12)
    p_ = lambda x: np.ascontiguousarray(x, dtype=np.uint64).ctypes.data_as(C.c_void_p)
    dn, gn = np.ascontiguousarray(pvk["delta_g2_neg_pc"].coeffs.reshape(-1)), np.ascontiguousarray(pvk["gamma_g2_neg_pc"].coeffs.reshape(-1))
    ab = np.ascontiguousarray(pvk["alpha_g1_beta_g2"], dtype=np.uint64)
    rnd = _sc(random % R_MOD)
    ok = C.c_int32(-1)
    rc = lib().dgpu_legogroth16_verify_batch(p_(ab), p_(dn), p_(gn), p_(gabc), len(gabc), p_(a), p_(b), p_(c), p_(d), n,
                                             pubs.ctypes.data_as(C.c_void_p), k, int(montgomery), p_(rnd), C.byref(ok))
    if rc == -3 and k + 1 > len(gabc):
        raise ValueError("MalformedVerifyingKey")
    if rc == -5:
        raise ValueError("UnexpectedIdentity")
    if rc:
        raise DockGpuError(rc, "dgpu_legogroth16_verify_batch")
    return ok.value == 1


def verify_proof(pvk, proof, public_inputs):
    """verifier.rs:62-99: e(A, B) e(C, -delta) e(d, -gamma) == e(alpha, beta)"""
    d = calculate_d(pvk, proof, public_inputs)
    ps = np.stack([proof["a"], proof["c"], d])
    qs = [proof["b"].reshape(1, 24), pvk["delta_g2_neg_pc"], pvk["gamma_g2_neg_pc"]]      # [b.into(), delta_g2_neg_pc.clone(), gamma_g2_neg_pc.clone()]  :69-76
    gt = pairing.multi_pairing(ps, qs)
    if gt is None:
        raise ValueError("UnexpectedIdentity")
    return bool((gt == pvk["alpha_g1_beta_g2"]).all())
