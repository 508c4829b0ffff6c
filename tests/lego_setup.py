"""Test infrastructure for the LegoGroth16 round trip: a synthetic R1CS (x_i = x_{i-1}^2 + i, the shape of the reference's
`nconstraints` circom fixture, legogroth16/test-vectors/circuits/nconstraints.circom), the trusted setup
(legogroth16/src/generator.rs:245-442 + r1cs_to_qap.rs:105-147,212-223) and the witness map
(r1cs_to_qap.rs:150-210, done here with plain big-integer polynomial arithmetic), all in Python on top of the CPU oracle.
None of this is product code: the product is handed the key, the assignment and h — what the mirrored reference function takes."""
import numpy as np
import oracle_c as O

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
inv = lambda a: pow(a, R - 2, R)


def circuit(m, x0):
    """variables: 0 = one, 1 = out (instance); witnesses x_0 .. x_m at indices 2 .. m+2.  m+1 constraints."""
    xs = [x0 % R]
    for i in range(1, m + 1):
        xs.append((xs[-1] * xs[-1] + i) % R)
    out = xs[-1]
    z = [1, out] + xs
    A, B, C = [], [], []
    w = lambda i: 2 + i
    for i in range(1, m + 1):
        A.append([(1, w(i - 1))]); B.append([(1, w(i - 1))]); C.append([(1, w(i)), ((-i) % R, 0)])
    A.append([(1, w(m))]); B.append([(1, 0)]); C.append([(1, 1)])
    return {"A": A, "B": B, "C": C, "z": z, "n_inst": 2, "n_wit": m + 1, "n_cons": m + 1}


def domain(size):
    D = 1
    while D < size:
        D *= 2
    omega = pow(7, (R - 1) // D, R)          # 7 = Fr::GENERATOR; two-adic root of unity of order D
    return D, omega


def setup(cs, commit_witness_count, seed=1):
    rng = np.random.default_rng(seed)
    rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
    alpha, beta, gamma, delta, eta, t, k1, k2 = (rnd() for _ in range(8))
    D, om = domain(cs["n_cons"] + cs["n_inst"])
    zt = (pow(t, D, R) - 1) % R
    # Lagrange coefficients u_i = zt / (D (t - w^i)) * w^i
    u, wi = [], 1
    dinv = inv(D)
    for i in range(D):
        u.append(zt * dinv % R * wi % R * inv((t - wi) % R) % R)
        wi = wi * om % R
    V = (cs["n_inst"] - 1) + cs["n_wit"]
    a = [0] * (V + 1); b = [0] * (V + 1); c = [0] * (V + 1)
    for j in range(cs["n_inst"]):
        a[j] = u[cs["n_cons"] + j]
    for i in range(cs["n_cons"]):
        for co, idx in cs["A"][i]: a[idx] = (a[idx] + u[i] * co) % R
        for co, idx in cs["B"][i]: b[idx] = (b[idx] + u[i] * co) % R
        for co, idx in cs["C"][i]: c[idx] = (c[idx] + u[i] * co) % R
    n = cs["n_inst"] + commit_witness_count
    gi, di = inv(gamma), inv(delta)
    gamma_abc = [(beta * a[j] + alpha * b[j] + c[j]) % R * gi % R for j in range(n)]
    l = [(beta * a[j] + alpha * b[j] + c[j]) % R * di % R for j in range(V + 1)]
    hq = [zt * di % R * pow(t, i, R) % R for i in range(D - 1)]
    g1 = lambda s: O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(s * k1 % R, 4)))[0]
    g2 = lambda s: O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(s * k2 % R, 4)))[0]
    key = {
        "alpha_g1": g1(alpha), "beta_g1": g1(beta), "beta_g2": g2(beta), "gamma_g2": g2(gamma), "delta_g1": g1(delta), "delta_g2": g2(delta),
        "gamma_abc_g1": np.stack([g1(x) for x in gamma_abc]), "eta_gamma_inv_g1": g1(eta * gi % R), "eta_delta_inv_g1": g1(eta * di % R),
        "a_query": np.stack([g1(x) for x in a]), "b_g1_query": np.stack([g1(x) for x in b]), "b_g2_query": np.stack([g2(x) for x in b]),
        "h_query": np.stack([g1(x) for x in hq]), "l_query": np.stack([g1(x) for x in l[n:]]),
        "commit_witness_count": commit_witness_count, "D": D,
        "_waste": {"alpha": alpha, "beta": beta, "gamma": gamma, "delta": delta, "eta": eta, "t": t, "k1": k1, "k2": k2},
    }
    return key


def witness_map(cs):
    """h = (a b - c) / Z_D as D coefficients (the last one is zero), by naive O(D^2) interpolation and multiplication"""
    D, om = domain(cs["n_cons"] + cs["n_inst"])
    z = cs["z"]
    ev = lambda rows: [sum(co * z[idx] for co, idx in row) % R for row in rows]
    ae, be, ce = ev(cs["A"]) + [0] * (D - cs["n_cons"]), ev(cs["B"]) + [0] * (D - cs["n_cons"]), ev(cs["C"]) + [0] * (D - cs["n_cons"])
    for j in range(cs["n_inst"]):
        ae[cs["n_cons"] + j] = z[j]
    omi, dinv = inv(om), inv(D)
    pw = [pow(omi, k, R) for k in range(D)]

    def interp(e):
        return [sum(e[i] * pw[(i * k) % D] for i in range(D)) % R * dinv % R for k in range(D)]
    ap, bp, cp = interp(ae), interp(be), interp(ce)
    prod = [0] * (2 * D)
    for i, x in enumerate(ap):
        if x:
            for j, y in enumerate(bp):
                prod[i + j] = (prod[i + j] + x * y) % R
    for k in range(D):
        prod[k] = (prod[k] - cp[k]) % R
    h = [prod[k + D] for k in range(D)]
    assert all((prod[k] + prod[k + D]) % R == 0 for k in range(D)), "a*b - c not divisible by Z (unsatisfied R1CS)"
    return h


def scalars(vals):
    return np.stack([O.int_to_limbs(v % R, 4) for v in vals]) if len(vals) else np.zeros((0, 4), np.uint64)
