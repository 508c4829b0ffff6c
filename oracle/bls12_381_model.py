"""BLS12-381 big-integer model — TEST INFRASTRUCTURE ONLY (oracle, never shipped).

A dependency-free Python restatement of the arithmetic that the reference
(docknetwork/crypto) reaches through arkworks 0.4 (`ark-ec`, `ark-ff`,
`ark-bls12-381`; third-party, NOT present under /root/reference — no
Cargo.lock is committed, semver resolves to ark-ec/ark-ff 0.4.2,
ark-bls12-381 0.4.0).  It follows the published algorithms as restated in
SURVEY.md Appendix A/B and is anchored on the reference's call sites:

  * variable-base MSM      utils/src/pairs.rs:143-156, legogroth16/src/prover.rs:286,299,592
  * multi_miller_loop      utils/src/randomized_pairing_check.rs:204-214, legogroth16/src/verifier.rs:62-84
  * final_exponentiation   utils/src/randomized_pairing_check.rs:213

Parity status: **parity unpinned** w.r.t. a real arkworks run (the reference
holds no known-answer vectors for this path and cannot be built here: no Rust
toolchain).  The model is pinned instead by (a) algebraic identities the
reference's own tests assert (utils/src/msm.rs:186-193: msm == sum of
mul_bigint; :268-275 prepared == unprepared pairing), (b) bilinearity, and
(c) agreement with the independent C restatement in oracle/oracle.c.

Only tests/, tests/golden/gen_golden.py, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import this file.
"""

# ----------------------------------------------------------------------------
# constants
# ----------------------------------------------------------------------------
P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
X_ABS = 0xD201000000010000  # |x|, x is negative
X_IS_NEG = True

G1_GEN = (
    0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
    0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
)
G2_GEN = (
    (
        0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
        0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
    ),
    (
        0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
        0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
    ),
)

FP_R = (1 << 384) % P       # Montgomery R for Fq (6x64 limbs)
FR_R = (1 << 256) % R       # Montgomery R for Fr (4x64 limbs)


def fp_inv(a):
    return pow(a, P - 2, P)


# ----------------------------------------------------------------------------
# Fp2 = Fp[u]/(u^2+1)
# ----------------------------------------------------------------------------
def f2(a, b=0):
    return (a % P, b % P)

F2_ZERO = (0, 0)
F2_ONE = (1, 0)

def f2_add(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)
def f2_sub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)
def f2_neg(a): return ((-a[0]) % P, (-a[1]) % P)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)
def f2_sqr(a): return f2_mul(a, a)
def f2_mul_fp(a, k): return ((a[0] * k) % P, (a[1] * k) % P)
def f2_conj(a): return (a[0], (-a[1]) % P)
def f2_inv(a):
    n = fp_inv((a[0] * a[0] + a[1] * a[1]) % P)
    return ((a[0] * n) % P, (-a[1] * n) % P)
def f2_pow(a, e):
    r = F2_ONE
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_sqr(a)
        e >>= 1
    return r

XI = (1, 1)  # cubic/quadratic non-residue 1+u
def f2_mul_xi(a):  # (a0 + a1 u)(1+u) = (a0-a1) + (a0+a1)u
    return ((a[0] - a[1]) % P, (a[0] + a[1]) % P)


# ----------------------------------------------------------------------------
# Fp6 = Fp2[v]/(v^3 - xi);  Fp12 = Fp6[w]/(w^2 - v)
# ----------------------------------------------------------------------------
F6_ZERO = (F2_ZERO, F2_ZERO, F2_ZERO)
F6_ONE = (F2_ONE, F2_ZERO, F2_ZERO)

def f6_add(a, b): return tuple(f2_add(x, y) for x, y in zip(a, b))
def f6_sub(a, b): return tuple(f2_sub(x, y) for x, y in zip(a, b))
def f6_neg(a): return tuple(f2_neg(x) for x in a)
def f6_mul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    c0 = f2_add(f2_mul(a0, b0), f2_mul_xi(f2_add(f2_mul(a1, b2), f2_mul(a2, b1))))
    c1 = f2_add(f2_add(f2_mul(a0, b1), f2_mul(a1, b0)), f2_mul_xi(f2_mul(a2, b2)))
    c2 = f2_add(f2_add(f2_mul(a0, b2), f2_mul(a1, b1)), f2_mul(a2, b0))
    return (c0, c1, c2)
def f6_mul_v(a):  # a * v
    return (f2_mul_xi(a[2]), a[0], a[1])
def f6_inv(a):
    a0, a1, a2 = a
    t0 = f2_sub(f2_sqr(a0), f2_mul_xi(f2_mul(a1, a2)))
    t1 = f2_sub(f2_mul_xi(f2_sqr(a2)), f2_mul(a0, a1))
    t2 = f2_sub(f2_sqr(a1), f2_mul(a0, a2))
    n = f2_add(f2_mul(a0, t0), f2_mul_xi(f2_add(f2_mul(a2, t1), f2_mul(a1, t2))))
    ni = f2_inv(n)
    return (f2_mul(t0, ni), f2_mul(t1, ni), f2_mul(t2, ni))

F12_ONE = (F6_ONE, F6_ZERO)

def f12_mul(a, b):
    a0, a1 = a
    b0, b1 = b
    t0 = f6_mul(a0, b0)
    t1 = f6_mul(a1, b1)
    c0 = f6_add(t0, f6_mul_v(t1))
    c1 = f6_sub(f6_sub(f6_mul(f6_add(a0, a1), f6_add(b0, b1)), t0), t1)
    return (c0, c1)
def f12_sqr(a): return f12_mul(a, a)
def f12_conj(a): return (a[0], f6_neg(a[1]))
def f12_inv(a):
    a0, a1 = a
    n = f6_sub(f6_mul(a0, a0), f6_mul_v(f6_mul(a1, a1)))
    ni = f6_inv(n)
    return (f6_mul(a0, ni), f6_neg(f6_mul(a1, ni)))
def f12_pow(a, e):
    r = F12_ONE
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_sqr(a)
        e >>= 1
    return r
def f12_is_zero(a):
    return all(c == 0 for h in a for q in h for c in q)

# Frobenius coefficients, derived (not copied): gamma_i = xi^(i*(p^k-1)/6)
def _frob_coeffs(k):
    e = (P ** k - 1) // 6
    return [f2_pow(XI, i * e) for i in range(6)]
_FROB = {1: _frob_coeffs(1), 2: _frob_coeffs(2)}

def f12_frob(a, k=1):
    """a^(p^k) for k in {1,2}.  Basis element v^i w^j = w^(2i+j); (w^m)^(p^k) = gamma_m * w^m."""
    g = _FROB[k]
    conj = (lambda t: f2_conj(t)) if (k % 2 == 1) else (lambda t: t)
    (c00, c01, c02), (c10, c11, c12) = a
    n00 = conj(c00)
    n01 = f2_mul(conj(c01), g[2])
    n02 = f2_mul(conj(c02), g[4])
    n10 = f2_mul(conj(c10), g[1])
    n11 = f2_mul(conj(c11), g[3])
    n12 = f2_mul(conj(c12), g[5])
    return ((n00, n01, n02), (n10, n11, n12))

def f12_mul_by_014(f, c0, c1, c4):
    """f * (c0 + c1 v + c4 v w)  — ark-ff Fp12::mul_by_014 (sparse operand at slots 0,1,4)."""
    s = ((c0, c1, F2_ZERO), (F2_ZERO, c4, F2_ZERO))
    return f12_mul(f, s)


# ----------------------------------------------------------------------------
# G1 / G2 affine group law (None = identity).  Generic over a tiny field vtable
# ----------------------------------------------------------------------------
class _FpOps:
    zero = 0
    @staticmethod
    def add(a, b): return (a + b) % P
    @staticmethod
    def sub(a, b): return (a - b) % P
    @staticmethod
    def mul(a, b): return (a * b) % P
    @staticmethod
    def inv(a): return fp_inv(a)
    @staticmethod
    def neg(a): return (-a) % P
    @staticmethod
    def small(k): return k % P

class _Fp2Ops:
    zero = F2_ZERO
    add = staticmethod(f2_add)
    sub = staticmethod(f2_sub)
    mul = staticmethod(f2_mul)
    inv = staticmethod(f2_inv)
    neg = staticmethod(f2_neg)
    @staticmethod
    def small(k): return (k % P, 0)

def _ec_add(F, p, q):
    if p is None: return q
    if q is None: return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if y1 == y2:
            if y1 == F.zero:
                return None
            lam = F.mul(F.mul(F.small(3), F.mul(x1, x1)), F.inv(F.add(y1, y1)))
        else:
            return None
    else:
        lam = F.mul(F.sub(y2, y1), F.inv(F.sub(x2, x1)))
    x3 = F.sub(F.sub(F.mul(lam, lam), x1), x2)
    y3 = F.sub(F.mul(lam, F.sub(x1, x3)), y1)
    return (x3, y3)

def _ec_neg(F, p):
    return None if p is None else (p[0], F.neg(p[1]))

def _ec_mul(F, p, k):
    """Naive double-and-add (the model's ground truth for scalar multiplication)."""
    if k < 0:
        return _ec_mul(F, _ec_neg(F, p), -k)
    acc = None
    while k:
        if k & 1:
            acc = _ec_add(F, acc, p)
        p = _ec_add(F, p, p)
        k >>= 1
    return acc

def g1_add(p, q): return _ec_add(_FpOps, p, q)
def g1_neg(p): return _ec_neg(_FpOps, p)
def g1_mul(p, k): return _ec_mul(_FpOps, p, k)
def g2_add(p, q): return _ec_add(_Fp2Ops, p, q)
def g2_neg(p): return _ec_neg(_Fp2Ops, p)
def g2_mul(p, k): return _ec_mul(_Fp2Ops, p, k)

def g1_on_curve(p):
    return p is None or (p[1] * p[1] - p[0] ** 3 - 4) % P == 0

B_TWIST = (4, 4)  # 4(1+u)
def g2_on_curve(p):
    if p is None: return True
    x, y = p
    return f2_sub(f2_sqr(y), f2_add(f2_mul(f2_sqr(x), x), B_TWIST)) == F2_ZERO


# ----------------------------------------------------------------------------
# ark-ec VariableBaseMSM restatement
# ----------------------------------------------------------------------------
def ark_log2(x):
    if x == 0: return 0
    if x & (x - 1) == 0: return x.bit_length() - 1
    return x.bit_length()

def ark_ln_without_floats(a):
    return ark_log2(a) * 69 // 100

def ark_window_c(size):
    return 3 if size < 32 else ark_ln_without_floats(size) + 2

def ark_make_digits(a, w, num_bits=255):
    """Signed radix-2^w recoding of a canonical scalar `a` (A.2)."""
    limbs = [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
    radix = 1 << w
    mask = radix - 1
    carry = 0
    digits_count = (num_bits + w - 1) // w
    out = []
    for i in range(digits_count):
        bit_offset = i * w
        u64_idx = bit_offset // 64
        bit_idx = bit_offset % 64
        if bit_idx < 64 - w or u64_idx == 3:
            bit_buf = limbs[u64_idx] >> bit_idx
        else:
            bit_buf = ((limbs[u64_idx] >> bit_idx) | (limbs[u64_idx + 1] << (64 - bit_idx))) & 0xFFFFFFFFFFFFFFFF
        coef = carry + (bit_buf & mask)
        carry = (coef + radix // 2) >> w
        out.append(coef - (carry << w))
    out[-1] += carry << w
    return out

def ark_msm(F, bases, scalars):
    """msm_bigint_wnaf with arkworks' window rule; bases affine-or-None, scalars canonical ints.
    Truncates to min(len) like the reference relies on (legogroth16/src/prover.rs:286)."""
    size = min(len(bases), len(scalars))
    bases, scalars = bases[:size], scalars[:size]
    if size == 0:
        return None
    c = ark_window_c(size)
    digits = [ark_make_digits(s, c) for s in scalars]
    nw = (255 + c - 1) // c
    window_sums = []
    for w in range(nw):
        buckets = [None] * (1 << c)
        for d, b in zip(digits, bases):
            k = d[w]
            if k > 0:
                buckets[k - 1] = _ec_add(F, buckets[k - 1], b)
            elif k < 0:
                buckets[-k - 1] = _ec_add(F, buckets[-k - 1], _ec_neg(F, b))
        running = None
        res = None
        for b in reversed(buckets):
            running = _ec_add(F, running, b)
            res = _ec_add(F, res, running)
        window_sums.append(res)
    total = None
    for s in reversed(window_sums[1:]):
        total = _ec_add(F, total, s)
        for _ in range(c):
            total = _ec_add(F, total, total)
    return _ec_add(F, window_sums[0], total)

def g1_msm(bases, scalars): return ark_msm(_FpOps, bases, scalars)
def g2_msm(bases, scalars): return ark_msm(_Fp2Ops, bases, scalars)

def naive_msm(F, bases, scalars):
    acc = None
    for b, s in zip(bases, scalars):
        acc = _ec_add(F, acc, _ec_mul(F, b, s))
    return acc


# ----------------------------------------------------------------------------
# BLS12 Miller loop + final exponentiation
# ----------------------------------------------------------------------------
TWO_INV = fp_inv(2)

def _dbl_step(Rp):
    X, Y, Z = Rp
    a = f2_mul_fp(f2_mul(X, Y), TWO_INV)
    b = f2_sqr(Y)
    c = f2_sqr(Z)
    e = f2_mul(B_TWIST, f2_add(f2_add(c, c), c))
    f = f2_add(f2_add(e, e), e)
    g = f2_mul_fp(f2_add(b, f), TWO_INV)
    h = f2_sub(f2_sqr(f2_add(Y, Z)), f2_add(b, c))
    i = f2_sub(e, b)
    j = f2_sqr(X)
    e2 = f2_sqr(e)
    X3 = f2_mul(a, f2_sub(b, f))
    Y3 = f2_sub(f2_sqr(g), f2_add(f2_add(e2, e2), e2))
    Z3 = f2_mul(b, h)
    return (X3, Y3, Z3), (i, f2_add(f2_add(j, j), j), f2_neg(h))

def _add_step(Rp, Q):
    X, Y, Z = Rp
    qx, qy = Q
    theta = f2_sub(Y, f2_mul(qy, Z))
    lam = f2_sub(X, f2_mul(qx, Z))
    c = f2_sqr(theta)
    d = f2_sqr(lam)
    e = f2_mul(lam, d)
    f = f2_mul(Z, c)
    g = f2_mul(X, d)
    h = f2_sub(f2_add(e, f), f2_add(g, g))
    X3 = f2_mul(lam, h)
    Y3 = f2_sub(f2_mul(theta, f2_sub(g, h)), f2_mul(e, Y))
    Z3 = f2_mul(Z, e)
    j = f2_sub(f2_mul(theta, qx), f2_mul(lam, qy))
    return (X3, Y3, Z3), (j, f2_neg(theta), lam)

def x_bits_be_skip_first():
    bits = bin(X_ABS)[2:]
    return [int(b) for b in bits[1:]]

def g2_prepare(Q):
    """G2Prepared::from — 68 (c0,c1,c2) line-coefficient triples, or None for identity."""
    if Q is None:
        return None
    Rp = (Q[0], Q[1], F2_ONE)
    coeffs = []
    for bit in x_bits_be_skip_first():
        Rp, co = _dbl_step(Rp)
        coeffs.append(co)
        if bit:
            Rp, co = _add_step(Rp, Q)
            coeffs.append(co)
    return coeffs

def _ell(f, co, Pa):
    c0, c1, c2 = co
    return f12_mul_by_014(f, c0, f2_mul_fp(c1, Pa[0]), f2_mul_fp(c2, Pa[1]))

def multi_miller_loop(ps, qs):
    """Raw MillerLoopOutput (Fp12) — product over pairs, identity pairs skipped, conj at the end."""
    if len(ps) != len(qs):
        raise ValueError("zip_eq: length mismatch")
    pairs = [(p, g2_prepare(q)) for p, q in zip(ps, qs) if p is not None and q is not None]
    f = F12_ONE
    idx = 0
    for bit in x_bits_be_skip_first():
        f = f12_sqr(f)
        for p, co in pairs:
            f = _ell(f, co[idx], p)
        idx += 1
        if bit:
            for p, co in pairs:
                f = _ell(f, co[idx], p)
            idx += 1
    if X_IS_NEG:
        f = f12_conj(f)
    return f

def _exp_by_x(g):
    r = f12_pow(g, X_ABS)
    return f12_conj(r) if X_IS_NEG else r  # cyclotomic subgroup: conj == inverse

def final_exponentiation(f):
    """ark-ec Bls12::final_exponentiation chain (A.4). Returns None where arkworks returns None."""
    if f12_is_zero(f):
        return None
    f1 = f12_conj(f)
    f2_ = f12_inv(f)
    r = f12_mul(f1, f2_)
    f2_ = r
    r = f12_mul(f12_frob(r, 2), f2_)
    y0 = f12_sqr(r)
    y1 = _exp_by_x(r)
    y2 = f12_conj(r)
    y1 = f12_mul(y1, y2)
    y2 = _exp_by_x(y1)
    y1 = f12_conj(y1)
    y1 = f12_mul(y1, y2)
    y2 = _exp_by_x(y1)
    y1 = f12_frob(y1, 1)
    y1 = f12_mul(y1, y2)
    r = f12_mul(r, y0)
    y0 = _exp_by_x(y1)
    y2 = _exp_by_x(y0)
    y0 = f12_frob(y1, 2)
    y1 = f12_conj(y1)
    y1 = f12_mul(y1, y2)
    y1 = f12_mul(y1, y0)
    r = f12_mul(r, y1)
    return r

def pairing(p, q):
    return final_exponentiation(multi_miller_loop([p], [q]))


# ----------------------------------------------------------------------------
# Deterministic PRNG shared with the C oracle / bench (SplitMix64)
# ----------------------------------------------------------------------------
class SplitMix64:
    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF
    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)
    def scalar(self):
        """Uniform in [0, r) by rejection sampling of 255-bit draws."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 255) - 1
            if v < R:
                return v


# ----------------------------------------------------------------------------
# limb (de)serialisation helpers for the C ABI (Montgomery, little-endian u64)
# ----------------------------------------------------------------------------
def fp_to_mont_limbs(a):
    v = (a * FP_R) % P
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]

def fp_from_mont_limbs(l):
    v = sum(int(x) << (64 * i) for i, x in enumerate(l))
    return (v * fp_inv(FP_R)) % P

def fr_to_limbs(a):
    return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]
