"""ctypes binding of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
All point/field arrays are numpy uint64 in the C-ABI layout (Montgomery limbs, little-endian).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    if force or not os.path.exists(_SO) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_SO)
        for f in ("oracle.c", "fields.h", "ec_tmpl.inc")
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_window_c.restype = C.c_int
        _lib.orc_window_c.argtypes = [C.c_size_t]
        for name in ("orc_g1_to_affine", "orc_g2_to_affine", "orc_g1_on_curve", "orc_g2_on_curve",
                     "orc_final_exponentiation"):
            getattr(_lib, name).restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def u64(shape):
    return np.zeros(shape, dtype=np.uint64)


def window_c(n):
    return lib().orc_window_c(n)


def make_digits(scalar_limbs, c):
    nw = (255 + c - 1) // c
    out = np.zeros(nw, dtype=np.int64)
    s = np.ascontiguousarray(scalar_limbs, dtype=np.uint64)
    lib().orc_make_digits(_p(s), C.c_int(c), _p(out))
    return out


def rand_scalars(seed, n):
    out = u64((n, 4))
    lib().orc_rand_scalars(C.c_uint64(seed), C.c_size_t(n), _p(out))
    return out


def fr_from_mont(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fr_from_mont(_p(a), _p(out), C.c_size_t(a.size // 4))
    return out


def fr_to_mont(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fr_to_mont(_p(a), _p(out), C.c_size_t(a.size // 4))
    return out


def fp_to_mont(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fp_to_mont(_p(a), _p(out), C.c_size_t(a.size // 6))
    return out


def fp_from_mont(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fp_from_mont(_p(a), _p(out), C.c_size_t(a.size // 6))
    return out


class _Group:
    def __init__(self, tag, aff_words):
        self.tag = tag
        self.AW = aff_words          # 12 (G1) / 24 (G2)
        self.JW = aff_words * 3 // 2  # 18 / 36

    def _f(self, name):
        return getattr(lib(), "orc_%s_%s" % (self.tag, name))

    def generator(self):
        out = u64(self.AW)
        self._f("generator")(_p(out))
        return out

    def msm(self, bases, scalars, is_inf=None, threads=1):
        """msm_bigint over min(len) pairs; returns Jacobian limbs (JW u64)."""
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, self.AW)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        n = min(len(bases), len(scalars))
        inf = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
        out = u64(self.JW)
        self._f("msm")(_p(bases), _p(inf), _p(scalars), C.c_size_t(n), C.c_int(threads), _p(out))
        return out

    def to_affine(self, jac):
        jac = np.ascontiguousarray(jac, dtype=np.uint64)
        out = u64(self.AW)
        inf = self._f("to_affine")(_p(jac), _p(out))
        return out, bool(inf)

    def mul(self, base, k_limbs, inf=False):
        base = np.ascontiguousarray(base, dtype=np.uint64)
        k = np.ascontiguousarray(k_limbs, dtype=np.uint64)
        out = u64(self.JW)
        self._f("mul")(_p(base), C.c_int(int(inf)), _p(k), _p(out))
        return out

    def add(self, a, b):
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = u64(self.JW)
        self._f("add")(_p(a), _p(b), _p(out))
        return out

    def on_curve(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.uint64)
        return bool(self._f("on_curve")(_p(xy)))

    def gen_seq(self, k0_limbs, d_limbs, n, threads=8):
        """P_i = (k0 + i*d)*G, i < n; affine Montgomery limbs (n, AW)."""
        out = u64((n, self.AW))
        k0 = np.ascontiguousarray(k0_limbs, dtype=np.uint64)
        d = np.ascontiguousarray(d_limbs, dtype=np.uint64)
        self._f("gen_seq")(_p(k0), _p(d), C.c_size_t(n), C.c_int(threads), _p(out))
        return out


G1 = _Group("g1", 12)
G2 = _Group("g2", 24)


def g2_prepare(q):
    q = np.ascontiguousarray(q, dtype=np.uint64)
    out = u64((68, 3, 12))
    lib().orc_g2_prepare(_p(q), _p(out))
    return out


def multi_miller_loop(p, q, skip=None, threads=1):
    p = np.ascontiguousarray(p, dtype=np.uint64).reshape(-1, 12)
    q = np.ascontiguousarray(q, dtype=np.uint64).reshape(-1, 24)
    assert len(p) == len(q), "zip_eq: length mismatch"
    sk = None if skip is None else np.ascontiguousarray(skip, dtype=np.uint8)
    out = u64(72)
    lib().orc_multi_miller_loop(_p(p), _p(q), _p(sk), C.c_size_t(len(p)), C.c_int(threads), _p(out))
    return out


def multi_miller_loop_mixed(p_aff, q_aff, p_prep, coeffs, threads=1):
    """the verifier's call (verifier.rs:69-76): affine (P, Q) pairs + pairs whose G2 member is already `G2Prepared` (g2_prepare output)"""
    p_aff = np.ascontiguousarray(p_aff, dtype=np.uint64).reshape(-1, 12)
    q_aff = np.ascontiguousarray(q_aff, dtype=np.uint64).reshape(-1, 24)
    p_prep = np.ascontiguousarray(p_prep, dtype=np.uint64).reshape(-1, 12)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(len(p_prep), 68 * 36)
    assert len(p_aff) == len(q_aff)
    out = u64(72)
    lib().orc_multi_miller_loop_mixed(_p(p_aff), _p(q_aff), C.c_size_t(len(p_aff)), _p(p_prep), _p(coeffs), C.c_size_t(len(p_prep)), C.c_int(threads), _p(out))
    return out


def witness_map(mats, z, num_inputs, num_constraints, threads=1):
    """LibsnarkReduction::witness_map_from_matrices (r1cs_to_qap.rs:150-210).  mats: three CSR triples (rowptr u64, cols u32, vals (nnz, 4) canonical);
    z: (num_vars, 4) canonical.  Returns the D canonical coefficients of h.  threads > 1: the threaded form (same values)."""
    z = np.ascontiguousarray(z, dtype=np.uint64).reshape(-1, 4)
    D = 1
    while D < num_constraints + num_inputs:
        D *= 2
    out = u64((D, 4))
    args, keep = [], []
    for rp, cl, vl in mats:
        rp = np.ascontiguousarray(rp, dtype=np.uint64); cl = np.ascontiguousarray(cl, dtype=np.uint32); vl = np.ascontiguousarray(vl, dtype=np.uint64)
        keep += [rp, cl, vl]
        args += [_p(rp), _p(cl), _p(vl)]
    L = lib()
    if threads > 1:
        L.orc_witness_map_mt.restype = C.c_int
        L.orc_witness_map_mt(*args, _p(z), C.c_size_t(len(z)), C.c_size_t(num_inputs), C.c_size_t(num_constraints), C.c_int(threads), _p(out))
    else:
        L.orc_witness_map.restype = C.c_int
        L.orc_witness_map(*args, _p(z), C.c_size_t(len(z)), C.c_size_t(num_inputs), C.c_size_t(num_constraints), _p(out))
    return out


def g1_scale_batch(points, scalars, negate=None, is_inf=None, threads=1):
    """out_i = (+-) s_i P_i as affine points (scalars: (n, 4) or ONE (4,) scalar for every point) — the checker's scalings, one task per point"""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12); n = len(points)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    stride = 0 if scalars.size == 4 else 4
    neg = None if negate is None else np.ascontiguousarray(negate, dtype=np.uint8)
    inf = None if is_inf is None else np.ascontiguousarray(is_inf, dtype=np.uint8)
    out = u64((n, 12)); oinf = np.zeros(n, np.uint8)
    lib().orc_g1_scale_batch(_p(points), _p(inf), _p(scalars), C.c_size_t(stride), _p(neg), C.c_size_t(n), int(threads), _p(out), _p(oinf))
    return out, oinf


def g2_prepare_batch(qs, threads=1):
    """G2Prepared::from for every point: (n, 68 * 36) words"""
    qs = np.ascontiguousarray(qs, dtype=np.uint64).reshape(-1, 24); n = len(qs)
    out = u64((n, 68 * 36))
    lib().orc_g2_prepare_batch(_p(qs), C.c_size_t(n), int(threads), _p(out))
    return out


def fp12_multi_pow(bases, exps, threads=1):
    """prod_i bases_i ^ exps_i (exps: (n, 4) canonical limbs); bases (72,): the same base for every exponent"""
    bases = np.ascontiguousarray(bases, dtype=np.uint64); exps = np.ascontiguousarray(exps, dtype=np.uint64).reshape(-1, 4)
    stride = 0 if bases.size == 72 else 72
    out = u64(72)
    lib().orc_fp12_multi_pow(_p(bases), C.c_size_t(stride), _p(exps), C.c_size_t(len(exps)), int(threads), _p(out))
    return out


def final_exponentiation(f):
    f = np.ascontiguousarray(f, dtype=np.uint64)
    out = u64(72)
    rc = lib().orc_final_exponentiation(_p(f), _p(out))
    return None if rc != 0 else out


def fp12_mul(a, b):
    out = u64(72)
    lib().orc_fp12_mul(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    return out


def fp12_one():
    out = u64(72)
    lib().orc_fp12_one(_p(out))
    return out


def fp12_pow(a, e_int):
    nl = max(1, (e_int.bit_length() + 63) // 64)
    e = np.array([(e_int >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nl)], dtype=np.uint64)
    out = u64(72)
    lib().orc_fp12_pow(_p(np.ascontiguousarray(a)), _p(e), C.c_int(nl), _p(out))
    return out


# ---- int <-> limb helpers -------------------------------------------------
def int_to_limbs(v, n):
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def limbs_to_int(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(np.asarray(l).reshape(-1)))
