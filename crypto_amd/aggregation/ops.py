"""Group / GT / serialization helpers shared by the aggregation modules (thin wrappers over the C ABI)."""
import ctypes as C
import numpy as np
from .._native import lib, DockGpuError
import importlib
M = importlib.import_module(__package__.rsplit(".", 1)[0] + ".msm")
from .. import pairing, serde
from ..pairing_check import fp12_mul, fp12_pow, fp12_one   # noqa: F401

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
_FP_RINV = pow(1 << 384, -1, P_MOD)
G1, G2 = M.G1, M.G2


def inv(a):
    return pow(a % R_MOD, -1, R_MOD)          # extended Euclid: ~20x faster than the Fermat power for a 255-bit modulus


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def limbs(vals):
    """ints -> (n, 4) uint64 canonical limbs"""
    return np.array([[(int(v) % R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64).reshape(-1, 4)


def pts(curve, a):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, curve.AW)


def mul_add(curve, points, scalars, addend=None):
    """[addend_i + s_i * P_i] affine; `scalars`: one int (same for all) or a list of ints"""
    M._ensure()
    P = pts(curve, points)
    n = len(P)
    out = np.zeros((n, curve.AW), dtype=np.uint64)
    if n == 0:
        return out
    inf = np.zeros(n, dtype=np.uint8)
    if isinstance(scalars, int):
        sc, stride = limbs([scalars]), 0
    else:
        sc, stride = limbs(scalars), 4
        if len(sc) != n:
            raise ValueError("scalar count")
    A = None if addend is None else pts(curve, addend)
    fn = lib().dgpu_g1_mul_add_batch if curve is G1 else lib().dgpu_g2_mul_add_batch
    rc = fn(_p(P), None, _p(sc), stride, _p(A), None, n, _p(out), _p(inf))
    if rc:
        raise DockGpuError(rc, "dgpu_mul_add_batch")
    return out


def msm(curve, points, scalars):
    """sum s_i P_i as an affine ABI point (identity: zero words).  Up to 16 terms (the `mul_bigint`s and two-term combinations of the KZG
    checks and of the final verification, CPU scalar multiplications in the reference): host arithmetic in the library (dgpu_lincomb_*);
    more: the MSM entry point."""
    P = pts(curve, points)
    if 0 < len(P) <= 16:
        sc = np.ascontiguousarray(limbs(scalars))
        inf = np.ascontiguousarray((~P.any(axis=1)).astype(np.uint8))
        jac = np.zeros(curve.JW, dtype=np.uint64)
        fn = lib().dgpu_lincomb_g1 if curve is G1 else lib().dgpu_lincomb_g2
        rc = fn(_p(np.ascontiguousarray(P)), _p(inf), _p(sc), len(P), _p(jac))
        if rc:
            raise DockGpuError(rc, "dgpu_lincomb")
        return np.zeros(curve.AW, dtype=np.uint64) if not jac[curve.AW:].any() else jac[:curve.AW].copy()
    jac = M.msm_bigint(curve, P, limbs(scalars))
    return np.zeros(curve.AW, dtype=np.uint64) if not jac[curve.AW:].any() else jac[:curve.AW].copy()


def neg(curve, pt):
    pt = np.array(pt, dtype=np.uint64).reshape(curve.AW)
    if not pt.any():
        return pt
    h = curve.AW // 2
    for k in range(h // 6):
        y = sum(int(x) << (64 * i) for i, x in enumerate(pt[h + 6 * k:h + 6 * k + 6]))
        y = (P_MOD - y) % P_MOD
        pt[h + 6 * k:h + 6 * k + 6] = [(y >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]
    return pt


def multi_pairing(ps, qs):
    return pairing.multi_pairing(pts(G1, ps), pts(G2, qs))


def multi_pairings(jobs):
    """[(ps, qs), ...] -> [GT, ...]: the Miller loops in one launch sequence, the final exponentiations on the library's host threads"""
    return pairing.multi_pairings([(pts(G1, a), pts(G2, b)) for a, b in jobs])


# ---- serialize_compressed of the transcript elements -----------------------------------------------------------------
def fr_bytes(v):
    return (int(v) % R_MOD).to_bytes(32, "little")


def g1_bytes(pt):
    return bytes(serde.serialize(G1, pts(G1, pt)))


def g2_bytes(pt):
    return bytes(serde.serialize(G2, pts(G2, pt)))


def gt_bytes(f):
    """PairingOutput / Fp12: twelve Fp, c0.c0.c0 first, each 48 bytes little-endian canonical"""
    f = np.asarray(f, dtype=np.uint64).reshape(12, 6)
    out = bytearray()
    for row in f:
        v = sum(int(x) << (64 * i) for i, x in enumerate(row)) * _FP_RINV % P_MOD
        out += v.to_bytes(48, "little")
    return bytes(out)


def gt_in_subgroup(f):
    """PairingOutput's `Valid::check` (ark-ec): the element has order dividing r (dgpu_gt_in_subgroup: a Frobenius identity and f^p == f^x on the host,
    ~0.1 ms per element instead of the 1.3 ms of f^r)"""
    a = np.ascontiguousarray(np.asarray(f, dtype=np.uint64).reshape(-1, 72))
    ok = np.zeros(len(a), dtype=np.uint8)
    rc = lib().dgpu_gt_in_subgroup(_p(a), len(a), _p(ok))
    if rc:
        raise DockGpuError(rc, "dgpu_gt_in_subgroup")
    return bool(ok.all())


def gt_multi_pow(bases, exps):
    """prod bases[i]^exps[i] in GT (host threads)"""
    n = len(bases)
    a = np.ascontiguousarray(np.stack([np.asarray(b, dtype=np.uint64).reshape(72) for b in bases])) if n else np.zeros((0, 72), np.uint64)
    e = limbs(exps)
    out = np.zeros(72, dtype=np.uint64)
    rc = lib().dgpu_fp12_multi_pow(_p(a), _p(e), n, _p(out))
    if rc:
        raise DockGpuError(rc, "dgpu_fp12_multi_pow")
    return out


_POOL = None
_HOST_POOL = None


def parallel(thunks, host=False):
    """run independent ABI calls from host threads (the library keeps six calls in flight on separate HIP streams, further callers queue for a slot;
    ctypes drops the GIL) — the reference issues them one after another, each rayon-parallel inside.  host=True: calls that only use host cores (wider pool)."""
    global _POOL, _HOST_POOL
    from concurrent.futures import ThreadPoolExecutor
    if host:
        if _HOST_POOL is None:
            _HOST_POOL = ThreadPoolExecutor(8)
        pool = _HOST_POOL
    else:
        if _POOL is None:
            _POOL = ThreadPoolExecutor(8)
        pool = _POOL
    return [f.result() for f in [pool.submit(t) for t in thunks]]
