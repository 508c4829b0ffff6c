"""GPU (-m gpu): the folding step with its doubling chains done ahead of the scalar — dgpu_g*_fold_prepare / dgpu_g*_fold_apply
(crypto_amd/csrc/fold_kernels.hip.h) — against dgpu_g*_mul_add_batch (the chain kernels, themselves checked against the oracle in
tests/test_gpu_aggregation.py::test_mul_add_batch_matches_oracle) and against the CPU oracle directly: out_i = A_i + c P_i for one scalar c.
What the aggregation's `compress` / Key::compress compute (/root/reference/legogroth16/src/aggregation/utils.rs:34-49, key.rs:160-184).
Scalars: 0, 1, r - 1, the borders of the GLV / GLS digits, values at and above r; points: identity points and addends, an addend equal to the
product (the doubling branch of the last addition) and to its negative (the identity comes out).  Bar: bit-exact."""
import ctypes as C
import numpy as np
import pytest
import torch
import oracle_c as O
import util as U
import crypto_amd as ca
from crypto_amd._native import lib
from crypto_amd.aggregation import ops

pytestmark = pytest.mark.gpu
R = U.R
X = 0xD201000000010000
p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module", autouse=True)
def _device():
    assert torch.cuda.is_available()
    ca.init(0)


def limbs(v):
    return np.array([(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)


def fold(curve, P, c, A):
    """prepare once, apply: (points, flags)"""
    g = "g1" if curve is ca.G1 else "g2"
    P = np.ascontiguousarray(P, dtype=np.uint64); n = len(P)
    h = C.c_uint64(0)
    assert getattr(lib(), "dgpu_%s_fold_prepare" % g)(p(P), n, C.byref(h)) == 0
    out = np.zeros_like(P); inf = np.zeros(n, np.uint8)
    try:
        A = None if A is None else np.ascontiguousarray(A, dtype=np.uint64)
        assert getattr(lib(), "dgpu_%s_fold_apply" % g)(h.value, p(limbs(c)), p(A), p(out), p(inf)) == 0
    finally:
        assert lib().dgpu_fold_free(h.value) == 0
    return out, inf


SCALARS = [0, 1, 2, 3, R - 1, R - 2, X - 1, X, X + 1, X * X - 2, X * X - 1, X * X, X * X + 1, X ** 3 - 1, X ** 3, X ** 3 + X, 2 ** 64 - 1, 2 ** 64, 2 ** 127, 2 ** 128 - 1, 2 ** 128,
           2 ** 191, 2 ** 254 + 12345, R - X, R - X * X, (X ** 3) * (R // X ** 3), 0xDEADBEEF, int.from_bytes(b"\x5a" * 31, "little")]


@pytest.mark.parametrize("curve", ["g1", "g2"])
def test_fold_equals_mul_add_batch_and_oracle(curve):
    grp, cv = (O.G1, ca.G1) if curve == "g1" else (O.G2, ca.G2)
    n = 37
    ks = O.rand_scalars(31, n); ad = O.rand_scalars(33, n)
    P = np.stack([grp.to_affine(grp.mul(grp.generator(), k))[0] for k in ks])
    A = np.stack([grp.to_affine(grp.mul(grp.generator(), k))[0] for k in ad])
    A[3] = 0; P[4] = 0; A[7] = 0; P[7] = 0                           # identity addend, identity point, both
    rnd = [O.limbs_to_int(s) for s in O.rand_scalars(34, 6)]
    for c in SCALARS + rnd:
        Ac = A.copy()
        cl = O.int_to_limbs(c % R, 4)
        Ac[5] = ops.neg(cv, grp.to_affine(grp.mul(P[5], cl))[0]) if c % R else 0        # A = -(c P): the identity comes out
        prod6, inf6 = grp.to_affine(grp.mul(P[6], cl))
        Ac[6] = 0 if inf6 else prod6                                                    # A = c P: the last addition doubles
        got, ginf = fold(cv, P, c, Ac)
        want = ops.mul_add(cv, P, c, Ac)
        assert (got == want).all(), hex(c)
        assert (ginf == (~got.any(axis=1)).astype(np.uint8)).all(), hex(c)
        for i in (0, 3, 4, 5, 6, 7, n - 1):                                             # ... and the oracle on a sample of rows
            e = grp.mul(P[i], cl, inf=not P[i].any())
            if Ac[i].any():
                e = grp.add(e, grp.mul(Ac[i], O.int_to_limbs(1, 4)))
            ea, einf = grp.to_affine(e)
            assert (got[i] == (np.zeros_like(ea) if einf else ea)).all(), (hex(c), i)
    # no addend: plain scaling; a scalar at or above r straight through the ABI (reduced before it is split)
    got, _ = fold(cv, P, rnd[0], None)
    assert (got == ops.mul_add(cv, P, rnd[0])).all()
    for v in (R, R + 5, 2 ** 256 - 1):
        got, _ = fold(cv, P, v, A)
        assert (got == ops.mul_add(cv, P, v % R, A)).all(), hex(v)


@pytest.mark.parametrize("curve,n", [("g1", 1), ("g1", 2), ("g1", 700), ("g2", 1), ("g2", 300)])
def test_fold_sizes_and_reuse(curve, n):
    """one table, several scalars (a handle serves any number of applies); the pool of kept tables is reused by the next prepare"""
    grp, cv = (O.G1, ca.G1) if curve == "g1" else (O.G2, ca.G2)
    g = curve
    P = U.seq_bases(grp, n, 4100 + n, threads=16)[0]
    A = U.seq_bases(grp, n, 4200 + n, threads=16)[0]
    h = C.c_uint64(0)
    assert getattr(lib(), "dgpu_%s_fold_prepare" % g)(p(P), n, C.byref(h)) == 0
    for c in (5, R - 7, O.limbs_to_int(O.rand_scalars(77, 1)[0])):
        out = np.zeros_like(P); inf = np.zeros(n, np.uint8)
        assert getattr(lib(), "dgpu_%s_fold_apply" % g)(h.value, p(limbs(c)), p(A), p(out), p(inf)) == 0
        assert (out == ops.mul_add(cv, P, c, A)).all()
    assert lib().dgpu_fold_free(h.value) == 0
    h2 = C.c_uint64(0)
    half = max(1, n // 2)
    for _ in range(8):                                                    # (every slot's staging buffer has grown after a pass over the six slots)
        assert getattr(lib(), "dgpu_%s_fold_prepare" % g)(p(P[:half]), half, C.byref(h2)) == 0 and lib().dgpu_fold_free(h2.value) == 0
    a0 = ca.device_alloc_count()
    for _ in range(8):
        assert getattr(lib(), "dgpu_%s_fold_prepare" % g)(p(P[:half]), half, C.byref(h2)) == 0       # the halved vector of the next round ...
        assert lib().dgpu_fold_free(h2.value) == 0
    assert ca.device_alloc_count() == a0                                  # ... lives in a buffer an earlier table left behind
    # argument checks
    assert getattr(lib(), "dgpu_%s_fold_apply" % g)(h.value, p(limbs(1)), None, p(out), p(inf)) == -3              # a freed handle
    assert getattr(lib(), "dgpu_%s_fold_prepare" % g)(None, 3, C.byref(h2)) == -3
    other = "g2" if g == "g1" else "g1"
    assert getattr(lib(), "dgpu_%s_fold_prepare" % g)(p(P), n, C.byref(h)) == 0
    assert getattr(lib(), "dgpu_%s_fold_apply" % other)(h.value, p(limbs(1)), None, p(out), p(inf)) == -3          # the other curve's entry point
    assert lib().dgpu_bases_free(h.value) == -3                                                                    # not a bases handle
    assert lib().dgpu_fold_free(h.value) == 0
