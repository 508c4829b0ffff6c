"""zkSNARK for linear subspaces (CP_link of LegoGroth16) over the C ABI — mirror of /root/reference/legogroth16/src/link/snark.rs
(`PESubspaceSnark::{keygen, prove, verify}` :87-162) and link/utils.rs (`SparseMatrix` :28-76, `sparse_vector_matrix_mult` :109-120,
`inner_product` :123-125, `scale_vector` :128-138, `multiples_of_g` :141-147).

The hot calls are the ones the reference makes: `prove` is one `msm_unchecked` (dgpu_msm_g1), `verify` one `multi_pairing` with prepared G2
operands (dgpu_g2_prepare / dgpu_multi_miller_loop_prepared + dgpu_final_exponentiation), `keygen` one small MSM per matrix column and one
fixed-base batch (dgpu_fixed_base_g2).  Points are ABI-layout numpy arrays (identity = all-zero words), scalars Python ints."""
import numpy as np
from . import pairing
from .msm import G1, G2, msm_bigint
from .fixed_base import multiply_field_elems_with_same_group_elem

R_MOD = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
P_MOD = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB


class LinkError(ValueError):
    pass


def _sc(v):
    v %= R_MOD
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _affine(curve, jac):
    jac = np.asarray(jac, dtype=np.uint64)
    return np.zeros(curve.AW, dtype=np.uint64) if not jac[curve.AW:].any() else jac[:curve.AW].copy()


def _neg_g2(pt):
    pt = np.array(pt, dtype=np.uint64)
    if not pt.any():
        return pt
    for k in range(2):
        y = sum(int(x) << (64 * i) for i, x in enumerate(pt[12 + 6 * k:18 + 6 * k]))
        y = (P_MOD - y) % P_MOD
        pt[12 + 6 * k:18 + 6 * k] = [(y >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]
    return pt


class SparseMatrix:
    """Column-major sparse matrix of G1 points (link/utils.rs:28-76)"""

    def __init__(self, nr, nc):
        self.nr, self.nc = nr, nc
        self.cols = [[] for _ in range(nc)]

    def insert_val(self, r, c, v):
        if len(self.cols) <= c:
            raise LinkError("InvalidIndex(%d, %d)" % (c, len(self.cols)))
        self.cols[c].append((r, np.asarray(v, dtype=np.uint64).reshape(12)))

    def insert_row_slice(self, r, c_offset, vs):
        for i, x in enumerate(vs):
            self.insert_val(r, c_offset + i, x)


def sparse_vector_matrix_mult(v, m):
    """m^T . v: for every column, sum_i v[pos_i] * val_i (utils.rs:85-120; the reference notes "can be optimized using MSM": it is one here)"""
    out = []
    for col in m.cols:
        if not col:
            out.append(np.zeros(12, dtype=np.uint64)); continue
        for r, _ in col:
            if len(v) <= r:
                raise LinkError("InvalidIndex(%d, %d)" % (r, len(v)))
        pts = np.stack([p for _, p in col]); sc = np.stack([_sc(v[r]) for r, _ in col])
        out.append(_affine(G1, msm_bigint(G1, pts, sc)))
    return np.stack(out)


def inner_product(a, b):
    """utils.rs:123-125: msm_unchecked(b, a).into_affine()"""
    b = np.ascontiguousarray(b, dtype=np.uint64).reshape(-1, 12)
    n = min(len(a), len(b))
    return _affine(G1, msm_bigint(G1, b[:n], np.stack([_sc(x) for x in a[:n]]) if n else np.zeros((0, 4), np.uint64)))


class PP:
    def __init__(self, l, t, g1, g2):
        self.l, self.t, self.g1, self.g2 = l, t, np.asarray(g1, dtype=np.uint64), np.asarray(g2, dtype=np.uint64)


def keygen(pp, m, k, a):
    """snark.rs:101-123 with the trapdoor `k` (pp.l scalars) and `a` passed in (the reference draws them from `rng`)"""
    if len(k) != pp.l:
        raise LinkError("trapdoor length")
    p = sparse_vector_matrix_mult(k, m)
    c = [(a * x) % R_MOD for x in k]                                   # scale_vector
    cg, _ = multiply_field_elems_with_same_group_elem(G2, pp.g2, c + [a])      # multiples_of_g(&pp.g2, &c) and pp.g2.mul(a)
    return {"p": p}, {"c": cg[:len(c)], "a": cg[len(c)]}


def prove(pp, ek, w):
    """snark.rs:125-130"""
    if pp.t < len(w):
        raise LinkError("VectorLongerThanExpected(%d, %d)" % (len(w), pp.t))
    return inner_product(w, ek["p"])


def verify(pp, vk, x, pi):
    """snark.rs:132-161: e(x_0, c_0) ... e(x_{l-1}, c_{l-1}) e(pi, -a) == 1, G2 operands prepared"""
    x = np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 12)
    if pp.l != len(x):
        raise LinkError("VectorWithUnexpectedLength(%d, %d)" % (len(x), pp.l))
    if len(vk["c"]) < len(x):
        raise LinkError("VectorLongerThanExpected(%d, %d)" % (len(x), len(vk["c"])))
    a = np.concatenate([x, np.asarray(pi, dtype=np.uint64).reshape(1, 12)])
    b = pairing.G2Prepared.from_affine(np.concatenate([vk["c"][:len(x)], _neg_g2(vk["a"]).reshape(1, 24)]))
    gt = pairing.multi_pairing(a, b)
    one = np.zeros(72, dtype=np.uint64); one[:6] = pairing.FP_ONE_MONT
    if gt is None or not (gt == one).all():
        raise LinkError("InvalidProof")
