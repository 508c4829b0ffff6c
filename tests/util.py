"""Shared helpers for the tests: int <-> ABI limb conversion and fixture loading.
The oracle (oracle/) is test infrastructure; nothing in crypto_amd/ imports it."""
import json
import os
import numpy as np
import oracle_c as O
import bls12_381_model as M

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
P, R = M.P, M.R
_RI = pow(M.FP_R, -1, P)


def load(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def fp_abi(v):
    """canonical int -> 6 u64 Montgomery limbs (ark-ff layout)"""
    return O.int_to_limbs(v * M.FP_R % P, 6)


def fp_int(l):
    return O.limbs_to_int(l) * _RI % P


def g1_abi(pt):
    """model point (x, y) or None -> (12 u64, is_inf)"""
    if pt is None:
        return np.zeros(12, np.uint64), 1
    return np.concatenate([fp_abi(pt[0]), fp_abi(pt[1])]), 0


def g2_abi(pt):
    if pt is None:
        return np.zeros(24, np.uint64), 1
    return np.concatenate([fp_abi(pt[0][0]), fp_abi(pt[0][1]), fp_abi(pt[1][0]), fp_abi(pt[1][1])]), 0


def dec_g1(e):
    return None if e is None else (int(e[0], 16), int(e[1], 16))


def dec_g2(e):
    return None if e is None else ((int(e[0][0], 16), int(e[0][1], 16)), (int(e[1][0], 16), int(e[1][1], 16)))


def jac_to_model(G, jac):
    """Jacobian limbs -> model affine point (ints) or None, via the oracle's to_affine"""
    a, inf = G.to_affine(np.ascontiguousarray(jac, dtype=np.uint64))
    if inf:
        return None
    v = [fp_int(a[6 * i:6 * i + 6]) for i in range(G.AW // 6)]
    return (v[0], v[1]) if G.AW == 12 else ((v[0], v[1]), (v[2], v[3]))


def case_arrays(case):
    """fixture MSM case -> (bases ABI array, is_inf, scalars array, expected model point)"""
    g2 = case["group"] == "G2"
    enc, dec = (g2_abi, dec_g2) if g2 else (g1_abi, dec_g1)
    pts = [enc(dec(b)) for b in case["bases"]]
    w = 24 if g2 else 12
    bases = np.stack([p[0] for p in pts]) if pts else np.zeros((0, w), np.uint64)
    inf = np.array([p[1] for p in pts], dtype=np.uint8)
    sc = np.stack([O.int_to_limbs(int(s, 16), 4) for s in case["scalars"]]) if case["scalars"] else np.zeros((0, 4), np.uint64)
    return bases, inf, sc, dec(case["expected"])


def f12_abi(vals):
    return np.concatenate([fp_abi(int(v, 16) if isinstance(v, str) else v) for v in vals])


def f12_ints(limbs):
    return [fp_int(limbs[6 * i:6 * i + 6]) for i in range(12)]


def seq_bases(G, n, seed, threads=8):
    """P_i = (k0 + i d) G with known dlogs; returns (bases, k0, d)"""
    k0 = O.rand_scalars(seed, 1)[0]
    d = O.rand_scalars(seed + 1, 1)[0]
    return G.gen_seq(k0, d, n, threads=threads), O.limbs_to_int(k0), O.limbs_to_int(d)


def closed_form(G, scalars, k0, d):
    """(sum s_i (k0 + i d)) * generator as a model point, computed by the oracle's double-and-add"""
    sv = [O.limbs_to_int(x) for x in scalars]
    tot = (sum(sv) * k0 + sum(i * s for i, s in enumerate(sv)) * d) % R
    return jac_to_model(G, G.mul(G.generator(), O.int_to_limbs(tot, 4)))


import contextlib


@contextlib.contextmanager
def bucket_pipeline():
    """Inside the block MSMs of up to 8192 terms on plain bases run the Pippenger bucket pipeline (sort, k_accumulate, fix-up, reduction) instead of
    the bucket-free tree path that serves them since round 4 (dgpu_set_small_msm_max): tests that are ABOUT windows, chunks and buckets use it."""
    from crypto_amd._native import lib
    assert lib().dgpu_set_small_msm_max(0) == 0
    try:
        yield
    finally:
        lib().dgpu_set_small_msm_max(8192)


def on_both_paths(fn):
    """fn() on the tree path (the default for n <= 8192) and on the bucket pipeline; returns both results"""
    a = fn()
    with bucket_pipeline():
        b = fn()
    return a, b
