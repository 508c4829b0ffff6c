"""CPU: the DEVICE field / group code (crypto_amd/csrc/fp29.hip.h, fp2_29.hip.h, ec29.hip.h) compiled for the host
with the FP29_CHECK worst-case bound tracker, checked against the big-integer model.  Every assertion inside
the shim that fires would mean a lazy-limb overflow is possible for SOME input of the same classes, so a green
run here proves the carry-free arithmetic the kernels use is overflow-free, not just right on these inputs."""
import ctypes as C
import os
import random
import subprocess
import numpy as np
import pytest
import bls12_381_model as M
import util as U

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "native", "fp29_host_shim.cpp")
SO = os.path.join(HERE, "native", "libfp29_host_shim.so")
P = M.P


@pytest.fixture(scope="module")
def shim():
    deps = [SRC] + [os.path.join(HERE, "..", "crypto_amd", "csrc", f) for f in ("fp29.hip.h", "fp30s.hip.h", "fs2_pair.hip.h", "fp2_29.hip.h", "ec29.hip.h", "pairing29.hip.h", "fr29.hip.h", "fp_safegcd.hip.h")]
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", SO, SRC])
    return C.CDLL(SO)


def p_(a):
    return a.ctypes.data_as(C.c_void_p)


def test_fp_ops(shim):
    random.seed(1)
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, 2 ** 380, 2 ** 377 - 1] + [random.randrange(P) for _ in range(200)]
    out = np.zeros(6, np.uint64)
    for a in vals:
        A = U.fp_abi(a)
        shim.shim_fp_roundtrip(p_(A), p_(out)); assert U.fp_int(out) == a
        shim.shim_fp_sqr(p_(A), p_(out)); assert U.fp_int(out) == a * a % P
    for i in range(len(vals) - 2):
        a, b, c = vals[i], vals[i + 1], vals[i + 2]
        A, B, Cc = U.fp_abi(a), U.fp_abi(b), U.fp_abi(c)
        shim.shim_fp_mul(p_(A), p_(B), p_(out)); assert U.fp_int(out) == a * b % P
        shim.shim_fp_submul(p_(A), p_(B), p_(Cc), p_(out)); assert U.fp_int(out) == (a - b) * c % P
        assert ((shim.shim_fp_is_zero(p_(A), p_(B)) & 2) != 0) == (a == b)
        assert shim.shim_fp_is_zero(p_(A), p_(A)) == 3


def test_fs_ops(shim):
    """fp30s.hip.h (13 signed 30-bit limbs, the field of the G1 MSM kernels) against big integers: products, squares, the fused a b - c d, lazy
    subtractions with both carry passes, negation, the zero tests, and the conversions to / from the 14 x 29-bit field"""
    random.seed(11)
    vals = [0, 1, 2, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2 ** 380, 2 ** 377 - 1, 2 ** 360, 2 ** 360 - 1, 2 ** 30, 2 ** 29, 2 ** 29 - 1, P - 2 ** 29] + [random.randrange(P) for _ in range(300)]
    out, out2 = np.zeros(6, np.uint64), np.zeros(6, np.uint64)
    for a in vals:
        A = U.fp_abi(a)
        shim.shim_fs_roundtrip(p_(A), p_(out)); assert U.fp_int(out) == a
        shim.shim_fs_sqr(p_(A), p_(out)); assert U.fp_int(out) == a * a % P
        shim.shim_fs_via_fp(p_(A), p_(out), p_(out2)); assert U.fp_int(out) == a and U.fp_int(out2) == a
    for i in range(len(vals) - 3):
        a, b, c, d = vals[i], vals[i + 1], vals[i + 2], vals[i + 3]
        A, B, Cc, D = U.fp_abi(a), U.fp_abi(b), U.fp_abi(c), U.fp_abi(d)
        shim.shim_fs_mul(p_(A), p_(B), p_(out)); assert U.fp_int(out) == a * b % P
        shim.shim_fs_mul2(p_(A), p_(B), p_(Cc), p_(D), p_(out)); assert U.fp_int(out) == (a * b - c * d) % P
        shim.shim_fs_submul(p_(A), p_(B), p_(Cc), p_(out)); assert U.fp_int(out) == (a - b) * c % P
        assert ((shim.shim_fs_is_zero(p_(A), p_(B)) & 2) != 0) == (a == b)
        assert shim.shim_fs_is_zero(p_(A), p_(A)) == 3


def test_fs_ops_with_carry_seeded_chains(shim):
    """the opt-in shape of the same products (-DFS_SERIAL_LOW -DFS_ALT_HIGH, DESIGN.md section 10: the carry as the addend of the next column's first
    multiply-add; every other output column started from 2^29 + 2^59): same values, and the bound tracker accepts the extra 2^59 in the column sums"""
    so = os.path.join(HERE, "native", "libfp29_host_shim_serial.so")
    deps = [SRC, os.path.join(HERE, "..", "crypto_amd", "csrc", "fp30s.hip.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-DFS_SERIAL_LOW", "-DFS_ALT_HIGH", "-o", so, SRC])
    alt = C.CDLL(so)
    test_fs_ops(alt)
    random.seed(12)
    out, ref = np.zeros(6, np.uint64), np.zeros(6, np.uint64)
    for _ in range(200):                               # limb for limb what the default shape gives (both are exact; the canonical words must agree)
        A, B = U.fp_abi(random.randrange(P)), U.fp_abi(random.randrange(P))
        alt.shim_fs_mul(p_(A), p_(B), p_(out)); shim.shim_fs_mul(p_(A), p_(B), p_(ref)); assert (out == ref).all()
        alt.shim_fs_sqr(p_(A), p_(out)); shim.shim_fs_sqr(p_(A), p_(ref)); assert (out == ref).all()


def test_fp_inversion_by_division_steps(shim):
    """fp_safegcd.hip.h against the big-integer inverse: edge values, small and large values, random ones, lazily added operands"""
    random.seed(7)
    vals = [0, 1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 2 ** 29, 2 ** 29 - 1, 2 ** 58 + 1, 2 ** 380, 2 ** 377 - 1, 2 ** 377, P - 2 ** 29]
    vals += [random.randrange(P) for _ in range(400)] + [random.randrange(1 << k) for k in range(1, 381, 7)]
    out = np.zeros(6, np.uint64)
    for a in vals:
        for k in (1, 3, 7):
            shim.shim_fp_inv(p_(U.fp_abi(a)), k, p_(out))
            assert U.fp_int(out) == (pow(k * a, -1, P) if a else 0), (a, k)


def test_fp2_ops(shim):
    random.seed(2)
    vals = [(0, 0), (1, 0), (0, 1), (P - 1, P - 1), (P - 1, 1)] + [(random.randrange(P), random.randrange(P)) for _ in range(60)]
    o = np.zeros(12, np.uint64)
    enc = lambda v: np.concatenate([U.fp_abi(v[0]), U.fp_abi(v[1])])
    dec = lambda l: (U.fp_int(l[:6]), U.fp_int(l[6:]))
    for a, b in zip(vals, vals[1:]):
        shim.shim_fp2_mul(p_(enc(a)), p_(enc(b)), p_(o)); assert dec(o) == M.f2_mul(a, b)
        shim.shim_fp2_sqr(p_(enc(a)), p_(o)); assert dec(o) == M.f2_sqr(a)
        shim.shim_fs2_mul(p_(enc(a)), p_(enc(b)), p_(o)); assert dec(o) == M.f2_mul(a, b)          # the same over the signed 30-bit field (fs2_pair.hip.h)
        shim.shim_fs2_sqr(p_(enc(a)), p_(o)); assert dec(o) == M.f2_sqr(a)


def _xyzz_g1(o):
    X, Y, ZZ, ZZZ = [U.fp_int(o[6 * i:6 * i + 6]) for i in range(4)]
    if ZZ == 0 and X == 0:
        return None
    return (X * pow(ZZ, -1, P) % P, Y * pow(ZZZ, -1, P) % P)


def _xyzz_g2(o):
    f = lambda i: (U.fp_int(o[12 * i:12 * i + 6]), U.fp_int(o[12 * i + 6:12 * i + 12]))
    X, Y, ZZ, ZZZ = f(0), f(1), f(2), f(3)
    if ZZ == (0, 0) and X == (0, 0):
        return None
    return (M.f2_mul(X, M.f2_inv(ZZ)), M.f2_mul(Y, M.f2_inv(ZZZ)))


CHAINS = [([], []), ([0], [0]), ([0], [1]), ([0, 1], [0, 0]), ([0, 0], [0, 0]), ([0, 0], [0, 1]), ([0, 0, 0], [0, 1, 0]),
          ([0, 1, 0, 1], [0, 0, 1, 1]), ([0] * 5, [0] * 5), ([1, 2, 3, 1, 2, 3], [0, 0, 0, 1, 1, 1]),
          ([1, 2, 3, 1, 2, 3, 5], [0, 0, 0, 1, 1, 1, 1]), (list(range(12)) * 2, [i % 3 == 0 for i in range(24)])]
TREES = [[0], [0, 1], [0, 0], [0, 1, 2, 3, 4], list(range(12)) * 2, [0, 1, 0, 1], [5] * 8]


@pytest.mark.parametrize("group", [1, 2, 3, 4, 5, 6])
def test_group_law_complete(shim, group):
    """groups 3 / 4 / 5 = G1 / G2 / G2 with the early-return mixed addition over the signed 30-bit field (what the MSM kernels instantiate);
    groups 5 and 6 also run the general addition / doubling in their round-by-round form (xyzz_add_rounds / xyzz_dbl_rounds)"""
    random.seed(3)
    ks = [random.randrange(1, M.R) for _ in range(12)]
    if group in (4, 5):
        pts = [M.g2_mul(M.G2_GEN, k) for k in ks]
        add, neg, enc, dec, W = M.g2_add, M.g2_neg, lambda p: U.g2_abi(p)[0], _xyzz_g2, 48
        chain_fn = lambda a, b, c, d: shim.shim_g2s_madd_chain(a, b, c, group - 4, d)
        tree_fn = shim.shim_g2s_add_tree if group == 4 else shim.shim_g2s_add_tree_rounds      # (5: the round-by-round addition of k_reduce_top's lanes-per-point form)
        o = np.zeros(48, np.uint64)
        for k in (1, 2, 16, 20):
            (shim.shim_g2s_dbl_chain if group == 4 else shim.shim_g2s_dbl_chain_rounds)(p_(enc(pts[2])), k, p_(o))
            assert dec(o) == M.g2_mul(pts[2], 1 << k)
    elif group in (3, 6):                  # (6: the round-by-round addition / doubling)
        pts = [M.g1_mul(M.G1_GEN, k) for k in ks]
        add, neg, enc, dec, W = M.g1_add, M.g1_neg, lambda p: U.g1_abi(p)[0], _xyzz_g1, 24
        chain_fn, tree_fn = shim.shim_g1s_madd_chain, (shim.shim_g1s_add_tree if group == 3 else shim.shim_g1s_add_tree_rounds)
        o = np.zeros(24, np.uint64)
        for k in (1, 2, 16, 20):
            (shim.shim_g1s_dbl_chain if group == 3 else shim.shim_g1s_dbl_chain_rounds)(p_(enc(pts[2])), k, p_(o))
            assert dec(o) == M.g1_mul(pts[2], 1 << k)
    elif group == 1:
        pts = [M.g1_mul(M.G1_GEN, k) for k in ks]
        add, neg, enc, dec, W = M.g1_add, M.g1_neg, lambda p: U.g1_abi(p)[0], _xyzz_g1, 24
        chain_fn, tree_fn = shim.shim_g1_madd_chain, shim.shim_g1_add_tree
    else:
        pts = [M.g2_mul(M.G2_GEN, k) for k in ks]
        add, neg, enc, dec, W = M.g2_add, M.g2_neg, lambda p: U.g2_abi(p)[0], _xyzz_g2, 48
        chain_fn, tree_fn = shim.shim_g2_madd_chain, shim.shim_g2_add_tree

    def expect(idx, ng):
        acc = None
        for i, s in zip(idx, ng):
            acc = add(acc, neg(pts[i]) if s else pts[i])
        return acc

    for idx, ng in CHAINS:
        arr = np.concatenate([enc(pts[i]) for i in idx]) if idx else np.zeros(0, np.uint64)
        o = np.zeros(W, np.uint64)
        chain_fn(p_(arr), p_(np.array(ng, np.uint8)), len(idx), p_(o))
        assert dec(o) == expect(idx, ng), (group, idx, ng)
    for idx in TREES:
        arr = np.concatenate([enc(pts[i]) for i in idx])
        o = np.zeros(W, np.uint64)
        tree_fn(p_(arr), len(idx), p_(o))
        assert dec(o) == expect(idx, [0] * len(idx)), (group, idx)
    # P + (-P) through the general addition, then + Q
    arr = np.concatenate([enc(pts[0]), enc(neg(pts[0])), enc(pts[3])])
    o = np.zeros(W, np.uint64)
    tree_fn(p_(arr), 2, p_(o)); assert dec(o) is None
    tree_fn(p_(arr), 3, p_(o)); assert dec(o) == pts[3]


# ---- pairing tower / Miller-loop lines / Fr: the device code of the pairing and NTT kernels on the host ----
def _f12_rand(rng):
    return tuple(tuple((rng.randrange(P), rng.randrange(P)) for _ in range(3)) for _ in range(2))


def _f12_flat(f):
    return [c for six in f for two in six for c in two]


def _f12_enc(f):
    return np.ascontiguousarray(U.f12_abi(_f12_flat(f)))


def _f2_enc(a):
    return np.ascontiguousarray(np.concatenate([U.fp_abi(a[0]), U.fp_abi(a[1])]))


def test_fp12_products_on_host(shim):
    """pairing29.hip.h under the bound tracker: the dense product (f12_mul, what k_product_tree runs), the same product regrouped into
    18 role products + 6 output combinations (f12_mul_roles, what k_product_tree18 runs) and the sparse product (f12_mul_by_014, what
    k_line_products runs) against the big-integer model; outputs are fed back in (reps) so the value bounds must close, and the two dense
    forms alternate so each must accept the other's outputs"""
    rng = random.Random(12)
    for trial in range(6):
        a, b = _f12_rand(rng), _f12_rand(rng)
        for reps in (1, 2, 7):
            want = M.f12_mul(a, b)
            for _ in range(reps - 1):
                want = M.f12_mul(want, b)
            for fn in (shim.shim_f12_mul, shim.shim_f12_mul_roles):
                o = np.zeros(72, np.uint64)
                fn(p_(_f12_enc(a)), p_(_f12_enc(b)), reps, p_(o))
                assert U.f12_ints(o) == _f12_flat(want), (fn, reps)
        c0, c1, c4 = [(rng.randrange(P), rng.randrange(P)) for _ in range(3)]
        want = a
        for _ in range(5):
            want = M.f12_mul_by_014(want, c0, c1, c4)
        o = np.zeros(72, np.uint64)
        shim.shim_f12_mul_by_014(p_(_f12_enc(a)), p_(_f2_enc(c0)), p_(_f2_enc(c1)), p_(_f2_enc(c4)), 5, p_(o))
        assert U.f12_ints(o) == _f12_flat(want)
    # edge values: zero, one, p - 1 in every coefficient
    one = ((( 1, 0), (0, 0), (0, 0)), ((0, 0), (0, 0), (0, 0)))
    top = tuple(tuple((P - 1, P - 1) for _ in range(3)) for _ in range(2))
    for a, b in ((one, top), (top, top), (top, one)):
        for fn in (shim.shim_f12_mul, shim.shim_f12_mul_roles):
            o = np.zeros(72, np.uint64)
            fn(p_(_f12_enc(a)), p_(_f12_enc(b)), 4, p_(o))
            want = M.f12_mul(a, b)
            for _ in range(3):
                want = M.f12_mul(want, b)
            assert U.f12_ints(o) == _f12_flat(want)


def test_miller_loop_dataflow_on_host(shim):
    """the kernels' dataflow (lines per pair with G2Prepared::from fused, per-step sparse products, dense product, host square-and-multiply)
    run on the host under the bound tracker == the model's textbook Miller loop"""
    ks = [(3, 5), (0x1234567, 0x89abcdef), (M.R - 2, 7)]
    ps = [M.g1_mul(M.G1_GEN, a) for a, _ in ks]
    qs = [M.g2_mul(M.G2_GEN, b) for _, b in ks]
    for n in (1, 2, 3):
        pa = np.ascontiguousarray(np.concatenate([U.g1_abi(p)[0] for p in ps[:n]]))
        qa = np.ascontiguousarray(np.concatenate([U.g2_abi(q)[0] for q in qs[:n]]))
        o = np.zeros(72, np.uint64)
        shim.shim_multi_miller(p_(pa), p_(qa), n, p_(o))
        assert U.f12_ints(o) == _f12_flat(M.multi_miller_loop(ps[:n], qs[:n]))


def test_fast_doubling_step_of_the_sixteen_lane_kernel(shim):
    """line_dbl_step_fast (pairing29.hip.h: what k_miller_lines_hex distributes over a 16-lane row: all-squarings first round, X Y from
    (X + Y)^2, the scaled x 12 carry pass, un-normalised f / Y' / line coefficients, one halving instead of two) under the bound tracker:
    every one of the 68 evaluated lines equals line_dbl_step / line_add_step's, for generator multiples and for random points"""
    rng = random.Random(77)
    ks = [(1, 1), (3, 5), (M.R - 2, 7), (rng.randrange(M.R), rng.randrange(M.R)), (rng.randrange(M.R), rng.randrange(M.R))]
    for a, b in ks:
        pa = np.ascontiguousarray(U.g1_abi(M.g1_mul(M.G1_GEN, a))[0]); qa = np.ascontiguousarray(U.g2_abi(M.g2_mul(M.G2_GEN, b))[0])
        ref = np.zeros(68 * 6 * 6, np.uint64); got = np.zeros_like(ref)
        shim.shim_miller_lines(p_(pa), p_(qa), p_(ref)); shim.shim_miller_lines_fast(p_(pa), p_(qa), p_(got))
        assert (ref == got).all()
        # k_miller_lines_ws's steps (a wave per role: four-lane products, X Y as a product, e^2 / g^2 as squarings) under the same tracker
        got[:] = 0; shim.shim_miller_lines_ws(p_(pa), p_(qa), p_(got))
        assert (ref == got).all()
    out = np.zeros(6, np.uint64)
    for v in [0, 1, P - 1, (P - 1) // 2, 2 ** 380, 2 ** 377 - 1] + [rng.randrange(P) for _ in range(60)]:
        for k in (1, 2, 5, 7):
            shim.shim_fp_mul12(p_(U.fp_abi(v)), k, p_(out)); assert U.fp_int(out) == 12 * k * v % P


def test_fr_ops_on_host(shim):
    """fr29.hip.h (the NTT's field) under the bound tracker: Montgomery and canonical products, a chain of lazy butterflies"""
    rng = random.Random(5)
    enc = lambda v: np.ascontiguousarray(np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], np.uint64))
    dec = lambda o: sum(int(o[i]) << (64 * i) for i in range(4))
    RI = pow(M.FR_R, -1, M.R)
    for trial in range(40):
        a, b, w = (rng.randrange(M.R) for _ in range(3))
        if trial == 0:
            a, b, w = M.R - 1, M.R - 1, M.R - 1
        o = np.zeros(4, np.uint64)
        shim.shim_fr_mul(p_(enc(a)), p_(enc(b)), 0, p_(o)); assert dec(o) == a * b % M.R
        shim.shim_fr_mul(p_(enc(a * M.FR_R % M.R)), p_(enc(b * M.FR_R % M.R)), 1, p_(o)); assert dec(o) * RI % M.R == a * b % M.R
        x, y = a, b
        for _ in range(30):
            t = y * w % M.R
            x, y = (x + t) % M.R, (x - t) % M.R
        ox, oy = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
        shim.shim_fr_butterflies(p_(enc(a)), p_(enc(b)), p_(enc(w)), 30, p_(ox), p_(oy))
        assert (dec(ox), dec(oy)) == (x, y)
