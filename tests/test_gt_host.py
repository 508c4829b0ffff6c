"""CPU: GT arithmetic on the host cores (crypto_amd/csrc/dock_gt.cpp: dgpu_fp12_pow, dgpu_fp12_multi_pow, dgpu_gt_in_subgroup — host code, no device)
against the CPU oracle.  The multi-exponentiation takes cyclotomic shortcuts (Granger-Scott squarings, conjugates for negative digits) only for bases
that pass the cyclotomic-subgroup test and generic arithmetic otherwise: both paths, and mixtures, give the oracle's value for GT elements, for raw
Miller-loop outputs (outside the subgroup) and for elements of the cyclotomic subgroup outside GT.  The GT membership test (a Frobenius identity and
f^p == f^x, Scott 2021 / blst) agrees with f^r == 1 (what ark-ec's `PairingOutput::check` computes) on all three kinds."""
import ctypes as C
import numpy as np
import oracle_c as O
import util as U
from crypto_amd._native import lib

R, P = U.R, U.P
p_ = lambda a: a.ctypes.data_as(C.c_void_p)
limbs = lambda v: np.array([(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)


def elements():
    g1 = lambda k: O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(k, 4)))[0]
    g2 = lambda k: O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(k, 4)))[0]
    raw = [np.asarray(O.multi_miller_loop(g1(3 + i).reshape(1, 12), g2(5 + 2 * i).reshape(1, 24)), dtype=np.uint64).reshape(72) for i in range(3)]      # not even cyclotomic
    gt = [O.final_exponentiation(m) for m in raw]                                                                                                       # order r
    cyc = [O.fp12_pow(m, (P ** 6 - 1) * (P ** 2 + 1)) for m in raw[:2]]                                                                                 # cyclotomic subgroup, not GT
    return raw, gt, cyc


def pow_(a, e):
    out = np.zeros(72, np.uint64)
    assert lib().dgpu_fp12_pow(p_(np.ascontiguousarray(a)), p_(limbs(e)), p_(out)) == 0
    return out


def multi_pow(bases, exps):
    a = np.ascontiguousarray(np.stack(bases)); e = np.ascontiguousarray(np.stack([limbs(x) for x in exps])); out = np.zeros(72, np.uint64)
    assert lib().dgpu_fp12_multi_pow(p_(a), p_(e), len(bases), p_(out)) == 0
    return out


def oracle_multi(bases, exps):
    acc = O.fp12_one()
    for b, e in zip(bases, exps):
        acc = O.fp12_mul(acc, O.fp12_pow(b, e))
    return acc


def test_pow_and_multi_pow_on_every_kind_of_base():
    raw, gt, cyc = elements()
    rng = np.random.default_rng(12)
    rnd = lambda: int.from_bytes(rng.bytes(32), "little")
    exps = [0, 1, 2, 7, 8, 9, 15, 16, R - 1, R, R + 1, 2 ** 255 - 1, 2 ** 256 - 1, int("8" * 64, 16), int("7" * 64, 16), int("9" * 64, 16), rnd(), rnd() % R]
    for kind in (gt, cyc, raw):
        for e in exps:
            assert (pow_(kind[0], e) == O.fp12_pow(kind[0], e)).all(), hex(e)
    for bases in (gt, gt + cyc, gt + raw[:1], raw, cyc + raw, [gt[0]] * 5 + [gt[1]] * 4):      # all-cyclotomic sets take the short path, one raw element sends the set down the generic one
        for rep in range(3):
            ex = [rnd() % R if rep else exps[(3 * k + 1) % len(exps)] for k in range(len(bases))]
            assert (multi_pow(bases, ex) == oracle_multi(bases, ex)).all()
    one = O.fp12_one()
    assert (multi_pow([one, gt[0]], [5, 0]) == one).all()


def test_gt_membership_agrees_with_f_to_the_r():
    raw, gt, cyc = elements()
    one = O.fp12_one(); zero = np.zeros(72, np.uint64)
    prod = O.fp12_mul(gt[0], O.fp12_pow(gt[1], 12345))                    # GT is a group
    h_only = O.fp12_pow(cyc[0], R)                                        # order divides the cofactor of r in Phi_12(p)
    allv = gt + [one, prod] + cyc + [h_only, O.fp12_mul(gt[0], cyc[1])] + raw + [zero]
    ok = np.zeros(len(allv), np.uint8)
    a = np.ascontiguousarray(np.stack(allv))
    assert lib().dgpu_gt_in_subgroup(p_(a), len(allv), p_(ok)) == 0
    want = [bool((O.fp12_pow(f, R) == one).all()) and bool(f.any()) for f in allv]
    assert [bool(x) for x in ok] == want
    assert want[:5] == [True] * 5 and not any(want[5:])                   # (and the test cases are what they claim to be)
    assert lib().dgpu_gt_in_subgroup(None, 2, p_(ok)) == -3 and lib().dgpu_gt_in_subgroup(None, 0, None) == 0
