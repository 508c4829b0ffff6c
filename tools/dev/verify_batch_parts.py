"""Development helper (GPU box): the pieces of dgpu_legogroth16_verify_batch for 1024 proofs, timed one by one through the ABI"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tests"]
import numpy as np
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing, fixed_base as FB
from crypto_amd.pairing_check import g1_scale_each, fp12_pow
import bench as B
ca.init(0)
n = 1024
with FB.WindowTable(ca.G1, O.G1.generator()) as t1, FB.WindowTable(ca.G2, O.G2.generator()) as t2:
    A, _ = t1.multiply_many(B.seeded_scalars(1, n)); Q, _ = t2.multiply_many(B.seeded_scalars(2, n))
m = B.seeded_scalars(3, n)
pc = pairing.G2Prepared.from_affine(Q[:2])
def timed(f, k=10):
    f(); f(); t0 = time.perf_counter()
    for _ in range(k): r = f()
    return (time.perf_counter() - t0) / k * 1e3
print("scale 1024 points, one scalar each: %.3f ms" % timed(lambda: g1_scale_each(A, m, None)))
print("MSM 1024 terms one-shot:            %.3f ms" % timed(lambda: ca.msm_bigint(ca.G1, A, m)))
f = pairing.multi_miller_loop(np.concatenate([A, A[:2]]), [Q, pc[0:1], pc[1:2]])
print("Miller loop 1024 affine + 2 prepared: %.3f ms" % timed(lambda: pairing.multi_miller_loop(np.concatenate([A, A[:2]]), [Q, pc[0:1], pc[1:2]])))
print("final exponentiation:               %.3f ms" % timed(lambda: ca.final_exponentiation(f)))
g = ca.final_exponentiation(f)
print("GT power (255 bits):                %.3f ms" % timed(lambda: fp12_pow(g, 0x5EED0029 ** 7 % B.R_MOD)))
