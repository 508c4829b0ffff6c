"""Development helper (GPU box): per-call times of the plain Miller loop right after a run of dgpu_multi_miller_loop_scaled calls (an after-effect seen in r05_scaled_ml_time.py)"""
import sys, os; R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_, R_ + "/oracle", R_ + "/tests"]
import time, numpy as np, oracle_c as O, crypto_amd as ca, bench as B
from crypto_amd import pairing, fixed_base as FB
from crypto_amd.pairing_check import g1_scale_each
ca.init(0); n = 1024
with FB.WindowTable(ca.G1, O.G1.generator()) as t1, FB.WindowTable(ca.G2, O.G2.generator()) as t2:
    A, _ = t1.multiply_many(B.seeded_scalars(1, n)); Q, _ = t2.multiply_many(B.seeded_scalars(2, n))
m = B.seeded_scalars(3, n)
def seq(f, k):
    out = []
    for _ in range(k):
        t0 = time.perf_counter(); f(); out.append((time.perf_counter() - t0) * 1e3)
    return " ".join("%.2f" % x for x in out)
plain = lambda: pairing.multi_miller_loop(A, Q)
scaled = lambda: pairing.multi_miller_loop_scaled(A, m, Q)
scale = lambda: g1_scale_each(A, m)
for r in range(2):
    a0 = ca.device_alloc_count()
    print("plain :", seq(plain, 12))
    print("scaled:", seq(scaled, 12))
    print("plain :", seq(plain, 40))
    print("scale :", seq(scale, 12))
    print("plain :", seq(plain, 12))
    print("device allocations in this round:", ca.device_alloc_count() - a0)
