"""Development helper (GPU box): prod e([m_i] P_i, Q_i) for 1024 pairs — dgpu_multi_miller_loop_scaled against dgpu_g1_scale_batch + dgpu_multi_miller_loop,
and the plain Miller loop, each timed alone in rotating order"""
import sys, os; R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_, R_ + "/oracle", R_ + "/tests"]
import time, numpy as np, oracle_c as O, crypto_amd as ca, bench as B
from crypto_amd import pairing, fixed_base as FB
from crypto_amd.pairing_check import g1_scale_each
ca.init(0); n = int(os.environ.get("N", "1024"))
with FB.WindowTable(ca.G1, O.G1.generator()) as t1, FB.WindowTable(ca.G2, O.G2.generator()) as t2:
    A, _ = t1.multiply_many(B.seeded_scalars(1, n)); Q, _ = t2.multiply_many(B.seeded_scalars(2, n))
m = B.seeded_scalars(3, n)
def timed(f, k=30):
    f(); f(); t0 = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t0) / k * 1e3
fs = {"scaled (one call)": lambda: pairing.multi_miller_loop_scaled(A, m, Q),
      "scale, then Miller": lambda: pairing.multi_miller_loop(g1_scale_each(A, m)[0], Q),
      "Miller alone": lambda: pairing.multi_miller_loop(A, Q),
      "scale alone": lambda: g1_scale_each(A, m)}
names = list(fs)
for r in range(3):
    for k in names[r:] + names[:r]:
        print("%-20s %.3f ms" % (k, timed(fs[k])), flush=True)
