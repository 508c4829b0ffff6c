# tools/dev/gt_time.py — host-side pieces of the aggregation verifier: the GT multi-exponentiation and the small linear combinations
import sys, os, time, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd.aggregation import ops
ca.init(0)
g, h = O.G1.generator(), O.G2.generator()
e11 = O.final_exponentiation(O.multi_miller_loop(g.reshape(1, 12), h.reshape(1, 24)))
rng = np.random.default_rng(1)
ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % U.R for _ in range(k)]
def timed(f, k=10):
    f(); t0 = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t0) / k * 1e3
for n in (1, 2, 8, 21):
    bases = [ops.fp12_pow(e11, x) for x in ints(n)]; ex = ints(n)
    print("GT multi-pow of %2d bases: %.3f ms" % (n, timed(lambda: ops.gt_multi_pow(bases, ex))))
    print("   5 of them from 5 threads: %.3f ms" % timed(lambda: ops.parallel([lambda: ops.gt_multi_pow(bases, ex)] * 5, host=True)))
P = np.stack([g] * 13); Q = np.stack([h] * 2)
print("G1 lincomb 13 terms %.3f ms, 2 terms %.3f; G2 lincomb 2 terms %.3f ms" % (timed(lambda: ops.msm(ca.G1, P, ints(13))), timed(lambda: ops.msm(ca.G1, P[:2], ints(2))), timed(lambda: ops.msm(ca.G2, Q, ints(2)))))
