# tools/dev/agg_parts.py — the per-round calls of the aggregation (one GIPA round at split s), timed one by one
import sys, os, time, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd.aggregation import ops
from crypto_amd import pairing
from crypto_amd.fixed_base import WindowTable
ca.init(0)
rng = np.random.default_rng(3)
ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % (U.R - 1) + 1 for _ in range(k)]
def fixed(curve, g, ks):
    with WindowTable(curve, g, len(ks)) as t:
        return t.multiply_many(ks)[0]
g, h = O.G1.generator(), O.G2.generator()
def timed(f, k=10):
    f(); f(); t0 = time.perf_counter()
    for _ in range(k): r = f()
    return (time.perf_counter() - t0) / k * 1e3
for s in (1, 4, 32, 512):
    P, Q = fixed(ca.G1, g, ints(2 * s)), fixed(ca.G2, h, ints(2 * s))
    jobs = [(P, Q)] * 4 + [(P[:s], Q[:s])] * 10
    sc = ints(s)
    t_pair = timed(lambda: ops.multi_pairings(jobs))
    t_ml = timed(lambda: pairing.multi_miller_loops(jobs))
    t_msm = timed(lambda: ops.msm(ca.G1, P[:s], sc))
    t_f1 = timed(lambda: ops.mul_add(ca.G1, np.concatenate([P] * 3)[:5 * s], sc[0], np.concatenate([P] * 3)[:5 * s]))
    t_f2 = timed(lambda: ops.mul_add(ca.G2, np.concatenate([Q] * 2)[:3 * s], sc[0], np.concatenate([Q] * 2)[:3 * s]))
    print("split %4d: 14 multi-pairings %.2f ms (Miller loops alone %.2f), one MSM %.2f, fold G1 %.2f, fold G2 %.2f" % (s, t_pair, t_ml, t_msm, t_f1, t_f2), flush=True)
