# tools/dev/small_msm_time.py — one-shot MSM latency at small sizes, last reduction kernel with 1 / 4 members per point
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
for gname in ("G1", "G2"):
    curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
    for n in (600, 2048, 8192, 40000):
        bases, _, _ = U.seq_bases(G, n, 77, threads=32); sc = O.rand_scalars(78, n)
        out = {}
        for lanes in (1, 4, 1, 4):
            lib().dgpu_set_reduce_lanes(lanes)
            r = ca.msm_bigint(curve, bases, sc)
            for _ in range(3): ca.msm_bigint(curve, bases, sc)
            t0 = time.perf_counter()
            for _ in range(20): ca.msm_bigint(curve, bases, sc)
            out.setdefault(lanes, []).append(((time.perf_counter() - t0) / 20 * 1e3, r))
        same = (out[1][0][1] == out[4][0][1]).all()
        print(gname, n, "same" if same else "MISMATCH", "ms one-shot: 1 lane", ["%.3f" % v[0] for v in out[1]], " 4 lanes", ["%.3f" % v[0] for v in out[4]], flush=True)
lib().dgpu_set_reduce_lanes(4)
