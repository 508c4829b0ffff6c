import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
for gname, n in (("G1", 1 << 20), ("G2", 1 << 18)):
    curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
    bases, _, _ = U.seq_bases(G, n, 77, threads=64)
    sc = O.rand_scalars(78, n)
    tab = ca.DeviceBases(curve, bases).precompute(0)
    for lanes in (1, 4):
        lib().dgpu_set_reduce_lanes(lanes)
        for _ in range(6): tab.msm_bigint(sc)
