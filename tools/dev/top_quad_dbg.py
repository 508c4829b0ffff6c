import sys, numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
lib().dgpu_set_min_gpu_n(0)
G, curve = O.G1, ca.G1
for n in (1, 2, 3, 5, 17, 64, 65, 600, 5000):
    bases, _, _ = U.seq_bases(G, n, 77, threads=8)
    for kind in ("one", "small", "rand"):
        sc = O.rand_scalars(78, n)
        if kind == "one": sc[:] = 0; sc[:, 0] = 1
        if kind == "small": sc[:, 1:] = 0; sc[:, 0] &= np.uint64(0xffff)
        res = {}
        for lanes in (1, 4):
            lib().dgpu_set_reduce_lanes(lanes)
            res[lanes] = ca.msm_bigint(curve, bases, sc)
        print(n, kind, "same" if (res[1] == res[4]).all() else "MISMATCH", flush=True)
