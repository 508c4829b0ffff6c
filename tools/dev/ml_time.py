# tools/dev/ml_time.py — latency of dgpu_multi_miller_loop at 1024 pairs, two-launch line kernel against the one-launch form
import time, numpy as np, sys
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle_c as O, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
for n in (3, 64, 256, 1024, 2048, 4096, 8192):
    k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
    ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
    res = {}
    for on in (2, 3, 6, 7, 2, 3, 6, 7):
        lib().dgpu_set_miller_pipeline(on)
        for _ in range(3): ca.multi_miller_loop(ps, qs)
        t0 = time.perf_counter()
        for _ in range(20): f = ca.multi_miller_loop(ps, qs)
        res.setdefault(on, []).append((time.perf_counter() - t0) / 20 * 1e3)
    lib().dgpu_set_miller_pipeline(7)
    print("n=%5d  ms per call by mode (bit 0: two launches, bit 1: 18-role tree, bit 2: sixteen lanes per pair): %s" % (n, {m: ["%.3f" % v for v in r] for m, r in res.items()}))
