# tools/dev/verify_time.py — the verifier's Miller loop (one affine pair + two prepared, legogroth16/src/verifier.rs:69-76) by kernel form
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
from crypto_amd._native import lib
ca.init(0)
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
for n, cut in ((3, 1), (64, 32), (1024, 512)):
    ps = O.G1.gen_seq(k0, d, n, threads=8); qs = O.G2.gen_seq(d, k0, n, threads=8)
    pc = pairing.G2Prepared.from_affine(qs)
    items = [qs[:cut], pc[cut:]]
    res = {}
    for mode in (3, 7, 3, 7):
        lib().dgpu_set_miller_pipeline(mode)
        for _ in range(3): pairing.multi_miller_loop(ps, items)
        t0 = time.perf_counter()
        for _ in range(20): pairing.multi_miller_loop(ps, items)
        res.setdefault(mode, []).append((time.perf_counter() - t0) / 20 * 1e3)
    print("%d pairs (%d affine): ms per mixed call by mode %s" % (n, cut, {m: ["%.3f" % v for v in r] for m, r in res.items()}), flush=True)
lib().dgpu_set_miller_pipeline(7)
