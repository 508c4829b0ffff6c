# tools/dev/prep_time.py — dgpu_g2_prepare (G2Prepared::from) of 1024 points: lane-pair chain with conversions inside / four-lane chain + parallel conversion
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
from crypto_amd._native import lib
ca.init(0)
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
for n in (3, 64, 1024, 8192):
    qs = O.G2.gen_seq(d, k0, n, threads=32)
    res = {}; ref = None
    for mode in (2, 3, 2, 3):
        lib().dgpu_set_miller_pipeline(mode)
        pc = pairing.G2Prepared.from_affine(qs)
        if ref is None: ref = pc.coeffs.copy()
        assert (pc.coeffs == ref).all()
        t0 = time.perf_counter()
        for _ in range(10): pairing.G2Prepared.from_affine(qs)
        res.setdefault(mode, []).append((time.perf_counter() - t0) / 10 * 1e3)
    print("n=%d  ms per dgpu_g2_prepare: lane-pair chain %s   four-lane chain + conversion pass %s" % (n, ["%.3f" % v for v in res[2]], ["%.3f" % v for v in res[3]]), flush=True)
lib().dgpu_set_miller_pipeline(3)
