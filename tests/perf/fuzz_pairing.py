"""One-off randomized cross-check of multi_miller_loop (raw Fp12) and final_exponentiation against the CPU oracle: batch sizes around the
slice / group boundaries of the product kernels, identity members.  Usage: ITER=200 python tests/perf/fuzz_pairing.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import crypto_amd as ca
import oracle_c as O

ca.init(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
P = O.G1.gen_seq(k0, d, 9000, threads=32); Q = O.G2.gen_seq(d, k0, 9000, threads=32)
sizes = [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 2047, 2048, 2049, 4100, 8193, 9000]
bad = 0
for it in range(int(os.environ.get("ITER", "100"))):
    n = sizes[it % len(sizes)]
    i0 = int(rng.integers(0, 9000 - n + 1))
    p, q = P[i0:i0 + n].copy(), Q[i0:i0 + n].copy()
    for j in np.nonzero(rng.integers(0, 20, n) == 0)[0]:
        if rng.integers(0, 2): p[j] = 0
        else: q[j] = 0
    skip = np.array([0 if (p[j].any() and q[j].any()) else 1 for j in range(n)], dtype=np.uint8)
    f = ca.multi_miller_loop(p, q)
    ref = O.multi_miller_loop(p, q, skip, threads=32)
    ok = (f == ref).all()
    if ok and it % 10 == 0:
        ok = (ca.final_exponentiation(f) == O.final_exponentiation(ref)).all()
    if not ok:
        bad += 1
        print("MISMATCH", it, n, flush=True)
print("fuzz_pairing: %d iterations, %d mismatches" % (it + 1, bad))
sys.exit(1 if bad else 0)
