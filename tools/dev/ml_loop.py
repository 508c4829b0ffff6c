"""Development helper (GPU box): K 1024-pair Miller loops, one call in flight (the two-launch form), then K verifier-shaped mixed calls — the
loop rocprofv3 wraps for the pairing kernels' statistics."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/oracle", ROOT]
import numpy as np
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
ca.init(0)
K = int(os.environ.get("K", "12")); n = int(os.environ.get("PAIRS", "1024"))
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
f = ca.multi_miller_loop(ps, qs)
t0 = time.perf_counter()
for _ in range(K): assert (ca.multi_miller_loop(ps, qs) == f).all()
print("%d-pair multi_miller_loop: %.3f ms per call" % (n, (time.perf_counter() - t0) / K * 1e3))
pc = pairing.G2Prepared.from_affine(qs[:3])
g = pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]])
t0 = time.perf_counter()
for _ in range(K): assert (pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]]) == g).all()
print("verifier's call (1 affine + 2 prepared pairs): %.3f ms per call" % ((time.perf_counter() - t0) / K * 1e3))
