import sys, numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle_c as O, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
N = 600
ps = O.G1.gen_seq(k0, d, N, threads=32); qs = O.G2.gen_seq(d, k0, N, threads=32)
for n in (1, 2, 3, 4, 5, 8, 9, 16, 17, 33, 64, 65, 129, 200, 257, 600):
    ref = O.multi_miller_loop(ps[:n], qs[:n], None, threads=32)
    out = []
    for m in (0, 1, 2, 3):
        lib().dgpu_set_miller_pipeline(m); out.append("ok" if (ca.multi_miller_loop(ps[:n], qs[:n]) == ref).all() else "BAD")
    print(n, out, flush=True)
