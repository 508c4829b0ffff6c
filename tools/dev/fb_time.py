import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, crypto_amd as ca, oracle_c as O
from crypto_amd import fixed_base as FB
ca.init(0)
sc = O.rand_scalars(3, 1 << 20)
for cv, G in ((ca.G1, O.G1), (ca.G2, O.G2)):
    with FB.WindowTable(cv, G.generator()) as t:
        t.multiply_many(sc[:1000])
        ca.prof.enable(True); ca.prof.reset()
        t0 = time.time(); out, inf = t.multiply_many(sc); dt = time.time() - t0
        print(cv, "2^20 products wall ms", round(dt * 1e3, 1), {k: round(v[0], 2) for k, v in ca.prof.read().items()})
        ca.prof.enable(False)
