"""Fixed-base batch timing (development helper, GPU box): 2^20 products of one base in G1 and G2 through WindowTable.multiply_many with the per-stage device times."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
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
