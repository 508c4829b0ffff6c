# tools/dev/ml_prof.py — stage timers of the one-launch Miller loop (stage timers switch the two-launch form off), old / 18-role tree
import sys, numpy as np, time
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle_c as O, crypto_amd as ca
from crypto_amd._native import lib
from crypto_amd.msm import prof
ca.init(0)
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
for n in (1024, 8192):
    ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
    for mode in (0, 2):
        lib().dgpu_set_miller_pipeline(mode)
        for _ in range(3): ca.multi_miller_loop(ps, qs)
        prof.enable(True); prof.reset()
        t0 = time.perf_counter()
        for _ in range(10): ca.multi_miller_loop(ps, qs)
        wall = (time.perf_counter() - t0) / 10 * 1e3
        r = prof.read(); prof.enable(False)
        print(n, "mode", mode, "wall %.3f ms" % wall, {k: round(v[0] / v[1], 4) for k, v in r.items() if k.startswith("ml.")})
lib().dgpu_set_miller_pipeline(3)
