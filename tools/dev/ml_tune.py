# tools/dev/ml_tune.py — development (a -DDGPU_DEV build of dock_pairing.hip, DGPU_LIB pointing at it): where the Miller chain is cut and the
# slice length of the last piece's products; each configuration in a fresh process (the switches are read per call, but keep runs independent)
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, time, os
sys.path[:0] = ["%s/oracle", "%s"]
import oracle_c as O, crypto_amd as ca
ca.init(0)
out = []
for n in (3, 1024, 2048):
    k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
    ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
    for _ in range(5): ca.multi_miller_loop(ps, qs)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20): ca.multi_miller_loop(ps, qs)
        best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
    out.append("%%d: %%.3f" %% (n, best))
print(os.environ.get("DGPU_ML_CUTS"), os.environ.get("DGPU_ML_TAIL_SLICE"), "  ".join(out), flush=True)
''' % (R, R)
for cuts in ("40,17", "40,14", "38,12", "36,10", "42,20", "44,24"):
    for ts in ("", "2", "1"):
        env = dict(os.environ, DGPU_LIB=R + "/crypto_amd/libdock_gpu_dev.so", DGPU_ML_CUTS=cuts)
        if ts: env["DGPU_ML_TAIL_SLICE"] = ts
        subprocess.call([sys.executable, "-c", code], env=env)
