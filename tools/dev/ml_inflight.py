# tools/dev/ml_inflight.py — 1024-pair Miller loops, one call at a time and six in flight, by kernel form
import sys, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle_c as O, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
n = 1024
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
for mode in (0, 1, 2, 3, 0, 1, 2, 3):
    lib().dgpu_set_miller_pipeline(mode)
    for _ in range(3): ca.multi_miller_loop(ps, qs)
    t0 = time.perf_counter()
    for _ in range(20): ca.multi_miller_loop(ps, qs)
    one = (time.perf_counter() - t0) / 20 * 1e3
    out = []
    for th in (2, 4, 6, 10):
        with ThreadPoolExecutor(th) as ex:
            list(ex.map(lambda _: ca.multi_miller_loop(ps, qs), range(12)))
            t0 = time.perf_counter(); list(ex.map(lambda _: ca.multi_miller_loop(ps, qs), range(60))); out.append((time.perf_counter() - t0) / 60 * 1e3)
    print("mode %d  one call %.3f ms   per call with 2 / 4 / 6 / 10 in flight: %s" % (mode, one, ["%.3f" % v for v in out]), flush=True)
lib().dgpu_set_miller_pipeline(31)
