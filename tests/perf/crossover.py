"""Timing driver (GPU box): where the one-shot GPU MSM (dgpu_msm_g1: host bases + scalars in, point out) overtakes the CPU path — the
measured value behind DGPU_DEFAULT_MIN_GPU_N.  CPU = the oracle's arkworks-style Pippenger with one thread per window (rayon's structure)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import crypto_amd as ca
import oracle_c as O
ca.init(0)
from crypto_amd._native import lib as _lib
_lib().dgpu_set_min_gpu_n(1)      # measure below the shipped threshold too
k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
N = 1 << 14
bases = O.G1.gen_seq(k0, d, N, threads=32); sc = O.rand_scalars(3, N)
ncpu = os.cpu_count() or 1
def best(fn, k=7):
    fn(); ts = []
    for _ in range(k):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
for lg in range(4, 15):
    n = 1 << lg
    c = O.window_c(n); nw = (255 + c - 1) // c
    thr = max(1, min(ncpu, nw))
    g = best(lambda: ca.msm_bigint(ca.G1, bases[:n], sc[:n]))
    cpu = best(lambda: O.G1.msm(bases[:n], sc[:n], threads=thr), 5)
    cpu1 = best(lambda: O.G1.msm(bases[:n], sc[:n], threads=1), 3) if lg <= 12 else float("nan")
    print("n=2^%-2d  gpu one-shot %.3f ms | cpu %2d threads %.3f ms | cpu 1 thread %.3f ms" % (lg, g, thr, cpu, cpu1), flush=True)
