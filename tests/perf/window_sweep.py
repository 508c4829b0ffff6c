"""Stage timings of one resident G1 MSM (n = 2^20) for window sizes c = 16..20 (development helper): shows why c is capped at 16 -
wider codes slow the sort sweeps and sparse buckets turn the accumulate kernel into a store-bound pass."""
import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import numpy as np, crypto_amd as ca, oracle_c as O
from crypto_amd._native import lib
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
n = 1 << 20
k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
bases = O.G1.gen_seq(k0, d, n, threads=64); sc = O.rand_scalars(3, n)
db = ca.DeviceBases(ca.G1, bases); ds = ca.DeviceScalars(sc)
ref = None
for c in (16, 17, 18, 19, 20):
    assert lib().dgpu_set_window_bits(c) == 0
    r = db.msm_resident(ds)
    if ref is None: ref = r
    assert (r == ref).all()
    ca.prof.enable(True); ca.prof.reset()
    t0 = time.time()
    for _ in range(3): db.msm_resident(ds)
    dt = (time.time() - t0) / 3
    st = ca.prof.read(); ca.prof.enable(False)
    print("c=%d %.3f ms | %s" % (c, dt * 1e3, " ".join("%s=%.3f" % (k.split(".")[1], v[0] / v[1]) for k, v in st.items())), flush=True)
