# tools/dev/small_msm_time.py — one-shot MSM latency at small sizes: the tree path (small_kernels.hip.h) against the bucket pipeline
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
lib().dgpu_set_min_gpu_n(1)
for gname in ("G1", "G2"):
    curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
    for n in (16, 128, 600, 2048, 4096, 8192):
        bases, _, _ = U.seq_bases(G, n, 77, threads=32); sc = O.rand_scalars(78, n)
        db = ca.DeviceBases(curve, bases); ds = ca.DeviceScalars(sc)
        out = {}
        for mx in (8192, 0, 8192, 0):
            lib().dgpu_set_small_msm_max(mx)
            r = ca.msm_bigint(curve, bases, sc)
            for _ in range(3): ca.msm_bigint(curve, bases, sc)
            t0 = time.perf_counter()
            for _ in range(20): ca.msm_bigint(curve, bases, sc)
            t1 = (time.perf_counter() - t0) / 20 * 1e3
            for _ in range(3): db.msm_resident(ds)
            t0 = time.perf_counter()
            for _ in range(20): rr = db.msm_resident(ds)
            t2 = (time.perf_counter() - t0) / 20 * 1e3
            assert (rr == r).all()
            out.setdefault(mx, []).append((t1, t2, r))
        same = (out[0][0][2] == out[8192][0][2]).all()
        f = lambda v: "%.3f/%.3f" % (v[0], v[1])
        print(gname, n, "same" if same else "MISMATCH", "ms one-shot/resident: tree", [f(v) for v in out[8192]], " buckets", [f(v) for v in out[0]], flush=True)
        db.free(); ds.free()
lib().dgpu_set_small_msm_max(8192)
