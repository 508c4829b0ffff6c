"""Timing driver (GPU box): one MSM over a Groth16-like witness (bench.py's distribution: 37.5 % zeros, 12.5 % ones, 25 % 16-bit, 25 % full-size)
on G1 and G2, plain pipeline and tables of several window widths: per-stage times with one call in flight, and ms per MSM with four."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import fixed_base as FB, serde
import bench as B
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
if os.environ.get("CHUNK"):      # terms per lane of the accumulation forced (dgpu_set_chunk)
    from crypto_amd._native import lib as _lib
    assert _lib().dgpu_set_chunk(int(os.environ["CHUNK"])) == 0
n = 1 << int(os.environ.get("LOG2N", "20"))
gens = {ca.G1: serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))[0][0], ca.G2: serde.deserialize(ca.G2, bytes.fromhex(B.G2_GEN_COMPRESSED))[0][0]}
z = B.seeded_scalars(7, n)
rng = np.random.Generator(np.random.PCG64(8)); kd = rng.integers(0, 4, n)
z[kd <= 1] = 0; z[kd == 1, 0] = rng.integers(0, 2, int((kd == 1).sum()), dtype=np.uint64); mk = kd == 2; z[mk, 1:] = 0; z[mk, 0] &= np.uint64(0xFFFF)
dense = B.seeded_scalars(9, n)
pool = ThreadPoolExecutor(4)
for cv in (ca.G1, ca.G2):
    for c in [int(x) for x in os.environ.get("CS", "0,16,18,20").split(",")]:
        with FB.WindowTable(cv, gens[cv]) as t:
            db = t.multiply_many_to_bases(B.seeded_scalars(1, n))
        if c:
            db.precompute(c)
        for name, s in (("witness", z), ("dense", dense)):
            ds = ca.DeviceScalars(s)
            for _ in range(2):
                db.msm_resident(ds)
            ca.prof.enable(True); ca.prof.reset()
            t0 = time.perf_counter()
            for _ in range(3):
                db.msm_resident(ds)
            lat = (time.perf_counter() - t0) / 3 * 1e3
            st = ca.prof.read(); ca.prof.enable(False)
            t0 = time.perf_counter()
            list(pool.map(lambda _: db.msm_resident(ds), range(16)))
            thr = (time.perf_counter() - t0) / 16 * 1e3
            print("%s %-8s %-7s latency %.3f ms, 4 in flight %.3f ms/MSM | %s" % (cv.tag, "c=%d" % c if c else "plain", name, lat, thr,
                  " ".join("%s=%.3f" % (k.replace("msm.", ""), v[0] / max(1, v[1])) for k, v in st.items())), flush=True)
            ds.free()
        db.free()
