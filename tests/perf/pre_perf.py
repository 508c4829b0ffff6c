"""Timing driver (GPU box): plain vs precomputed-table MSM on resident operands — per-stage HIP-event times with one call in flight,
wall time with 1 and IN_FLIGHT (default 6) calls in flight.  WHAT=g1|g2|both, LOG2N, CS=comma list of table widths (0 = plain pipeline)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import fixed_base as FB, serde
import bench as B

ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
log2n = int(os.environ.get("LOG2N", "20"))
n = 1 << log2n
what = os.environ.get("WHAT", "g1")
cs = [int(x) for x in os.environ.get("CS", "0,20").split(",")]
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
gen2, _ = serde.deserialize(ca.G2, bytes.fromhex(B.G2_GEN_COMPRESSED))
sc = B.seeded_scalars(0x5EED1000, n)
ds = ca.DeviceScalars(sc)
inflight = int(os.environ.get("IN_FLIGHT", "6"))
for curve, gen, tag in ((ca.G1, gen1, "g1"), (ca.G2, gen2, "g2")):
    if what not in (tag, "both"):
        continue
    ref = None
    for c in cs:
        with FB.WindowTable(curve, gen[0]) as t:
            db = t.multiply_many_to_bases(B.seeded_scalars(0x5EED0003, n))
        if c:
            t0 = time.perf_counter(); db.precompute(c); tp = (time.perf_counter() - t0) * 1e3
        else:
            tp = 0.0
        r = db.msm_resident(ds)
        if ref is None:
            ref = r
        assert (r == ref).all()
        for _ in range(3):
            db.msm_resident(ds)
        ca.prof.enable(True); ca.prof.reset()
        t0 = time.perf_counter()
        for _ in range(5):
            db.msm_resident(ds)
        lat = (time.perf_counter() - t0) / 5 * 1e3
        st = ca.prof.read(); ca.prof.enable(False)
        B.run_inflight(lambda: db.msm_resident(ds), 2 * inflight, inflight)
        thr = min(B.run_inflight(lambda: db.msm_resident(ds), 6 * inflight, inflight)[0] for _ in range(3)) / (6 * inflight) * 1e3      # (host threads parked before the clock: bench.py's InFlight)
        print("%s n=2^%d c=%2d  precompute %.1f ms | latency %.3f ms | %d in flight %.3f ms/MSM (%.1f MSM/s) | %s" % (
            tag, log2n, c, tp, lat, inflight, thr, 1e3 / thr, " ".join("%s=%.3f" % (k.replace("msm.", ""), v[0] / max(1, v[1])) for k, v in st.items())), flush=True)
        db.free()
