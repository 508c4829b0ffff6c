"""Development helper (GPU box): ms per MSM with six calls in flight on one resident table, same scalars every call.  Used with temporary
DGPU_X_SKIP_* switches (not in the product) to measure what each stage of a call costs in flight — results are in DESIGN.md §4."""
import os, sys, time
sys.path.insert(0, "/root/repo")
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import crypto_amd as ca
from crypto_amd import fixed_base as FB, serde
import bench as B
ca.init(0)
n = 1 << 20
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
with FB.WindowTable(ca.G1, gen1[0]) as t:
    db = t.multiply_many_to_bases(B.seeded_scalars(1, n))
db.precompute()
dss = [ca.DeviceScalars(B.seeded_scalars(2, n)) for _ in range(1)]
pool = ThreadPoolExecutor(6)
def run(k): return db.msm_resident(dss[0])
list(pool.map(run, range(60)))
t0 = time.perf_counter(); list(pool.map(run, range(120))); dt = (time.perf_counter() - t0) / 120 * 1e3
print("%.3f ms per MSM (6 in flight) = %.1f MSM/s" % (dt, 1e3 / dt))
