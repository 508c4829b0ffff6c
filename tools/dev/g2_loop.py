"""Development helper (GPU box): K resident G2 MSMs of 2^LOG2N terms on a precomputed table, one call in flight — the loop rocprofv3 wraps
for the G2 kernel statistics / counters (tools/dev/round3_profiles.sh)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import serde, fixed_base as FB
import bench as B
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
if os.environ.get("REDUCE_LANES"):
    from crypto_amd._native import lib
    assert lib().dgpu_set_reduce_lanes(int(os.environ["REDUCE_LANES"])) == 0     # A/B of the bucket reduction's forms (tools/dev/r05_reduce_ab2.sh)
n = 1 << int(os.environ.get("LOG2N", "20")); K = int(os.environ.get("K", "12"))
gen2, _ = serde.deserialize(ca.G2, bytes.fromhex(B.G2_GEN_COMPRESSED))
with FB.WindowTable(ca.G2, gen2[0]) as t2:
    db = t2.multiply_many_to_bases(B.seeded_scalars(0x5EED0003, n))
if not os.environ.get("PLAIN"):
    db.precompute()
ds = ca.DeviceScalars(B.seeded_scalars(0x5EED1000, n))
r0 = db.msm_resident(ds)
ca.prof.enable(True); ca.prof.reset()
t0 = time.perf_counter()
for _ in range(K):
    assert (db.msm_resident(ds) == r0).all()
dt = (time.perf_counter() - t0) / K * 1e3
print("G2 MSM n=2^%d: %.3f ms per call; stages:" % (n.bit_length() - 1, dt), {k: round(v[0] / max(1, v[1]), 4) for k, v in ca.prof.read().items()})
