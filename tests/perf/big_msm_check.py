"""One-off (GPU box): G1 MSM at n = 2^LOG2N (default 26) on one GPU, table and plain, against the closed form over seeded discrete logs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import fixed_base as FB, serde
import bench as B
ca.init(0)
n = 1 << int(os.environ.get("LOG2N", "26"))
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
ks = B.seeded_scalars(11, n); sc = B.seeded_scalars(12, n)
t0 = time.time(); tot = B.dot_mod_r(ks, sc); print("closed form %.1f s" % (time.time() - t0), flush=True)
with FB.WindowTable(ca.G1, gen1[0]) as t:
    t0 = time.time(); db = t.multiply_many_to_bases(ks); print("bases %.1f s" % (time.time() - t0), flush=True)
    exp_xy, _ = t.multiply(tot)
ds = ca.DeviceScalars(sc)
for _ in range(2):
    t0 = time.time(); r = db.msm_resident(ds); dt = time.time() - t0
print("plain  n=2^%d: %.1f ms, closed form %s" % (n.bit_length() - 1, dt * 1e3, bool((r[:12] == exp_xy).all())), flush=True)
t0 = time.time(); db.precompute(); print("table build %.1f s" % (time.time() - t0), flush=True)
for _ in range(2):
    t0 = time.time(); r2 = db.msm_resident(ds); dt = time.time() - t0
print("table  n=2^%d: %.1f ms, closed form %s, == plain %s" % (n.bit_length() - 1, dt * 1e3, bool((r2[:12] == exp_xy).all()), bool((r2 == r).all())), flush=True)
