import os, sys, time
sys.path[:0] = ["/root/repo"]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
import crypto_amd as ca, bench as B
from crypto_amd import serde, fixed_base as FB
sys.setswitchinterval(float(os.environ.get("SWI", "1e-4")))
ca.init(0)
n = 1 << 20
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
ks = B.seeded_scalars(0x5EED2000, n); scalars = B.seeded_scalars(0x5EED1000, n)
with FB.WindowTable(ca.G1, gen1[0]) as gtab: db = gtab.multiply_many_to_bases(ks)
db.precompute(); ds = ca.DeviceScalars(scalars)
def pad():
    r = db.msm_resident(ds)
    for _ in range(int(os.environ.get("PAD", "0"))): pass
    return r
for T in (6, 8, 8, 9, 8):
    B.InFlight(pad, 2 * T, T).go()
    w = B.InFlight(pad, 20, T); t0 = time.perf_counter(); w.go(); dt = time.perf_counter() - t0
    print("T=%d: %.3f ms per call" % (T, dt / 20 * 1e3), flush=True)
