"""Development helper (GPU box): the headline MSM (n = 2^20, per-key table) at several forced chunk lengths (dgpu_set_chunk) and buckets per lane of
k_reduce_l0 (dgpu_set_reduce_shift), one call and six calls in flight, interleaved so that the box's clock drift hits every setting alike."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import serde, fixed_base as FB
from crypto_amd._native import lib
import bench as B
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
n = 1 << 20
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
with FB.WindowTable(ca.G1, gen1[0]) as t1:
    db = t1.multiply_many_to_bases(B.seeded_scalars(0x5EED0003, n))
db.precompute()
ds = [ca.DeviceScalars(B.seeded_scalars(0x5EED1000 + k, n)) for k in range(6)]
r0 = [db.msm_resident(d) for d in ds]
def one(K=10):
    t0 = time.perf_counter()
    for _ in range(K): db.msm_resident(ds[0])
    return (time.perf_counter() - t0) / K * 1e3
def six(K=8):
    def w(k):
        for _ in range(K): db.msm_resident(ds[k])
    th = [threading.Thread(target=w, args=(k,)) for k in range(6)]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]
    return (time.perf_counter() - t0) / (6 * K) * 1e3
settings = [("chunk", c) for c in [int(x) for x in os.environ.get("CHUNKS", "0,40,52,64,80,104").split(",")]] + [("shift", s) for s in [int(x) for x in os.environ.get("SHIFTS", "2,4").split(",") if x]]
res = {s: [] for s in settings}
for rep in range(4):
    for s in settings:
        lib().dgpu_set_chunk(s[1] if s[0] == "chunk" else 0); lib().dgpu_set_reduce_shift(s[1] if s[0] == "shift" else -1)
        one(3)
        res[s].append((one(), six()))
lib().dgpu_set_chunk(0); lib().dgpu_set_reduce_shift(-1)
for s in settings:
    a = np.array(res[s]); print(s, "one in flight %.3f ms (min %.3f)   six in flight %.3f ms per MSM (min %.3f)" % (a[:, 0].mean(), a[:, 0].min(), a[:, 1].mean(), a[:, 1].min()))
assert all((db.msm_resident(d) == r).all() for d, r in zip(ds, r0))
