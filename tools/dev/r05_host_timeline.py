"""Development helper (GPU box): host-side timeline of the bench's in-flight loop — start / end of every dgpu_msm_g1_resident call per host thread.
EXP=twin_load_only loads the development twin first (the state in which the same loop runs at full rate)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import _native, serde, fixed_base as FB
import bench as B
if os.environ.get("EXP") == "twin_load_only":
    _native.dev_lib()
ca.init(0)
n = 1 << 20
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
with FB.WindowTable(ca.G1, gen1[0]) as t:
    db = t.multiply_many_to_bases(B.seeded_scalars(1, n))
db.precompute(); ds = ca.DeviceScalars(B.seeded_scalars(2, n))
from concurrent.futures import ThreadPoolExecutor
K = int(os.environ.get("INFLIGHT", "6"))
pool = ThreadPoolExecutor(K)
log = []
def call(i):
    t0 = time.perf_counter(); db.msm_resident(ds); t1 = time.perf_counter()
    log.append((t0, t1, threading.get_ident(), i))
list(pool.map(call, range(2 * K)))
if os.environ.get("SLEEP"):
    time.sleep(float(os.environ["SLEEP"]))
log.clear()
T0 = time.perf_counter()
futs = []
for i in range(36):
    futs.append(pool.submit(call, i))
Ts = time.perf_counter()
for f in futs:
    f.result()
T1 = time.perf_counter()
print("submitting 36 tasks took %.2f ms" % ((Ts - T0) * 1e3))
print("inflight %d: %.3f ms per MSM" % (K, (T1 - T0) / 36 * 1e3))
tids = {t: k for k, t in enumerate(sorted(set(x[2] for x in log)))}
for t0, t1, tid, i in sorted(log)[:30]:
    print("call %2d thread %d  start %7.2f  end %7.2f  dur %6.2f ms" % (i, tids[tid], (t0 - T0) * 1e3, (t1 - T0) * 1e3, (t1 - t0) * 1e3))
