"""Development helper (GPU box): the headline's loop (resident table, 2^20 terms, T Python threads) in a process WITHOUT torch — the library then runs on the
system ROCm runtime instead of the one bundled with the torch wheel (crypto_amd/_native.py loads torch first when it is importable).  NOTORCH=1 T=6 python tools/dev/inflight_probe3.py"""
import os, sys, time, threading, itertools
sys.path[:0] = ["/root/repo"]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if os.environ.get("NOTORCH"): sys.modules["torch"] = None
import numpy as np
import crypto_amd as ca
from crypto_amd import serde, fixed_base as FB
sys.setswitchinterval(1e-4)
ca.init(0)
print("torch loaded:", sys.modules.get("torch") is not None, [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
n = 1 << 20
G1_GEN = "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(G1_GEN))
def scal(seed):
    a = np.random.Generator(np.random.PCG64(seed)).integers(0, 1 << 64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(2); return a
ks, sc = scal(1), scal(2)
with FB.WindowTable(ca.G1, gen1[0]) as gtab: db = gtab.multiply_many_to_bases(ks)
db.precompute(); ds = ca.DeviceScalars(sc)
def inflight(count, T):
    nxt = itertools.count(); bar = threading.Barrier(T + 1)
    def run():
        bar.wait()
        while next(nxt) < count: db.msm_resident(ds)
    th = [threading.Thread(target=run) for _ in range(T)]
    for t in th: t.start()
    t0 = time.perf_counter(); bar.wait()
    for t in th: t.join()
    return time.perf_counter() - t0
for T in (6, 6, 8, 6):
    inflight(2 * T, T); inflight(2 * T, T)
    print("T=%d: %.3f ms per call" % (T, min(inflight(20, T) for _ in range(3)) / 20 * 1e3), flush=True)
