"""Timing driver (GPU box): multi_miller_loop at N pairs (per-stage HIP-event times)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import fixed_base as FB, serde
import bench as B
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED)); gen2, _ = serde.deserialize(ca.G2, bytes.fromhex(B.G2_GEN_COMPRESSED))
for n in [int(x) for x in os.environ.get("NS", "3,64,1024,4096,8192").split(",")]:
    with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
        P, _ = t1.multiply_many(B.seeded_scalars(5, n)); Q, _ = t2.multiply_many(B.seeded_scalars(6, n))
    for _ in range(3): ca.multi_miller_loop(P, Q)
    ca.prof.enable(True); ca.prof.reset()
    t0 = time.perf_counter()
    for _ in range(10): ca.multi_miller_loop(P, Q)
    dt = (time.perf_counter() - t0) / 10 * 1e3
    st = ca.prof.read(); ca.prof.enable(False)
    print("pairs=%5d  %.3f ms (%.0f pairs/s) | %s" % (n, dt, n / dt * 1e3, " ".join("%s=%.3f" % (k, v[0] / max(1, v[1])) for k, v in st.items())), flush=True)
if os.environ.get("INFLIGHT"):          # K callers, each a 1024-pair loop
    from concurrent.futures import ThreadPoolExecutor
    n = 1024
    with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
        P, _ = t1.multiply_many(B.seeded_scalars(5, n)); Q, _ = t2.multiply_many(B.seeded_scalars(6, n))
    ref = ca.multi_miller_loop(P, Q)
    for K in [int(x) for x in os.environ["INFLIGHT"].split(",")]:
        with ThreadPoolExecutor(K) as ex:
            list(ex.map(lambda _: ca.multi_miller_loop(P, Q), range(4 * K)))
            t0 = time.perf_counter(); res = list(ex.map(lambda _: ca.multi_miller_loop(P, Q), range(60))); dt = (time.perf_counter() - t0) / 60 * 1e3
        assert all((r == ref).all() for r in res)
        print("%d calls in flight: %.3f ms per 1024-pair loop (%.0f pairs/s)" % (K, dt, n / dt * 1e3), flush=True)
