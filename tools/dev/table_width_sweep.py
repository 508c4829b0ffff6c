"""G2 table width sweep at 2^20 terms (development helper): latency and 4-in-flight rate per window width of the precomputed table."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, crypto_amd as ca, oracle_c as O
from crypto_amd import fixed_base as FB
from concurrent.futures import ThreadPoolExecutor
ca.init(0)
n = 1 << int(os.environ.get("LOG2N", "20"))
CV, GEN = (ca.G1, O.G1.generator()) if os.environ.get("G1") else (ca.G2, O.G2.generator())
ds = ca.DeviceScalars(O.rand_scalars(4, n))
for c in [int(x) for x in os.environ.get("CS", "18,19,20,21").split(",")]:
    with FB.WindowTable(CV, GEN) as t: db = t.multiply_many_to_bases(O.rand_scalars(3, n))
    db.precompute(c)
    ref = db.msm_resident(ds)
    for _ in range(4): db.msm_resident(ds)
    t0 = time.time()
    for _ in range(10): db.msm_resident(ds)
    lat = (time.time() - t0) / 10 * 1e3
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda _: db.msm_resident(ds), range(8)))
        t0 = time.time(); rs = list(ex.map(lambda _: db.msm_resident(ds), range(24))); thr = (time.time() - t0) / 24 * 1e3
    assert all((r == ref).all() for r in rs)
    print("c=%d latency %.3f ms, 4 in flight %.3f ms per MSM" % (c, lat, thr), flush=True)
    db.free()
