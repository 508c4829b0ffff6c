"""Shared-sort MSM timing (development helper, GPU box): dgpu_msm_*_resident against dgpu_scalars_sort + dgpu_msm_*_sorted at 2^20 terms, per-stage device times."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, crypto_amd as ca, oracle_c as O
from crypto_amd import fixed_base as FB
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
n = 1 << int(os.environ.get("LOG2N", "20"))
ds = ca.DeviceScalars(O.rand_scalars(4, n))
for cv, G in ((ca.G1, O.G1), (ca.G2, O.G2)):
    with FB.WindowTable(cv, G.generator()) as t: db = t.multiply_many_to_bases(O.rand_scalars(3, n))
    db.precompute()
    ref = db.msm_resident(ds)
    srt = ca.SortedScalars(db, ds, n)
    assert (db.msm_sorted(srt) == ref).all()
    for name, fn in (("resident", lambda: db.msm_resident(ds)), ("sorted (list ready)", lambda: db.msm_sorted(srt)), ("sort only", lambda: ca.SortedScalars(db, ds, n).free())):
        for _ in range(8): fn()
        ca.prof.enable(True); ca.prof.reset()
        t0 = time.time()
        for _ in range(10): fn()
        dt = (time.time() - t0) / 10 * 1e3
        print(cv is ca.G1 and "G1" or "G2", name, "%.3f ms" % dt, {k: round(v[0] / max(1, v[1]), 3) for k, v in ca.prof.read().items()}, flush=True)
        ca.prof.enable(False)
    srt.free(); db.free()
