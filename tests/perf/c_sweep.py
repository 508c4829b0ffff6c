import sys, time
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/oracle")
import numpy as np, crypto_amd as ca, oracle_c as O
from crypto_amd._native import lib
from crypto_amd import fixed_base as FB
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
import os
CV, GEN = (ca.G2, O.G2.generator()) if os.environ.get("G2") else (ca.G1, O.G1.generator())
for lg in [int(x) for x in os.environ.get("LOGS", "10,12,14,16,18,20").split(",")]:
    n=1<<lg
    with FB.WindowTable(CV, GEN) as t: db=t.multiply_many_to_bases(O.rand_scalars(3,n))
    ds=ca.DeviceScalars(O.rand_scalars(4,n))
    res=[]
    for c in (0, 8, 9, 10, 11, 12, 13, 14, 15, 16):
        lib().dgpu_set_window_bits(c)
        for _ in range(6): db.msm_resident(ds)
        t0=time.time()
        for _ in range(10): db.msm_resident(ds)
        res.append((c, round((time.time()-t0)/10*1e3,3)))
    lib().dgpu_set_window_bits(0)
    print("2^%d"%lg, res, flush=True)
    db.free(); ds.free()
