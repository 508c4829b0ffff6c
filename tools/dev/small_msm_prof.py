import sys, os, time, numpy as np
R = "/root/repo"; sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0); lib().dgpu_set_min_gpu_n(1)
for n in (600, 4096):
    bases, _, _ = U.seq_bases(O.G1, n, 77, threads=32); sc = O.rand_scalars(78, n)
    db = ca.DeviceBases(ca.G1, bases); ds = ca.DeviceScalars(sc)
    for _ in range(6): db.msm_resident(ds)
