import sys, time
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/oracle")
import numpy as np, crypto_amd as ca, oracle_c as O
from crypto_amd import pairing_check as pc
from crypto_amd.aggregation import ops
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
for n in (16, 1024, 16384):
    P = O.G1.gen_seq(k0, d, n, threads=16); Q = O.G2.gen_seq(d, k0, min(n, 2048), threads=16)
    sc = O.rand_scalars(5, n)
    pc.g1_scale_each(P, sc); ops.mul_add(ca.G1, P, 12345678901234567890123456789, P); ops.mul_add(ca.G2, Q, 12345678901234567890123456789, Q)
    ca.prof.enable(True); ca.prof.reset()
    for _ in range(3):
        pc.g1_scale_each(P, sc); ops.mul_add(ca.G1, P, 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000000, P)
    st = ca.prof.read(); ca.prof.enable(False)
    print(n, {k: round(v[0]/v[1], 3) for k, v in st.items()})
