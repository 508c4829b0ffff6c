# tools/dev/host_time.py — host-side pieces on this box: final exponentiation, one Fp12 product, the 1024-pair Miller loop and pairing
import sys, os, time, numpy as np, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
L = lib(); p = lambda a: a.ctypes.data_as(C.c_void_p)
f = O.multi_miller_loop(O.G1.generator().reshape(1, 12), O.G2.generator().reshape(1, 24)); out = np.zeros(72, np.uint64)
t0 = time.perf_counter()
for _ in range(300): L.dgpu_final_exponentiation(p(f), p(out))
print("final exponentiation %.3f ms" % ((time.perf_counter() - t0) / 300 * 1e3))
o = np.zeros(72, np.uint64); t0 = time.perf_counter()
for _ in range(3000): L.dgpu_fp12_mul(p(f), p(out), p(o))
print("Fp12 product through the ABI %.2f us" % ((time.perf_counter() - t0) / 3000 * 1e6))
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
ps = O.G1.gen_seq(k0, d, 1024, threads=32); qs = O.G2.gen_seq(d, k0, 1024, threads=32)
for name, fn in (("multi_miller_loop", lambda: ca.multi_miller_loop(ps, qs)), ("multi_pairing", lambda: ca.multi_pairing(ps, qs))):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    print("1024-pair %s %.3f ms" % (name, (time.perf_counter() - t0) / 20 * 1e3))
