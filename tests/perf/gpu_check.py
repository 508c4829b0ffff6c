"""Quick on-GPU sanity + stage timing (development helper; the judged tests live in tests/)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import crypto_amd as ca
from crypto_amd._native import lib
import oracle_c as O
import ctypes as C

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
def p_(a): return a.ctypes.data_as(C.c_void_p)

ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
big = int(os.environ.get("BIG", "20"))
# 1. field selftest
a = O.fp_to_mont(O.rand_scalars(11, 600).reshape(-1, 6)[:300] & np.uint64(0x00ffffffffffffff))
b = O.fp_to_mont(O.rand_scalars(12, 600).reshape(-1, 6)[:300] & np.uint64(0x00ffffffffffffff))
out = np.zeros_like(a)
rc = lib().dgpu_selftest_fp_mul(p_(a), p_(b), len(a), p_(out)); assert rc == 0, rc
import ctypes
Lo = O.lib()
exp = np.zeros_like(a)
for i in range(len(a)):
    # oracle fp_mul via fp12-free path: use to_mont(from_mont(a)*from_mont(b)) through python ints
    x = O.limbs_to_int(O.fp_from_mont(a[i])); y = O.limbs_to_int(O.fp_from_mont(b[i]))
    P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    exp[i] = O.fp_to_mont(O.int_to_limbs(x * y % P, 6))
assert (out == exp).all(), "fp_mul selftest mismatch"
print("selftest fp_mul ok")

def check(curve, G, n, seed, threads=16, label=""):
    k0 = O.rand_scalars(seed, 1)[0]; d = O.rand_scalars(seed + 1, 1)[0]
    bases = G.gen_seq(k0, d, n, threads=threads) if n else np.zeros((0, G.AW), np.uint64)
    ss = O.rand_scalars(seed + 2, n)
    t0 = time.time(); got = ca.msm_bigint(curve, bases, ss); t1 = time.time()
    if n <= (1 << 16):
        ref = G.msm(bases, ss, threads=threads)
        ra, rinf = G.to_affine(ref)
    else:
        K0, D = O.limbs_to_int(k0), O.limbs_to_int(d)
        sv = [O.limbs_to_int(x) for x in ss]
        tot = (sum(sv) * K0 + sum(i * s for i, s in enumerate(sv)) * D) % R
        ra, rinf = G.to_affine(G.mul(G.generator(), O.int_to_limbs(tot, 4)))
    ga, ginf = G.to_affine(got)
    ok = (rinf == ginf) and (ga == ra).all()
    print("%s n=%d %s  (one-shot %.1f ms)" % (label, n, "OK" if ok else "MISMATCH", (t1 - t0) * 1e3))
    assert ok or os.environ.get('NOCHECK')
    return bases, ss

for n in (0, 1, 2, 31, 32, 33, 1000, 1 << 12, 1 << 16):
    check(ca.G1, O.G1, n, 100 + n, label="G1")
for n in (0, 1, 33, 1 << 10):
    check(ca.G2, O.G2, n, 500 + n, label="G2")
for c in (7, 9, 13, 15, 16):
    lib().dgpu_set_window_bits(c); check(ca.G1, O.G1, 3000, 900 + c, label="G1 c=%d" % c)
lib().dgpu_set_window_bits(0)
for ch in (16, 128):
    lib().dgpu_set_chunk(ch); check(ca.G1, O.G1, 5000, 950 + ch, label="G1 chunk=%d" % ch)
lib().dgpu_set_chunk(0)

n = 1 << big
t0 = time.time()
bases, ss = check(ca.G1, O.G1, n, 7777, threads=64, label="G1 big")
print("big check total %.1fs" % (time.time() - t0))
db = ca.DeviceBases(ca.G1, bases); ds = ca.DeviceScalars(ss)
ref = db.msm_resident(ds)
for c in [int(x) for x in os.environ.get("CS", "13,14,15,16,17").split(",")]:
    for ch in [int(x) for x in os.environ.get("CHS", "32,64").split(",")]:
        lib().dgpu_set_window_bits(c); lib().dgpu_set_chunk(ch)
        r = db.msm_resident(ds); assert (r == ref).all() or os.environ.get('NOCHECK')
        ca.prof.enable(True); ca.prof.reset()
        t0 = time.time(); K = 5
        for _ in range(K): db.msm_resident(ds)
        dt = (time.time() - t0) / K
        st = ca.prof.read(); ca.prof.enable(False)
        print("c=%d chunk=%d wall %.3f ms | " % (c, ch, dt * 1e3) + " ".join("%s=%.3f" % (k.replace("msm.", ""), v[0] / v[1]) for k, v in st.items()))
