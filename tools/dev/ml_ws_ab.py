# tools/dev/ml_ws_ab.py — the line kernel with a wave per role (dgpu_set_miller_pipeline bit 3) against sixteen lanes per pair: 1024-pair Miller loop,
# the verifier's call (1 affine + 2 prepared pairs), G2Prepared of 1024 points; one call at a time and six in flight, alternating on one box
import os, sys, time, numpy as np
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
from crypto_amd._native import lib
ca.init(0)
_twin = ca.twin(); _twin.__enter__()
n = 1024
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
lib().dgpu_set_miller_pipeline(7)
f = ca.multi_miller_loop(ps, qs); pc = pairing.G2Prepared.from_affine(qs[:3]); g = pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]])
def t(fn, k=30):
    for _ in range(3): fn()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    return (time.perf_counter() - t0) / k * 1e3
for mode in [int(m) for m in os.environ.get('MODES', '15,31,15,31,15,31').split(',')]:
    lib().dgpu_set_miller_pipeline(mode)
    assert (ca.multi_miller_loop(ps, qs) == f).all() and (pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]]) == g).all()
    one = t(lambda: ca.multi_miller_loop(ps, qs)); ver = t(lambda: pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]])); prep = t(lambda: pairing.G2Prepared.from_affine(qs), 10)
    with ThreadPoolExecutor(6) as ex:
        list(ex.map(lambda _: ca.multi_miller_loop(ps, qs), range(12)))
        t0 = time.perf_counter(); list(ex.map(lambda _: ca.multi_miller_loop(ps, qs), range(60))); six = (time.perf_counter() - t0) / 60 * 1e3
    print("mode %2d  1024-pair loop %.3f ms (six in flight %.3f per call)   verifier's call %.3f ms   G2Prepared x 1024 %.3f ms" % (mode, one, six, ver, prep), flush=True)
lib().dgpu_set_miller_pipeline(31)
