# tools/dev/ml_cuts_sweep.py — where the 68-step chain is cut into three launches (dgpu_set_miller_pipeline bits 8-21, development twin), with the line kernel as a wave per role:
# 1024-pair Miller loop and the verifier's call (1 affine + 2 prepared pairs), three alternations
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/oracle", ROOT]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
from crypto_amd._native import lib
ca.init(0)
_tw = ca.twin(); _tw.__enter__()
n = 1024
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
f = ca.multi_miller_loop(ps, qs); pc = pairing.G2Prepared.from_affine(qs[:3]); g = pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]])
def t(fn, k=40):
    for _ in range(4): fn()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    return (time.perf_counter() - t0) / k * 1e3
cuts = os.environ.get("CUTS", "40,17 44,22 36,14 46,26 40,12 44,17 36,17 48,30 42,20").split()
for rep in range(3):
    for c in cuts:
        a_, b_ = map(int, c.split(",")); assert lib().dgpu_set_miller_pipeline(31 | a_ << 8 | b_ << 16) == 0
        assert (ca.multi_miller_loop(ps, qs) == f).all() and (pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]]) == g).all()
        print("cuts %-6s  1024-pair loop %.3f ms   verifier's call %.3f ms" % (c, t(lambda: ca.multi_miller_loop(ps, qs)), t(lambda: pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]]))), flush=True)
