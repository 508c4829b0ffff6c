# tools/dev/ml_lp3_limit.py — up to how many blocks a launch of the sparse products takes the three-wave form (dgpu_set_miller_pipeline bits 28-29, twin):
# prepared pairs (all 68 steps in one launch) and affine pairs at 1024 / 2048 / 4096 / 8192
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/oracle", ROOT]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
from crypto_amd._native import lib
ca.init(0)
_tw = ca.twin(); _tw.__enter__()
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
def t(fn, k=30):
    for _ in range(4): fn()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    return (time.perf_counter() - t0) / k * 1e3
for n in (1024, 2048, 4096, 8192):
    ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
    lib().dgpu_set_miller_pipeline(31)
    pc = pairing.G2Prepared.from_affine(qs); f = ca.multi_miller_loop(ps, qs)
    for rep in range(2):
        row = []
        for sh in (0, 1, 2, 3, -1):
            assert lib().dgpu_set_miller_pipeline((31 | sh << 28) if sh >= 0 else 15) == 0
            assert (pairing.multi_miller_loop(ps, [pc]) == f).all() and (ca.multi_miller_loop(ps, qs) == f).all()
            row.append("%s: %.3f / %.3f" % ("x%d" % (1 << sh) if sh >= 0 else "off", t(lambda: pairing.multi_miller_loop(ps, [pc])), t(lambda: ca.multi_miller_loop(ps, qs))))
        print("n = %d  prepared / affine ms per call by block limit   %s" % (n, "   ".join(row)), flush=True)
lib().dgpu_set_miller_pipeline(31)
