# tools/dev/ml_tail_sweep.py — slice length of the LAST piece's sparse products (dgpu_set_miller_pipeline bits 24-27, development twin) at 1024 / 256 / 4096 pairs, three alternations
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/oracle", ROOT]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import oracle_c as O, crypto_amd as ca
from crypto_amd._native import lib
ca.init(0)
_tw = ca.twin(); _tw.__enter__()
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
def t(fn, k=40):
    for _ in range(4): fn()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    return (time.perf_counter() - t0) / k * 1e3
for n in (1024, 256, 4096):
    ps = O.G1.gen_seq(k0, d, n, threads=32); qs = O.G2.gen_seq(d, k0, n, threads=32)
    lib().dgpu_set_miller_pipeline(31)
    f = ca.multi_miller_loop(ps, qs)
    for rep in range(3):
        row = []
        for v in ("", "1", "2", "4", "8"):
            assert lib().dgpu_set_miller_pipeline(31 | (int(v) if v else 0) << 24) == 0
            assert (ca.multi_miller_loop(ps, qs) == f).all()
            row.append("%s: %.3f" % (v or "auto", t(lambda: ca.multi_miller_loop(ps, qs))))
        print("n = %d  tail slice  %s  (ms per call)" % (n, "   ".join(row)), flush=True)
