# tools/dev/fold_time.py — the folding step: the chain kernels (dgpu_g*_mul_add_batch, one scalar) against prepare + apply (fold_kernels.hip.h)
import sys, os, time, ctypes as C, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd._native import lib
from crypto_amd.aggregation import ops
ca.init(0)
p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
def timed(f, k=10):
    f(); f(); t0 = time.perf_counter()
    for _ in range(k): f()
    return (time.perf_counter() - t0) / k * 1e3
c = O.limbs_to_int(O.rand_scalars(5, 1)[0]); cl = np.array([(c >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(4)], dtype=np.uint64)
for g, grp, cv, sizes in (("g1", O.G1, ca.G1, (5, 160, 2560)), ("g2", O.G2, ca.G2, (3, 96, 1536))):
    for n in sizes:
        P = U.seq_bases(grp, n, 900 + n, threads=32)[0]; A = U.seq_bases(grp, n, 950 + n, threads=32)[0]
        t_old = timed(lambda: ops.mul_add(cv, P, c, A))
        h = C.c_uint64(0)
        def prep():
            if h.value: lib().dgpu_fold_free(h.value)
            assert getattr(lib(), "dgpu_%s_fold_prepare" % g)(p(P), n, C.byref(h)) == 0
        t_prep = timed(prep)
        out = np.zeros_like(P); inf = np.zeros(n, np.uint8)
        t_app = timed(lambda: getattr(lib(), "dgpu_%s_fold_apply" % g)(h.value, p(cl), p(A), p(out), p(inf)))
        assert (out == ops.mul_add(cv, P, c, A)).all()
        lib().dgpu_fold_free(h.value)
        print("%s n = %4d: chain kernel %.3f ms | prepare %.3f ms (hidden behind the round's pairings) + apply %.3f ms" % (g, n, t_old, t_prep, t_app), flush=True)
