# tools/dev/small_msm_loop.py — resident small MSMs in a loop (what rocprofv3 wraps to see the tree path's kernels): N terms, G1 (or G2=1)
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import oracle_c as O, util as U, crypto_amd as ca
ca.init(0)
n = int(os.environ.get("N", "600")); g2 = os.environ.get("G2") == "1"
curve, G = (ca.G2, O.G2) if g2 else (ca.G1, O.G1)
bases, _, _ = U.seq_bases(G, n, 77, threads=32); sc = O.rand_scalars(78, n)
db = ca.DeviceBases(curve, bases); ds = ca.DeviceScalars(sc)
for _ in range(5): db.msm_resident(ds)
t0 = time.perf_counter()
for _ in range(30): db.msm_resident(ds)
print("n = %d %s: %.3f ms per resident MSM" % (n, "G2" if g2 else "G1", (time.perf_counter() - t0) / 30 * 1e3))
