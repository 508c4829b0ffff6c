# tools/dev/ml_repeat.py — the same Miller loops over and over from one and from six host threads (sizes 1 .. 4096, the verifier's mixed call), every result compared
# with the first: a race between the waves of k_miller_lines_ws / k_line_products3 would show as a result that differs from run to run
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT + "/oracle", ROOT]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import oracle_c as O, crypto_amd as ca
from crypto_amd import pairing
ca.init(0)
k0 = O.rand_scalars(41, 1)[0]; d = O.rand_scalars(42, 1)[0]
N = 4096
ps = O.G1.gen_seq(k0, d, N, threads=32); qs = O.G2.gen_seq(d, k0, N, threads=32)
sizes = [1, 3, 5, 16, 17, 33, 100, 255, 256, 1000, 1024, 1025, 2048, 4096]
ref = {n: ca.multi_miller_loop(ps[:n], qs[:n]) for n in sizes}
pc = pairing.G2Prepared.from_affine(qs[:3]); refv = pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]])
secs = float(os.environ.get("SECONDS", "60")); bad = 0; calls = 0
def one(k):
    n = sizes[k % len(sizes)]
    if k % 5 == 4: return (pairing.multi_miller_loop(ps[:3], [qs[:1], pc[1:]]) == refv).all()
    return (ca.multi_miller_loop(ps[:n], qs[:n]) == ref[n]).all()
t0 = time.time(); k = 0
while time.time() - t0 < secs / 2:
    bad += 0 if one(k) else 1; k += 1; calls += 1
with ThreadPoolExecutor(6) as ex:
    while time.time() - t0 < secs:
        r = list(ex.map(one, range(k, k + 60))); bad += r.count(False); k += 60; calls += 60
print("ml_repeat: %d calls in %.0f s (half of the time one at a time, half six in flight), %d results differ from the first" % (calls, secs, bad))
