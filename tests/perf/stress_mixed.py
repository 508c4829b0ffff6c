"""One-off stress (GPU box): SECONDS of mixed calls from THREADS host threads — resident MSMs on a table and on plain bases, one-shot strided MSMs through the
resident-bases cache, 64- and 1024-pair Miller loops, a segmented multi-pairing, final exponentiations, a witness-map-free prover-sized G2 MSM — every
result compared with the one the same call gave single-threaded at start-up.  Usage: SECONDS=60 THREADS=16 python tests/perf/stress_mixed.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch  # noqa: F401
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import pairing
ca.init(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "7")))
n1, n2 = 1 << 17, 1 << 14
b1, _, _ = U.seq_bases(O.G1, n1, 21); b2, _, _ = U.seq_bases(O.G2, n2, 22)
s1, s2 = O.rand_scalars(23, n1), O.rand_scalars(24, n2)
tab = ca.DeviceBases(ca.G1, b1); tab.precompute(17)
plain = ca.DeviceBases(ca.G1, b1[: 1 << 15]); g2b = ca.DeviceBases(ca.G2, b2)
d1, d2 = ca.DeviceScalars(s1), ca.DeviceScalars(s2)
st = ca.to_affine_structs(ca.G1, b1)
ca.bases_cache_clear(); ca.bases_cache(min_n=1 << 12)
P = b1[:1024]; Q = b2[:1024]
jobs = [(P[a:b], Q[a:b]) for a, b in ((0, 300), (300, 301), (301, 301), (301, 900))]
calls = {
    "table": lambda: tab.msm_resident(d1),
    "plain": lambda: plain.msm_resident(d1, n=1 << 15),
    "g2": lambda: g2b.msm_resident(d2),
    "strided": lambda: ca.msm_strided(ca.G1, st, s1),
    "strided_sub": lambda: ca.msm_strided(ca.G1, st[1:], s1[1:]),
    "ml64": lambda: ca.multi_miller_loop(P[:64], Q[:64]),
    "ml1024": lambda: ca.multi_miller_loop(P, Q),
    "segments": lambda: np.stack(pairing.multi_pairings(jobs)),
    "fe": lambda: ca.final_exponentiation(ca.multi_miller_loop(P[:3], Q[:3])),
}
ref = {k: f() for k, f in calls.items()}
for k in ("strided", "strided_sub"):            # second and third sighting: resident from here on
    for _ in range(2): assert (calls[k]() == ref[k]).all()
names = list(calls)
T = int(os.environ.get("THREADS", "16")); SECS = float(os.environ.get("SECONDS", "30"))
stop = time.time() + SECS
count = {k: 0 for k in names}; bad = []; lock = threading.Lock()
def run(seed):
    r = np.random.default_rng(seed)
    while time.time() < stop:
        k = names[int(r.integers(0, len(names)))]
        try:
            out = calls[k]()
            ok = (np.asarray(out) == np.asarray(ref[k])).all()
        except Exception as e:          # noqa: BLE001
            ok = False; out = repr(e)
        with lock:
            count[k] += 1
            if not ok: bad.append((k, str(out)[:80]))
th = [threading.Thread(target=run, args=(100 + i,)) for i in range(T)]
for t in th: t.start()
for t in th: t.join()
print("stress_mixed: %d threads, %.0f s, calls %s, mismatches %d %s" % (T, SECS, count, len(bad), bad[:3]))
print("cache:", ca.bases_cache_stats())
