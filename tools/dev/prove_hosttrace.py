"""Development helper: host-side timestamps of the jobs of one LegoGroth16 proof (run after tests/perf/prove_perf.py's setup)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "perf"))
import runpy
ns = runpy.run_path(os.path.join(ROOT, "tests", "perf", "prove_perf.py"))
import crypto_amd as ca
from crypto_amd import msm as M, qap
LOG = []
T0 = [0.0]
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); LOG.append((label, (t - T0[0]) * 1e3, (time.perf_counter() - T0[0]) * 1e3, threading.get_ident() % 1000)); return r
    setattr(obj, name, g)
pk, circ = ns["pk"], ns["circ"]
for q, lab in ((pk.h_query, "h"), (pk.a_query, "a"), (pk.b_g1_query, "b1"), (pk.b_g2_query, "b2"), (pk.l_query, "l")):
    wrap(q, "msm_resident", "msm_" + lab)
wrap(circ, "witness_map", "witness_map")
for _ in range(3):
    LOG.clear(); T0[0] = time.perf_counter(); ns["ovl"](); tot = (time.perf_counter() - T0[0]) * 1e3
print("total %.2f ms" % tot)
for e in sorted(LOG, key=lambda e: e[1]):
    print("%-12s %6.2f -> %6.2f  (thread %d)" % e)
