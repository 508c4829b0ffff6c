# tools/dev/agg_host_trace.py — development helper: one 1024-proof aggregation with the compiled transcript, wall time of the call (the library's own phase marks go
# to stderr when it is built with them and DGPU_AGG_TRACE is set)
import sys, os, time, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import aggregation as AG
from crypto_amd.aggregation import native as NA
from crypto_amd.fixed_base import WindowTable
ca.init(0)
n = int(os.environ.get("N", "1024"))
rng = np.random.default_rng(3)
ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % (U.R - 1) + 1 for _ in range(k)]
def fixed(curve, g, ks):
    with WindowTable(curve, g, len(ks)) as t:
        return t.multiply_many(ks)[0]
g, h = O.G1.generator(), O.G2.generator()
A, Cc, D = fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)); B = fixed(ca.G2, h, ints(n))
proofs = [{"a": A[i], "b": B[i], "c": Cc[i], "d": D[i]} for i in range(n)]
pk, vsrs = AG.setup_fake_srs(ints(1)[0], ints(1)[0], n, g, h).specialize(n)
NT = AG.NativeMerlinTranscript
os.environ.pop("DGPU_AGG_TRACE", None)
for _ in range(6): NA.aggregate_proofs_words(pk, NT(b"t"), proofs, with_d=True)
ts = []
for _ in range(8):
    t0 = time.perf_counter(); NA.aggregate_proofs_words(pk, NT(b"t"), proofs, with_d=True); ts.append((time.perf_counter() - t0) * 1e3)
print("aggregate: min %.2f median %.2f ms" % (min(ts), sorted(ts)[len(ts) // 2]))
os.environ["DGPU_AGG_TRACE"] = "1"
sys.stderr.flush()
NA.aggregate_proofs_words(pk, NT(b"t"), proofs, with_d=True)
