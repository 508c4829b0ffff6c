# tools/dev/agg_time.py — SnarkPack aggregation of n LegoGroth16-shaped proofs: the library's aggregator against the Python statement above the ABI
import sys, os, time, numpy as np
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import aggregation as AG
from crypto_amd.aggregation import native as NA, legogroth16 as AL
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
def timed(f, k=3):
    f(); t0 = time.perf_counter()
    for _ in range(k): r = f()
    return (time.perf_counter() - t0) / k * 1e3, r
tn, w = timed(lambda: NA.aggregate_proofs_words(pk, AG.MerlinTranscript(b"t"), proofs, with_d=True))
tp, py = timed(lambda: AL.aggregate_proofs(pk, AG.MerlinTranscript(b"t"), proofs))
print("n = %d: aggregate library %.2f ms, Python above the ABI %.2f ms, identical: %s" % (n, tn, tp, bool((NA.proof_to_words(py) == w).all())))
