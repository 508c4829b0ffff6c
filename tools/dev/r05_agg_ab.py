# Development helper (GPU box): the library aggregator of 1024 LegoGroth16-shaped proofs with the native Merlin transcript, eight timed calls (min / median);
# DGPU_LIB=<another build> repeats it on that build (same-box A/B)
import sys, os, time, numpy as np
sys.path[:0] = ["/root/repo/oracle", "/root/repo/tests", "/root/repo"]
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import aggregation as AG
from crypto_amd.aggregation import native as NA
from crypto_amd.fixed_base import WindowTable
ca.init(0)
n = 1024
rng = np.random.default_rng(3)
ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % (U.R - 1) + 1 for _ in range(k)]
def fixed(curve, g, ks):
    with WindowTable(curve, g, len(ks)) as t:
        return t.multiply_many(ks)[0]
g, h = O.G1.generator(), O.G2.generator()
A, Cc, D = fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)); B = fixed(ca.G2, h, ints(n))
proofs = [{"a": A[i], "b": B[i], "c": Cc[i], "d": D[i]} for i in range(n)]
pk, vsrs = AG.setup_fake_srs(ints(1)[0], ints(1)[0], n, g, h).specialize(n)
f = lambda: NA.aggregate_proofs_words(pk, AG.NativeMerlinTranscript(b"t"), proofs, with_d=True)
f(); f()
ts = []
for _ in range(8):
    t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
print(os.environ.get("DGPU_LIB", "default"), "aggregate ms: min %.2f median %.2f" % (min(ts), sorted(ts)[4]))
