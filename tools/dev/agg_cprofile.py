"""cProfile of aggregate_proofs for n Groth16-shaped proofs (development helper): where the host wall time of a GIPA round goes."""
import os, sys, cProfile, pstats, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import crypto_amd as ca
from crypto_amd import aggregation as AG, fixed_base as FB
from crypto_amd.aggregation import ops
import oracle_c as O
R = ops.R_MOD
ca.init(0)
n = int(os.environ.get("N", "1024"))
rng = np.random.default_rng(1)
rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
with FB.WindowTable(ca.G1, O.G1.generator()) as t1, FB.WindowTable(ca.G2, O.G2.generator()) as t2:
    A, _ = t1.multiply_many([rnd() for _ in range(n)]); B, _ = t2.multiply_many([rnd() for _ in range(n)]); C, _ = t1.multiply_many([rnd() for _ in range(n)])
proofs = [{"a": A[i], "b": B[i], "c": C[i]} for i in range(n)]
srs = AG.setup_fake_srs(rnd(), rnd(), n, O.G1.generator(), O.G2.generator()); pk, vsrs = srs.specialize(n)
AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs)
t0 = time.time(); AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs); print("plain wall", round(time.time() - t0, 4))
_par = ops.parallel; log = []
def timed(thunks, host=False):
    t = time.time(); r = _par(thunks, host); log.append((len(thunks), host, round((time.time() - t) * 1e3, 2))); return r
ops.parallel = timed
from crypto_amd.aggregation import groth16 as _g; _g.ops.parallel = timed
t0 = time.time(); AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs); print("timed wall", round(time.time() - t0, 4)); print("parallel calls (thunks, host, ms):", log)
ops.parallel = _par
pr = cProfile.Profile(); pr.enable()
AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs)
pr.disable()
for key in ("cumulative", "tottime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35); print(s.getvalue()[:7000])
