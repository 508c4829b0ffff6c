"""cProfile of verify_aggregate_proof for n Groth16-shaped proofs (development helper): host time of the verifier's stages."""
import os, sys, cProfile, pstats, io, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import crypto_amd as ca
from crypto_amd import legogroth16 as LG, aggregation as AG, fixed_base as FB
from crypto_amd.aggregation import ops
import oracle_c as O
R = ops.R_MOD
ca.init(0)
n = int(os.environ.get("N", "1024"))
rng = np.random.default_rng(1)
rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
alpha, beta, gamma, delta = rnd(), rnd(), rnd(), rnd()
ks = [rnd() for _ in range(5)]
with FB.WindowTable(ca.G1, O.G1.generator()) as t1, FB.WindowTable(ca.G2, O.G2.generator()) as t2:
    small1, _ = t1.multiply_many([alpha, 1] + ks); small2, _ = t2.multiply_many([beta, gamma, delta])
    vk = LG.VerifyingKey(small1[0], small2[0], small2[1], small2[2], small1[2:], small1[1], 0)
    inputs, av, bv, cv = [], [], [], []
    di = pow(delta, R - 2, R)
    for _ in range(n):
        x = [rnd() for _ in range(4)]; a, b = rnd(), rnd()
        s = (ks[0] + sum(xi * ki for xi, ki in zip(x, ks[1:]))) % R
        inputs.append(x); av.append(a); bv.append(b); cv.append((a * b - alpha * beta - s * gamma) * di % R)
    A, _ = t1.multiply_many(av); B, _ = t2.multiply_many(bv); C, _ = t1.multiply_many(cv)
proofs = [{"a": A[i], "b": B[i], "c": C[i]} for i in range(n)]
pk, vsrs = AG.setup_fake_srs(rnd(), rnd(), n, O.G1.generator(), O.G2.generator()).specialize(n)
agg = AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs)
ver = lambda: AG.verify_aggregate_proof(vsrs, {"vk": vk}, inputs, agg, rnd(), AG.MerlinTranscript(b"bench"))
ver(); t0 = time.time(); ver(); print("plain wall", round(time.time() - t0, 4))
ca.prof.enable(True); ca.prof.reset()
pr = cProfile.Profile(); pr.enable(); ver(); pr.disable()
print("device ms / calls:", {k: (round(v[0], 2), v[1]) for k, v in sorted(ca.prof.read().items(), key=lambda kv: -kv[1][0])})
for key in ("cumulative", "tottime"):
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22); print(s.getvalue()[:4500])
