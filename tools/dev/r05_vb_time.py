"""Development helper (GPU box): dgpu_legogroth16_verify_batch on 1024 proofs of one key, per-call times (12 warm-up calls, then 40)"""
import sys, os; R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R_, R_ + "/oracle", R_ + "/tests"]
import time, numpy as np, oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import legogroth16 as LG, fixed_base as FB
ca.init(0); n = int(os.environ.get("N", "1024")); R = U.R
rng = np.random.default_rng(50 + n)
ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1 for _ in range(k)]
al, be, ga, de, g0, g1x = ints(6)
av, bv, dv, xv = ints(n), ints(n), ints(n), ints(n)
dinv = pow(de, R - 2, R)
cv = [((a * b - al * be - (g0 + x * g1x + d) * ga) * dinv) % R for a, b, d, x in zip(av, bv, dv, xv)]
lim = lambda vals: np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
with FB.WindowTable(ca.G2, O.G2.generator()) as t2, FB.WindowTable(ca.G1, O.G1.generator()) as t1:
    A_, _ = t1.multiply_many(lim(av)); C_, _ = t1.multiply_many(lim(cv)); D_, _ = t1.multiply_many(lim(dv)); K_, _ = t1.multiply_many(lim([al, g0, g1x, 1]))
    B_, _ = t2.multiply_many(lim(bv)); V_, _ = t2.multiply_many(lim([be, ga, de]))
vk = LG.VerifyingKey(K_[0], V_[0], V_[1], V_[2], K_[1:3], K_[3], 0)
pvk = LG.prepare_verifying_key(vk)
proofs = [{"a": A_[i], "b": B_[i], "c": C_[i], "d": D_[i]} for i in range(n)]
pubs = [lim([x]) for x in xv]
packed = LG.pack_proofs(proofs, pubs)
f = lambda: LG.verify_proofs_batch_abi(pvk, None, None, 0x5EED0029, packed=packed)
assert f()
for _ in range(12): f()
ts = []
for _ in range(40):
    t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
print("verify_batch n=%d: min %.3f median %.3f mean %.3f max %.3f ms" % (n, min(ts), sorted(ts)[20], sum(ts) / 40, max(ts)))
print(" ".join("%.2f" % t for t in ts))
