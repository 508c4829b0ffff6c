# tools/dev/verify_prof.py — where the time of the batch verifiers goes (cProfile, cumulative): verify_proofs_batch (the reference's
# RandomizedPairingChecker structure), verify_proofs_batch_merged, SnarkPack aggregate verification
import sys, os, cProfile, pstats, io, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R + "/oracle", R + "/tests", R]
import bench as B
import crypto_amd as ca
from crypto_amd import fixed_base as FB, legogroth16 as LGv
import oracle_c as O
ca.init(0)
R_MOD = B.R_MOD
gen1, gen2 = O.G1.generator().reshape(1, 12), O.G2.generator().reshape(1, 24)
nv = 1024
ints = lambda seed, k: [int(x[0]) | (int(x[1]) << 64) | (int(x[2]) << 128) | (int(x[3]) << 192) for x in B.seeded_scalars(seed, k)]
al, be, ga, de, g0, g1x = ints(0x5EED0020, 6)
av, bv, dv, xv = ints(0x5EED0021, nv), ints(0x5EED0022, nv), ints(0x5EED0023, nv), ints(0x5EED0024, nv)
dinv = pow(de, R_MOD - 2, R_MOD)
cv = [((a * b - al * be - (g0 + x * g1x + d) * ga) * dinv) % R_MOD for a, b, d, x in zip(av, bv, dv, xv)]
lim = lambda vals: np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
    A_, _ = t1.multiply_many(lim(av)); C_, _ = t1.multiply_many(lim(cv)); D_, _ = t1.multiply_many(lim(dv)); K_, _ = t1.multiply_many(lim([al, g0, g1x, 1]))
    B_, _ = t2.multiply_many(lim(bv)); V_, _ = t2.multiply_many(lim([be, ga, de]))
vkv = LGv.VerifyingKey(K_[0], V_[0], V_[1], V_[2], K_[1:3], K_[3], 0)
pvkv = LGv.prepare_verifying_key(vkv)
proofs_v = [{"a": A_[i], "b": B_[i], "c": C_[i], "d": D_[i]} for i in range(nv)]
pubs_v = [lim([x]) for x in xv]
for name, fn in (("verify_proofs_batch", lambda: LGv.verify_proofs_batch(pvkv, proofs_v, pubs_v, 0x5EED0028)),
                 ("verify_proofs_batch_merged", lambda: LGv.verify_proofs_batch_merged(pvkv, proofs_v, pubs_v, 0x5EED0029))):
    assert fn(); fn()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(3): fn()
    pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18)
    print("=====", name); print("\n".join(s.getvalue().splitlines()[:40]))
