"""One-off randomized cross-check of the MSM entry points against the CPU oracle (sizes, duplicate / negated / identity bases, small and
zero scalars, handles with offsets, and the same handle converted to a precomputed-multiples table of a random window width).  Usage: ITER=300 python tests/perf/fuzz_msm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import crypto_amd as ca
import oracle_c as O
from crypto_amd.aggregation.ops import neg

ca.init(0)
rng = np.random.default_rng(int(os.environ.get("SEED", "1")))
k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
pool = {ca.G1: O.G1.gen_seq(k0, d, 6000, threads=32), ca.G2: O.G2.gen_seq(d, k0, 3000, threads=32)}
grp = {ca.G1: O.G1, ca.G2: O.G2}
O_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
bad = 0
for it in range(int(os.environ.get("ITER", "200"))):
    cv = ca.G1 if rng.integers(0, 3) else ca.G2
    G = grp[cv]
    n = int(rng.choice([1, 2, 3, 17, 64, 65, 300, 1000, 2500, 6000 if rng.integers(0, 2) else 2500])) + int(rng.integers(0, 40))
    n = min(n, len(pool[cv]))
    idx = rng.integers(0, max(1, n // int(rng.choice([1, 1, 4, 50]))), n)          # many duplicates in some runs
    bases = pool[cv][idx].copy()
    flip = rng.integers(0, 8, n) == 0
    for i in np.nonzero(flip)[0]:
        bases[i] = neg(cv, bases[i])
    inf = (rng.integers(0, 30, n) == 0).astype(np.uint8)
    bases[inf == 1] = 0
    sc = O.rand_scalars(1000 + it, n)
    kind = rng.integers(0, 5, n)
    sc[kind == 0] = 0
    sc[kind == 1, 1:] = 0
    sc[kind == 1, 0] &= np.uint64(0xFF)
    if rng.integers(0, 4) == 0:
        sc[:] = sc[0]                                                              # all-equal scalars: one hot bucket per window
    if rng.integers(0, 3) == 0:                                                   # Groth16-like: many ones / minus-ones / tiny values (hot buckets, sparse pair lists)
        m1 = rng.integers(0, 3, n) == 0
        sc[m1] = 0; sc[m1, 0] = 1
        m2 = rng.integers(0, 9, n) == 0
        sc[m2] = O.int_to_limbs(O_R - 1, 4)
    exp = G.to_affine(G.msm(bases, sc, inf, threads=8))
    got = G.to_affine(ca.msm_bigint(cv, bases, sc, is_inf=inf))
    off = int(rng.integers(0, min(n, 5)))
    db = ca.DeviceBases(cv, bases, inf)
    got2 = G.to_affine(db.msm_bigint(sc[: n - off], offset=off))
    exp2 = G.to_affine(G.msm(bases[off:], sc[: n - off], inf[off:], threads=8))
    # the same handle as a precomputed-multiples table of a random width: limb-identical to the plain handle path, sub-ranges included
    c = int(rng.choice([16, 17, 18, 19, 20, 21, 22]))
    plain_full = db.msm_bigint(sc)
    plain_off = db.msm_bigint(sc[: n - off], offset=off)
    db.precompute(c)
    tab_ok = (db.msm_bigint(sc) == plain_full).all() and (db.msm_bigint(sc[: n - off], offset=off) == plain_off).all()
    if n > 4:
        ds = ca.DeviceScalars(sc)
        m = int(rng.integers(1, n - 2)); bo = int(rng.integers(0, n - m)); so = int(rng.integers(0, n - m))
        e3 = G.to_affine(G.msm(bases[bo:bo + m], sc[so:so + m], inf[bo:bo + m], threads=8))
        g3 = G.to_affine(db.msm_resident(ds, n=m, base_offset=bo, scalar_offset=so))
        tab_ok = tab_ok and e3[1] == g3[1] and (e3[1] or (e3[0] == g3[0]).all())
        ds.free()
    db.free()
    ok = exp[1] == got[1] and (exp[1] or (exp[0] == got[0]).all()) and exp2[1] == got2[1] and (exp2[1] or (exp2[0] == got2[0]).all()) and bool(tab_ok)
    if not ok:
        bad += 1
        print("MISMATCH", it, cv.tag, n, off, c, flush=True)
print("fuzz_msm: %d iterations, %d mismatches" % (it + 1, bad))
sys.exit(1 if bad else 0)
