# tools/dev/top_quad.py — k_reduce_top with 1 / 4 members per point: same limbs, stage times
import sys, time, numpy as np
sys.path.insert(0, "oracle"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd._native import lib
from crypto_amd.msm import prof
ca.init(0)
for gname, n in (("G1", 1 << 20), ("G2", 1 << 18), ("G1", 70000), ("G2", 5000)):
    curve, G = (ca.G1, O.G1) if gname == "G1" else (ca.G2, O.G2)
    bases, _, _ = U.seq_bases(G, n, 77, threads=64)
    sc = O.rand_scalars(78, n)
    plain = ca.DeviceBases(curve, bases)
    tab = ca.DeviceBases(curve, bases).precompute(0)
    for name, hb in (("plain", plain), ("table", tab)):
        out = {}
        for lanes in (1, 4, 1, 4):
            assert lib().dgpu_set_reduce_lanes(lanes) == 0
            r = hb.msm_bigint(sc)
            for _ in range(2): hb.msm_bigint(sc)
            t0 = time.perf_counter()
            for _ in range(10): hb.msm_bigint(sc)
            ms = (time.perf_counter() - t0) / 10 * 1e3
            prof.enable(True); prof.reset(); hb.msm_bigint(sc); hb.msm_bigint(sc); pr = prof.read(); prof.enable(False)
            out.setdefault(lanes, []).append((r, ms, pr.get("msm.reduce", (0, 1))))
        same = all((out[4][i][0] == out[1][0][0]).all() for i in range(2))
        print(gname, n, name, "same limbs" if same else "MISMATCH", " latency ms 1 lane:", ["%.3f" % v[1] for v in out[1]], " 4 lanes:", ["%.3f" % v[1] for v in out[4]],
              " reduce stage:", ["%.3f" % (v[2][0] / v[2][1]) for v in out[1]], ["%.3f" % (v[2][0] / v[2][1]) for v in out[4]], flush=True)
lib().dgpu_set_reduce_lanes(4)
