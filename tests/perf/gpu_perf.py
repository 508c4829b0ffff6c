"""Stage timings on the GPU box (development helper): G1/G2 MSM stages, 1024-pair Miller loop, CPU oracle alongside."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import crypto_amd as ca
from crypto_amd._native import lib
import oracle_c as O

ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
what = os.environ.get("WHAT", "g1,g2,ml").split(",")

def stages(fn, K=5):
    fn()
    ca.prof.enable(True); ca.prof.reset()
    t0 = time.time()
    for _ in range(K): fn()
    dt = (time.time() - t0) / K
    st = ca.prof.read(); ca.prof.enable(False)
    return dt, " ".join("%s=%.3f" % (k.split(".")[1], v[0] / v[1]) for k, v in st.items())

k0 = O.rand_scalars(1, 1)[0]; d = O.rand_scalars(2, 1)[0]
if "g1" in what:
    for lg in [int(x) for x in os.environ.get("G1_LOGS", "16,18,20,22").split(",")]:
        n = 1 << lg
        bases = O.G1.gen_seq(k0, d, n, threads=64); sc = O.rand_scalars(3, n)
        db = ca.DeviceBases(ca.G1, bases); ds = ca.DeviceScalars(sc)
        dt, s = stages(lambda: db.msm_resident(ds))
        t0 = time.time(); ca.msm_bigint(ca.G1, bases, sc); one = time.time() - t0
        print("G1 n=2^%d resident %.3f ms (one-shot %.1f ms) | %s" % (lg, dt * 1e3, one * 1e3, s), flush=True)
        db.free(); ds.free()
if "g2" in what:
    for lg in [int(x) for x in os.environ.get("G2_LOGS", "16,18,20").split(",")]:
        n = 1 << lg
        bases = O.G2.gen_seq(k0, d, n, threads=64); sc = O.rand_scalars(3, n)
        db = ca.DeviceBases(ca.G2, bases); ds = ca.DeviceScalars(sc)
        dt, s = stages(lambda: db.msm_resident(ds), K=3)
        print("G2 n=2^%d resident %.3f ms | %s" % (lg, dt * 1e3, s), flush=True)
        if lg <= 16:
            t0 = time.time(); O.G2.msm(bases, sc, threads=20); print("   CPU oracle G2 2^%d: %.3f s" % (lg, time.time() - t0))
        db.free(); ds.free()
if "ml" in what:
    g1, g2 = O.G1.generator(), O.G2.generator()
    n = 1024
    P = O.G1.gen_seq(k0, d, n, threads=16); Q = O.G2.gen_seq(d, k0, n, threads=16)
    f = ca.multi_miller_loop(P, Q)
    t0 = time.time(); ref = O.multi_miller_loop(P, Q, threads=64); tc = time.time() - t0
    print("ML 1024 pairs: GPU == oracle: %s; CPU oracle 64 thr %.3f s" % ((f == ref).all(), tc))
    dt, s = stages(lambda: ca.multi_miller_loop(P, Q))
    print("ML n=1024 %.3f ms (%.0f pairs/s) | %s" % (dt * 1e3, n / dt, s))
    t0 = time.time(); gt = ca.final_exponentiation(f); print("final_exp host %.3f ms" % ((time.time() - t0) * 1e3))
    for nn in [int(x) for x in os.environ.get("ML_NS", "").split(",") if x]:
        P = O.G1.gen_seq(k0, d, nn, threads=64); Q = O.G2.gen_seq(d, k0, nn, threads=64)
        dt, s = stages(lambda: ca.multi_miller_loop(P, Q), K=3)
        print("ML n=%d %.3f ms (%.0f pairs/s) | %s" % (nn, dt * 1e3, nn / dt, s), flush=True)
if "conc" in what:
    import threading
    n = 1 << 20
    bases = O.G1.gen_seq(k0, d, n, threads=64); sc = O.rand_scalars(3, n)
    db = ca.DeviceBases(ca.G1, bases); ds = ca.DeviceScalars(sc)
    ref = db.msm_resident(ds)
    for nthr in (1, 2, 3, 4):
        K = 24
        def work(k):
            for _ in range(k):
                r = db.msm_resident(ds)
                assert (r == ref).all()
        work(2)
        ths = [threading.Thread(target=work, args=(K // nthr,)) for _ in range(nthr)]
        t0 = time.time()
        for t in ths: t.start()
        for t in ths: t.join()
        dt = time.time() - t0
        print("G1 2^20 x%d in flight: %.3f ms per MSM (%.1f MSM/s)" % (nthr, dt / (K // nthr * nthr) * 1e3, (K // nthr * nthr) / dt), flush=True)
if "qap" in what:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lego_setup as LS
    from crypto_amd import qap
    for lg in [int(x) for x in os.environ.get("QAP_LOGS", "16,20").split(",")]:
        m = (1 << lg) - 3
        t0 = time.time()
        # nconstraints-shaped circuit built directly as arrays (python big ints would take minutes at 2^20)
        xs = np.zeros((m + 1, 4), np.uint64); xs[:, 0] = np.arange(1, m + 2, dtype=np.uint64)     # synthetic assignment (timing only)
        z = np.concatenate([LS.scalars([1, 5]), xs])
        idx = np.arange(m, dtype=np.uint32)
        one = np.zeros((1, 4), np.uint64); one[0, 0] = 1
        a_rp = np.arange(m + 2, dtype=np.uint64); a_cl = np.concatenate([2 + idx, [2 + m]]).astype(np.uint32); a_vl = np.repeat(one, m + 1, 0)
        b_rp = a_rp; b_cl = np.concatenate([2 + idx, [0]]).astype(np.uint32); b_vl = a_vl
        c_rp = np.concatenate([2 * np.arange(m + 1, dtype=np.uint64), [2 * m + 1]]).astype(np.uint64)
        c_cl = np.concatenate([np.stack([3 + idx, np.zeros(m, np.uint32)], 1).reshape(-1), [1]]).astype(np.uint32)
        c_vl = np.repeat(one, 2 * m + 1, 0)
        mats = [(a_rp, a_cl, a_vl), (b_rp, b_cl, b_vl), (c_rp, c_cl, c_vl)]
        dt, s = stages(lambda: qap.witness_map(*mats, z, 2, m + 1, to_host=False, resident=False) if False else qap.witness_map(*mats, z, 2, m + 1), K=3)
        print("witness map m=2^%d-2 (D=2^%d): %.2f ms | %s" % (lg, lg, dt * 1e3, s), flush=True)
if "fb" in what:
    from crypto_amd import fixed_base as fb
    for cv, grp, lgs in ((ca.G1, O.G1, (10, 16, 20)), (ca.G2, O.G2, (10, 16, 18))):
        t0 = time.time(); tab = fb.WindowTable(cv, grp.generator()); tb = time.time() - t0
        t0 = time.time(); tab2 = fb.WindowTable(cv, grp.generator()); tb2 = time.time() - t0; tab2.free()
        for lg in lgs:
            n = 1 << lg; sc = O.rand_scalars(5, n)
            dt, s = stages(lambda: tab.multiply_many(sc), K=3)
            print("fixed-base %s n=2^%d %.3f ms (%.2f M mul/s) | %s | table build %.2f / %.2f ms" % (cv.tag, lg, dt * 1e3, n / dt / 1e6, s, tb * 1e3, tb2 * 1e3), flush=True)
        tab.free()
if "dist" in what:
    # SURVEY 8(d) config 2 secondary scalar distributions, n = 2^20 resident
    n = 1 << 20
    bases = O.G1.gen_seq(k0, d, n, threads=64)
    rng = np.random.default_rng(5)
    uni = O.rand_scalars(31, n)
    dists = {"uniform": uni}
    eq = np.tile(uni[:1], (n, 1)); dists["all-equal"] = eq
    s16 = np.zeros((n, 4), np.uint64); s16[:, 0] = rng.integers(0, 1 << 16, n, dtype=np.uint64); dists["16-bit"] = s16
    zo = uni.copy(); kind = rng.integers(0, 4, n); zo[kind <= 1] = 0; zo[kind == 1, 0] = 1; dists["50% zeros/ones"] = zo
    inf = (rng.integers(0, 100, n) == 0).astype(np.uint8)
    for name, sc in dists.items():
        db = ca.DeviceBases(ca.G1, bases); ds = ca.DeviceScalars(sc)
        dt, s = stages(lambda: db.msm_resident(ds))
        print("G1 2^20 %-16s %.3f ms | %s" % (name, dt * 1e3, s), flush=True)
        db.free(); ds.free()
    db = ca.DeviceBases(ca.G1, bases, inf); ds = ca.DeviceScalars(uni)
    dt, s = stages(lambda: db.msm_resident(ds))
    print("G1 2^20 %-16s %.3f ms | %s" % ("1% identity bases", dt * 1e3, s), flush=True)
if "pc" in what:
    # RandomizedPairingChecker, lazy: N equations e(a_i, b_i) == t_i batched into one Miller loop + one final exponentiation
    from crypto_amd import pairing_check as pcm
    for N in (16, 512):
        a = O.G1.gen_seq(k0, d, N, threads=16); b = O.G2.gen_seq(d, k0, N, threads=16)
        ts = [ca.multi_pairing(a[i:i + 1], b[i:i + 1]) for i in range(min(N, 16))]
        # targets must match their equations for the check to hold: use the first 16 equations repeated
        aa = np.concatenate([a[:16]] * (N // 16)); bb = np.concatenate([b[:16]] * (N // 16))
        def run2():
            ch = pcm.RandomizedPairingChecker(0x1234567890ABCDEF1234567890ABCDEF, True)
            for i in range(N):
                ch.add_sources_and_target(aa[i], bb[i], ts[i % 16])
            return ch.verify()
        assert run2()
        t0 = time.time()
        for _ in range(3): assert run2()
        print("RandomizedPairingChecker lazy, %d equations: %.2f ms per batch" % (N, (time.time() - t0) / 3 * 1e3), flush=True)
