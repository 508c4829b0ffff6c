"""Timing driver (GPU box): LegoGroth16 create_proof at 2^LOG2N constraints on a synthetic key of precomputed tables — witness map first
then the MSMs (prover.rs order) vs the witness map overlapped with the MSMs that do not need h."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import fixed_base as FB, qap, serde, legogroth16 as LG
import bench as B
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
if os.environ.get("REDUCE_SHIFT"):
    from crypto_amd._native import lib as _lib
    assert _lib().dgpu_set_reduce_shift(int(os.environ["REDUCE_SHIFT"])) == 0
log2n = int(os.environ.get("LOG2N", "20")); n = 1 << log2n
table = os.environ.get("TABLE", "1") == "1"
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED)); gen2, _ = serde.deserialize(ca.G2, bytes.fromhex(B.G2_GEN_COMPRESSED))
m = n - 3
idx = np.arange(m, dtype=np.uint32)
one = np.zeros((1, 4), np.uint64); one[0, 0] = 1
a_rp = np.arange(m + 2, dtype=np.uint64); a_cl = np.concatenate([2 + idx, [2 + m]]).astype(np.uint32); a_vl = np.repeat(one, m + 1, 0)
b_cl = np.concatenate([2 + idx, [0]]).astype(np.uint32)
c_rp = np.concatenate([2 * np.arange(m + 1, dtype=np.uint64), [2 * m + 1]]).astype(np.uint64)
c_cl = np.concatenate([np.stack([3 + idx, np.zeros(m, np.uint32)], 1).reshape(-1), [1]]).astype(np.uint32)
circ = qap.DeviceR1cs((a_rp, a_cl, a_vl), (a_rp, b_cl, a_vl), (c_rp, c_cl, np.repeat(one, 2 * m + 1, 0)), m + 3, 2, m + 1)
z = B.seeded_scalars(7, m + 3)
rng = np.random.Generator(np.random.PCG64(8)); kd = rng.integers(0, 4, m + 3)
z[kd <= 1] = 0; z[kd == 1, 0] = rng.integers(0, 2, int((kd == 1).sum()), dtype=np.uint64); mk = kd == 2; z[mk, 1:] = 0; z[mk, 0] &= np.uint64(0xFFFF)
cw, V = 2, m + 2
wc = int(os.environ.get("WITNESS_C", "0")); hc = int(os.environ.get("H_C", "0"))     # window widths of the witness-query / h-query tables (0 = automatic)
pre = (lambda d, c=0: d.precompute(c)) if table else (lambda d, c=0: d)
with FB.WindowTable(ca.G2, gen2[0]) as t2, FB.WindowTable(ca.G1, gen1[0]) as t1:
    small1, _ = t1.multiply_many(B.seeded_scalars(10, 8 + 2 + cw)); small2, _ = t2.multiply_many(B.seeded_scalars(11, 4))
    qa = pre(t1.multiply_many_to_bases(B.seeded_scalars(12, V + 1)), wc); qb1 = pre(t1.multiply_many_to_bases(B.seeded_scalars(13, V + 1)), wc)
    qb2 = pre(t2.multiply_many_to_bases(B.seeded_scalars(14, V + 1)), wc); qh = pre(t1.multiply_many_to_bases(B.seeded_scalars(15, n - 1)), hc)
    ql = pre(t1.multiply_many_to_bases(B.seeded_scalars(16, m + 1 - cw)), wc)
vk = LG.VerifyingKey(small1[0], small2[0], small2[1], small2[2], small1[8:8 + 2 + cw], small1[1], cw)
pk = LG.ProvingKey.from_device(vk, small1[2], small1[3], small1[4], small1[5], small1[6], small2[3], qa, qb1, qb2, qh, ql)

def seq():
    _, dh = circ.witness_map(z, to_host=False, resident=True)
    pr = LG.create_proof(pk, 123456789, 987654321, 555, dh, z[:2], z[2:]); dh.free(); return pr
def ovl():
    return LG.create_proof_with_reduction(pk, circ, 123456789, 987654321, 555, z, share_sort=not os.environ.get("SEPARATE_SORTS"))
for name, fn in (("witness map, then MSMs", seq), ("witness map overlapped", ovl), ("witness map, then MSMs", seq), ("witness map overlapped", ovl)):
    ref = fn()
    for _ in range(4):
        fn()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); p = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    assert all((p[k] == ref[k]).all() for k in ref)
    print("%-26s table=%d wc=%d hc=%d  median %.2f ms  min %.2f  max %.2f  (%.1f M constraints/s)" % (name, table, wc, hc, sorted(ts)[4], min(ts), max(ts), (m + 1) / sorted(ts)[4] / 1e3), flush=True)
if os.environ.get("INFLIGHT"):          # proofs per second with K proofs in flight (host threads; the library queues calls beyond its six slots)
    from concurrent.futures import ThreadPoolExecutor
    for K in [int(x) for x in os.environ["INFLIGHT"].split(",")]:
        with ThreadPoolExecutor(K) as ex:
            list(ex.map(lambda _: ovl(), range(2 * K)))
            NIT = int(os.environ.get("NIT", "16")); t0 = time.perf_counter(); res = list(ex.map(lambda _: ovl(), range(NIT))); dt = (time.perf_counter() - t0) / NIT * 1e3
        assert all((r[k] == ref[k]).all() for r in res for k in ref)
        print("%d proofs in flight: %.2f ms per proof (%.1f M constraints/s)" % (K, dt, (m + 1) / dt / 1e3), flush=True)
if os.environ.get("PROF"):              # device time per stage of one proof
    ca.prof.enable(True); ca.prof.reset(); ovl()
    print("device ms / calls:", {k: (round(v[0], 2), v[1]) for k, v in sorted(ca.prof.read().items(), key=lambda kv: -kv[1][0])}); ca.prof.enable(False)
