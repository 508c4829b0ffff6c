"""Aggregate n synthetic Groth16 proofs and verify the aggregate on the GPU box; prints per-phase wall time (development helper).
Usage: N=1024 python tests/perf/bench_aggregation.py"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import crypto_amd as ca
from crypto_amd import legogroth16 as LG, aggregation as AG, fixed_base as FB
from crypto_amd.aggregation import ops
import oracle_c as O

R = ops.R_MOD
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
n = int(os.environ.get("N", "1024"))
rng = np.random.default_rng(1)
rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
alpha, beta, gamma, delta = rnd(), rnd(), rnd(), rnd()
n_pub = 4
ks = [rnd() for _ in range(n_pub + 1)]
t0 = time.time()
with FB.WindowTable(ca.G1, O.G1.generator()) as t1, FB.WindowTable(ca.G2, O.G2.generator()) as t2:
    small1, _ = t1.multiply_many([alpha, 1] + ks)
    small2, _ = t2.multiply_many([beta, gamma, delta])
    vk = LG.VerifyingKey(small1[0], small2[0], small2[1], small2[2], small1[2:], small1[1], 0)
    inputs, av, bv, cv = [], [], [], []
    di = pow(delta, R - 2, R)
    for _ in range(n):
        x = [rnd() for _ in range(n_pub)]
        a, b = rnd(), rnd()
        s = (ks[0] + sum(xi * ki for xi, ki in zip(x, ks[1:]))) % R
        inputs.append(x); av.append(a); bv.append(b); cv.append((a * b - alpha * beta - s * gamma) * di % R)
    A, _ = t1.multiply_many(av); B, _ = t2.multiply_many(bv); Cc, _ = t1.multiply_many(cv)
proofs = [{"a": A[i], "b": B[i], "c": Cc[i]} for i in range(n)]
t_make = time.time() - t0
t0 = time.time(); srs = AG.setup_fake_srs(rnd(), rnd(), n, O.G1.generator(), O.G2.generator()); pk, vsrs = srs.specialize(n); t_srs = time.time() - t0
AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs)          # warm the library's per-slot workspaces (first-call allocations)
ca.prof.enable(True); ca.prof.reset()
t0 = time.time(); agg = AG.aggregate_proofs(pk, AG.MerlinTranscript(b"bench"), proofs); t_agg = time.time() - t0
st_agg = ca.prof.read(); ca.prof.reset()
AG.verify_aggregate_proof(vsrs, {"vk": vk}, inputs, agg, rnd(), AG.MerlinTranscript(b"bench"))
ca.prof.reset()
t0 = time.time(); AG.verify_aggregate_proof(vsrs, {"vk": vk}, inputs, agg, rnd(), AG.MerlinTranscript(b"bench")); t_ver = time.time() - t0
st_ver = ca.prof.read(); ca.prof.enable(False)
fmt = lambda st: {k: [round(v[0], 2), v[1]] for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])[:10]}
print(json.dumps({"n_proofs": n, "make_proofs_s": round(t_make, 3), "srs_s": round(t_srs, 3), "aggregate_s": round(t_agg, 3), "verify_s": round(t_ver, 3),
                  "aggregate_device_ms_calls": fmt(st_agg), "verify_device_ms_calls": fmt(st_ver)}))
