"""Timing driver (GPU box): batch verification of N LegoGroth16 proofs of one circuit through the lazy RandomizedPairingChecker
(what proof_system/src/verifier.rs:1829-1835 does with a composite proof): 3 pairs per proof, the targets folded by one multi-exponentiation.
Prints proofs/s and pairs/s; the proofs are real (toy circuit of legogroth16/src/tests.rs) and one tampered batch must be rejected."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_c as O
import lego_setup as LS
import crypto_amd as ca
from crypto_amd import qap, legogroth16 as LG, pairing
from crypto_amd.pairing_check import RandomizedPairingChecker
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
R = LS.R
N = int(os.environ.get("N", "1024"))
rng = np.random.default_rng(1)
rnd = lambda: int.from_bytes(rng.bytes(40), "little") % (R - 1) + 1
g1 = lambda k: O.G1.to_affine(O.G1.mul(O.G1.generator(), O.int_to_limbs(k % R, 4)))[0]
g2 = lambda k: O.G2.to_affine(O.G2.mul(O.G2.generator(), O.int_to_limbs(k % R, 4)))[0]
silly = lambda a, b: {"A": [[(1, 2)]], "B": [[(1, 3)]], "C": [[(1, 1)]], "z": [1, a * b % R, a, b], "n_inst": 2, "n_wit": 2, "n_cons": 1}
shape = silly(1, 1)
pk, _ = LG.generate_parameters(shape["A"], shape["B"], shape["C"], 2, 2, 2, *[rnd() for _ in range(6)], g1(rnd()), g2(rnd()))
pvk = LG.prepare_verifying_key(pk.vk)
circ = qap.DeviceR1cs(*[qap.csr(shape[k]) for k in "ABC"], 4, 2, 1)
proofs, pubs = [], []
t0 = time.perf_counter()
for _ in range(N):
    cs = silly(rnd(), rnd())
    z = LS.scalars(cs["z"])
    proofs.append(LG.create_proof_with_reduction(pk, circ, rnd(), rnd(), rnd(), z)); pubs.append(z[1:2])
print("made %d proofs in %.2f s" % (N, time.perf_counter() - t0), flush=True)


def batch_verify(proofs, pubs):
    return LG.verify_proofs_batch(pvk, proofs, pubs, rnd())


assert batch_verify(proofs, pubs)
bad = list(proofs); bad[N // 2] = dict(bad[N // 2]); bad[N // 2]["c"] = proofs[0]["c"]
assert not batch_verify(bad, pubs)
ts = []
for _ in range(5):
    t0 = time.perf_counter(); ok = batch_verify(proofs, pubs); ts.append(time.perf_counter() - t0); assert ok
t = sorted(ts)[2]
if os.environ.get("PROF"):
    import cProfile, pstats, io
    ca.prof.enable(True); ca.prof.reset(); pr = cProfile.Profile(); pr.enable(); batch_verify(proofs, pubs); pr.disable()
    print("device ms / calls:", {k: (round(v[0], 2), v[1]) for k, v in sorted(ca.prof.read().items(), key=lambda kv: -kv[1][0])}); ca.prof.enable(False)
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(14); print(st.getvalue()[:3500])
t1 = time.perf_counter(); single = all(LG.verify_proof(pvk, p, x) for p, x in zip(proofs[:64], pubs[:64])); t1 = (time.perf_counter() - t1) / 64
tm = []
assert LG.verify_proofs_batch_merged(pvk, proofs, pubs, rnd()) and not LG.verify_proofs_batch_merged(pvk, bad, pubs, rnd())
for _ in range(5):
    t0 = time.perf_counter(); ok = LG.verify_proofs_batch_merged(pvk, proofs, pubs, rnd()); tm.append(time.perf_counter() - t0); assert ok
print("merged batch check of %d proofs (N + 2 pairs, two MSMs): %.1f ms = %.0f proofs/s" % (N, sorted(tm)[2] * 1e3, N / sorted(tm)[2]))
print("batch of %d proofs: %.1f ms = %.0f proofs/s = %.0f pairs/s; one at a time: %.2f ms per proof (%s)" % (N, t * 1e3, N / t, 3 * N / t, t1 * 1e3, single))
