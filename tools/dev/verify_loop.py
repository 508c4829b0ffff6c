"""Development helper (GPU box): the verifier-side calls round 5 added last, each in a loop of its own — what rocprofv3 wraps for their kernel statistics
and counters (tools/dev/round6_profiles.sh): WHAT = batch (dgpu_legogroth16_verify_batch over 1024 proofs), scaled (dgpu_multi_miller_loop_scaled, 1024
pairs), scale (dgpu_g1_scale_batch, 1024 points: k_g1_scale_oct), one (dgpu_legogroth16_verify of one proof), miller (the plain 1024-pair loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch  # noqa: F401
import crypto_amd as ca
import bench as B
from crypto_amd import pairing, fixed_base as FB, legogroth16 as LG
from crypto_amd.pairing_check import g1_scale_each
ca.init(0)
if os.environ.get("MLMODE"):      # a form of the Miller kernels other than the default (development twin)
    from crypto_amd._native import lib
    _tw = ca.twin(); _tw.__enter__(); assert lib().dgpu_set_miller_pipeline(int(os.environ["MLMODE"])) == 0
WHAT = os.environ.get("WHAT", "batch"); K = int(os.environ.get("K", "20")); nv = int(os.environ.get("N", "1024"))
R_MOD = B.R_MOD
ints = lambda seed, k: [int(x[0]) | (int(x[1]) << 64) | (int(x[2]) << 128) | (int(x[3]) << 192) for x in B.seeded_scalars(seed, k)]
lim = lambda vals: np.array([[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)] for v in vals], dtype=np.uint64)
al, be, ga, de, g0, g1x = ints(0x5EED0020, 6)
av, bv, dv, xv = ints(0x5EED0021, nv), ints(0x5EED0022, nv), ints(0x5EED0023, nv), ints(0x5EED0024, nv)
dinv = pow(de, R_MOD - 2, R_MOD)
cv = [((a * b - al * be - (g0 + x * g1x + d) * ga) * dinv) % R_MOD for a, b, d, x in zip(av, bv, dv, xv)]
from crypto_amd import serde
G1GEN = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))[0][0]
G2GEN = serde.deserialize(ca.G2, bytes.fromhex(B.G2_GEN_COMPRESSED))[0][0]
with FB.WindowTable(ca.G2, G2GEN) as t2, FB.WindowTable(ca.G1, G1GEN) as t1:
    A_, _ = t1.multiply_many(lim(av)); C_, _ = t1.multiply_many(lim(cv)); D_, _ = t1.multiply_many(lim(dv)); K_, _ = t1.multiply_many(lim([al, g0, g1x, 1]))
    B_, _ = t2.multiply_many(lim(bv)); V_, _ = t2.multiply_many(lim([be, ga, de]))
vk = LG.VerifyingKey(K_[0], V_[0], V_[1], V_[2], K_[1:3], K_[3], 0)
pvk = LG.prepare_verifying_key(vk)
proofs = [{"a": A_[i], "b": B_[i], "c": C_[i], "d": D_[i]} for i in range(nv)]
pubs = [lim([x]) for x in xv]
packed = LG.pack_proofs(proofs, pubs)
m = B.seeded_scalars(3, nv)
fs = {"batch": lambda: LG.verify_proofs_batch_abi(pvk, None, None, 0x5EED0029, packed=packed),
      "scaled": lambda: pairing.multi_miller_loop_scaled(A_, m, B_),
      "scale": lambda: g1_scale_each(A_, m),
      "one": lambda: LG.verify_proof_abi(pvk, proofs[1], pubs[1]),
      "miller": lambda: pairing.multi_miller_loop(A_, B_)}
f = fs[WHAT]
assert WHAT not in ("batch", "one") or f()
for _ in range(8): f()
t0 = time.perf_counter()
for _ in range(K): f()
print("%s (n = %d): %.3f ms per call" % (WHAT, nv, (time.perf_counter() - t0) / K * 1e3), flush=True)
