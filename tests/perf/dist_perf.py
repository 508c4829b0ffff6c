"""Timing driver (GPU box): per-stage times of the table / plain pipeline under the skewed scalar distributions of SURVEY 8d."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import fixed_base as FB, serde
import bench as B
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
n = 1 << int(os.environ.get("LOG2N", "20"))
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
ks = B.seeded_scalars(1, n); sc = B.seeded_scalars(2, n)
rng = np.random.Generator(np.random.PCG64(9))
dists = {"uniform": sc, "all_equal": np.tile(sc[7], (n, 1))}
s16 = np.zeros((n, 4), np.uint64); s16[:, 0] = rng.integers(0, 1 << 16, n, dtype=np.uint64); dists["16_bit"] = s16
zo = sc.copy(); kind = rng.integers(0, 4, n); zo[kind <= 1] = 0; zo[kind == 1, 0] = 1; dists["half_zeros_ones"] = zo
for table in (False, True):
    with FB.WindowTable(ca.G1, gen1[0]) as t:
        db = t.multiply_many_to_bases(ks)
    if table:
        db.precompute()
    for name, s in dists.items():
        ds = ca.DeviceScalars(s)
        for _ in range(2):
            db.msm_resident(ds)
        ca.prof.enable(True); ca.prof.reset()
        t0 = time.perf_counter()
        for _ in range(3):
            db.msm_resident(ds)
        lat = (time.perf_counter() - t0) / 3 * 1e3
        st = ca.prof.read(); ca.prof.enable(False)
        print("%-6s %-16s %.3f ms | %s" % ("table" if table else "plain", name, lat, " ".join("%s=%.3f" % (k.replace("msm.", ""), v[0] / max(1, v[1])) for k, v in st.items())), flush=True)
        ds.free()
    db.free()
