"""Development helper (GPU box): stage times of one resident table MSM at n = 2^LOG2N (closed form checked) — the partition sort at 2^22 .. 2^24."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import serde, fixed_base as FB
import bench as B
ca.init(0)
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
for lg in [int(x) for x in os.environ.get("LOGS", "21,22,23,24").split(",")]:
    n = 1 << lg
    ks = B.seeded_scalars(0x5EED2400 + lg, n); sc = B.seeded_scalars(0x5EED2500 + lg, n)
    with FB.WindowTable(ca.G1, gen1[0]) as t1:
        db = t1.multiply_many_to_bases(ks)
        exp, _ = t1.multiply(B.dot_mod_r(ks, sc))
    db.precompute(); ds = ca.DeviceScalars(sc)
    r = db.msm_resident(ds)
    ok = bool((r[:12] == exp).all())
    for _ in range(3): db.msm_resident(ds)
    ca.prof.enable(True); ca.prof.reset()
    t0 = time.perf_counter()
    for _ in range(4): db.msm_resident(ds)
    dt = (time.perf_counter() - t0) / 4 * 1e3
    st = ca.prof.read(); ca.prof.enable(False)
    print("n=2^%d ok=%s %.2f ms |" % (lg, ok, dt), " ".join("%s=%.3f" % (k.split(".")[-1], v[0] / max(1, v[1])) for k, v in st.items()), flush=True)
    db.free(); ds.free()
