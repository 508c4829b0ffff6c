"""Timing driver (GPU box): the witness map at D = 2^LOG2N with the circuit resident (the bench's circuit shape), for rocprofv3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_amd as ca
from crypto_amd import qap
import bench as B
ca.init(0)
_twin = ca.twin(); _twin.__enter__()      # knobs / stage timers live in the development twin (include/dock_gpu_dev.h): this script runs on it
n = 1 << int(os.environ.get("LOG2N", "20")); m = n - 3
idx = np.arange(m, dtype=np.uint32)
one = np.zeros((1, 4), np.uint64); one[0, 0] = 1
a_rp = np.arange(m + 2, dtype=np.uint64); a_cl = np.concatenate([2 + idx, [2 + m]]).astype(np.uint32); a_vl = np.repeat(one, m + 1, 0)
b_cl = np.concatenate([2 + idx, [0]]).astype(np.uint32)
c_rp = np.concatenate([2 * np.arange(m + 1, dtype=np.uint64), [2 * m + 1]]).astype(np.uint64)
c_cl = np.concatenate([np.stack([3 + idx, np.zeros(m, np.uint32)], 1).reshape(-1), [1]]).astype(np.uint32)
circ = qap.DeviceR1cs((a_rp, a_cl, a_vl), (a_rp, b_cl, a_vl), (c_rp, c_cl, np.repeat(one, 2 * m + 1, 0)), m + 3, 2, m + 1)
z = B.seeded_scalars(7, m + 3)
def wm():
    _, dh = circ.witness_map(z, to_host=False, resident=True); dh.free()
for _ in range(3): wm()
ca.prof.enable(True); ca.prof.reset()
t0 = time.perf_counter()
for _ in range(10): wm()
dt = (time.perf_counter() - t0) / 10 * 1e3
st = ca.prof.read()
print("witness map D=2^%d: %.3f ms | %s" % (int(os.environ.get("LOG2N", "20")), dt, " ".join("%s=%.3f" % (k, v[0] / max(1, v[1])) for k, v in st.items())))
