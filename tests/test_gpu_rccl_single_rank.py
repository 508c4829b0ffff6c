"""GPU (-m gpu): the collectives bench.py and crypto_amd/sharded.py issue at N > 1, over the backend they issue them on there — "nccl" (= RCCL) — with the one
rank a one-GPU box allows: the int64 all_gather of a partial point from device memory (gather_and_fold's payload), all_gather_object of a Python integer
(the closed form's per-rank dot products), barrier, and the float64 MAX all_reduce of the timed region.  RCCL refuses two ranks on one device, so the
two-rank tests run over gloo; this one pins the dtypes, device placement and `device_id` initialisation of the RCCL path itself."""
import os
import subprocess
import sys
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import crypto_amd as ca
from crypto_amd import sharded
ca.init(0)
part = (np.arange(18, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1 << 63)          # limbs with the top bit set: the int64 view must carry them
t = torch.from_numpy(part.view(np.int64).copy()).to(dev)
buf = [torch.empty_like(t)]
dist.all_gather(buf, t)
assert (buf[0].cpu().numpy().view(np.uint64) == part).all()
got = [None]
dist.all_gather_object(got, (1 << 254) + 12345)
assert got == [(1 << 254) + 12345]
dist.barrier()
x = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(x, op=dist.ReduceOp.MAX)
assert float(x.item()) == 1.25
# the library's own path: one rank, so gather_and_fold folds the local partial (a real MSM result) and returns the same point
from crypto_amd import fixed_base as FB, serde
import bench as B
gen1, _ = serde.deserialize(ca.G1, bytes.fromhex(B.G1_GEN_COMPRESSED))
ks = B.seeded_scalars(7, 4096); sc = B.seeded_scalars(8, 4096)
with FB.WindowTable(ca.G1, gen1[0]) as gtab:
    db = gtab.multiply_many_to_bases(ks)
res = sharded.gather_and_fold(ca.G1, db.msm_resident(ca.DeviceScalars(sc)), dev)
with FB.WindowTable(ca.G1, gen1[0]) as gtab:
    exp_xy, exp_inf = gtab.multiply(B.dot_mod_r(ks, sc))
assert not exp_inf and (res[:12] == exp_xy).all()
dist.destroy_process_group()
print("rccl single rank ok")
"""


def test_rccl_collectives_of_the_sharded_path_on_one_rank():
    assert torch.cuda.is_available()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "rccl single rank ok" in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
