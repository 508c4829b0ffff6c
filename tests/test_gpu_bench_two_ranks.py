"""GPU (-m gpu): bench.py's N > 1 path end to end on the box's ONE GPU — two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 …`), both on device 0 (DGPU_BENCH_SAME_DEVICE) and with gloo as the collective backend
(RCCL refuses two ranks on one device): per-rank tables, the timed region with its barriers, the all_gather + fold of the partial points, the closed
form over both ranks' terms, rank 0's stage leg on the development twin AFTER the collectives, and `scaling_base` (rank 0 alone computes config 5's 2^24
terms on its GPU while rank 1 leaves).  A plumbing test, not a measurement: the 1 -> 8 curve is the driver's SCALE run."""
import json
import os
import subprocess
import sys
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_on_one_gpu_print_one_valid_line():
    assert torch.cuda.is_available()
    env = dict(os.environ, DGPU_BENCH_SAME_DEVICE="1", DGPU_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29537",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--log2n", "18"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["bit_exact_vs_closed_form"] is True
    assert out["config"]["terms_per_step"] == 2 << 18 and len(out["config"]["per_key_setup_ms"]["precomputed_table_per_rank"]) == 2
    assert out["device_allocations_in_timed_region"] == 0
    assert out["stages_ms_one_in_flight"].get("accumulate", 0) > 0 and out["roofline"]["avg_ms"] > 0          # rank 0's stage leg ran (on the twin)
    assert out["scaling_base"] and out["scaling_base"] > 100, out.get("scaling_base_error")                      # the 1-GPU rate on 2^24 terms, measured by rank 0
    assert "cpu_baseline" not in out and "secondary" not in out
