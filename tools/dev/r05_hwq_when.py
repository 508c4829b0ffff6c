import os, sys, time
mode = sys.argv[1]
if mode == "before_all": os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("Q", "8")
import torch
if mode == "after_import": os.environ["GPU_MAX_HW_QUEUES"] = os.environ.get("Q", "8")
if mode == "after_cuda":
    torch.cuda.set_device(0); torch.zeros(1, device="cuda"); os.environ["GPU_MAX_HW_QUEUES"] = "8"
sys.argv = [sys.argv[0]]
exec(open("/root/repo/tools/dev/r05_agg_ab.py").read())
