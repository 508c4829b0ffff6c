import sys, os, time, numpy as np
R_ = "/root/repo"; sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import aggregation as AG
from crypto_amd.aggregation import native as NA
from test_gpu_aggregation import make_statement
ca.init(0)
n = int(os.environ.get("N", "64"))
vk, proofs, inputs, _ = make_statement(n, 1, seed=3)
pk, vsrs = AG.setup_fake_srs(11, 13, n, O.G1.generator(), O.G2.generator()).specialize(n)
w = NA.aggregate_proofs_words(pk, AG.MerlinTranscript(b"t"), proofs)
for _ in range(3):
    t0 = time.perf_counter(); NA.verify_aggregate_proof(vsrs, {"vk": vk}, inputs, w, 77, AG.MerlinTranscript(b"t")); print("verify %.2f ms" % ((time.perf_counter() - t0) * 1e3), file=sys.stderr)
