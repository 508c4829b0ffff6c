# Development helper (GPU box): kernel trace of the library's SnarkPack aggregator (1024 proofs, compiled transcript) -> gpurun_out/<TAG>_timeline_aggregate.txt
TAG=${TAG:-r06}
cd /tmp && export TMPDIR=/tmp
cat > /tmp/agg_one.py <<'PY'
import sys, os, time, numpy as np
R_ = "/root/repo"; sys.path[:0] = [R_ + "/oracle", R_ + "/tests", R_]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import oracle_c as O, util as U, crypto_amd as ca
from crypto_amd import aggregation as AG
from crypto_amd.aggregation import native as NA
from crypto_amd.fixed_base import WindowTable
ca.init(0)
n = 1024
rng = np.random.default_rng(3)
ints = lambda k: [int.from_bytes(rng.bytes(40), "little") % (U.R - 1) + 1 for _ in range(k)]
def fixed(curve, g, ks):
    with WindowTable(curve, g, len(ks)) as t:
        return t.multiply_many(ks)[0]
g, h = O.G1.generator(), O.G2.generator()
A, Cc, D = fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)), fixed(ca.G1, g, ints(n)); B = fixed(ca.G2, h, ints(n))
proofs = [{"a": A[i], "b": B[i], "c": Cc[i], "d": D[i]} for i in range(n)]
pk, vsrs = AG.setup_fake_srs(ints(1)[0], ints(1)[0], n, g, h).specialize(n)
NT = AG.NativeMerlinTranscript
for _ in range(4): NA.aggregate_proofs_words(pk, NT(b"t"), proofs, with_d=True)
time.sleep(0.05)
t0 = time.perf_counter(); NA.aggregate_proofs_words(pk, NT(b"t"), proofs, with_d=True); print("aggregate: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
PY
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof_agg -- python /tmp/agg_one.py 2>/dev/null | grep aggregate
GAP_NS=20000000 TAIL=600 python /root/repo/tools/dev/prove_timeline.py /root/repo/gpurun_out/prof_agg/*/*kernel_trace.csv 30 > /root/repo/gpurun_out/${TAG}_timeline_aggregate.txt
rm -rf /root/repo/gpurun_out/prof_agg
