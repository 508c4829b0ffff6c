# Development helper (GPU box): kernel traces of the verifier-side calls' loops (tools/dev/verify_loop.py) -> gpurun_out/<TAG>_timeline_<what>.txt: the kernels
# of the LAST call with start offset, duration, hardware queue and stream (tools/dev/prove_timeline.py)
TAG=${TAG:-r06}
cd /tmp && export TMPDIR=/tmp
for W in batch one scaled miller; do
  WHAT=$W K=4 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /root/repo/gpurun_out/prof_vt_$W -- python /root/repo/tools/dev/verify_loop.py > /dev/null 2>&1
  GAP_NS=400000 python /root/repo/tools/dev/prove_timeline.py /root/repo/gpurun_out/prof_vt_$W/*/*kernel_trace.csv 5 > /root/repo/gpurun_out/${TAG}_timeline_$W.txt
  rm -rf /root/repo/gpurun_out/prof_vt_$W
done
