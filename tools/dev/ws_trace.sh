# Development helper (GPU box): timelines + kernel statistics of the 1024-pair Miller loop and of one proof's verification with the line kernel as
# sixteen lanes per pair (MLMODE=7) and as a wave per role (MLMODE=15) -> gpurun_out/<TAG>_{timeline,stats}_<what>_mode<m>.txt
TAG=${TAG:-r06w}
cd /tmp && export TMPDIR=/tmp
for M in ${MODES:-15 31}; do for W in miller one; do
  D=/root/repo/gpurun_out/prof_ws_${W}_$M
  MLMODE=$M WHAT=$W K=6 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python /root/repo/tools/dev/verify_loop.py > /root/repo/gpurun_out/${TAG}_loop_${W}_mode$M.txt 2>&1
  GAP_NS=400000 python /root/repo/tools/dev/prove_timeline.py $D/*/*kernel_trace.csv 5 > /root/repo/gpurun_out/${TAG}_timeline_${W}_mode$M.txt
  cp $D/*/*kernel_stats.csv /root/repo/gpurun_out/${TAG}_stats_${W}_mode$M.csv
  rm -rf $D
done; done
