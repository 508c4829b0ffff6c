# Development helper (GPU box): per-kernel durations of the bucket reduction, scan form (lanes 4) against bit marginals (lanes 0), one call in flight
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
for L in 4 2 0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r05a_prof_l$L -- python /root/repo/bench.py --steps 20 --no-secondary --no-cpu-baseline --inflight 1 --reduce-lanes $L > /dev/null 2>&1
  cp $O/r05a_prof_l$L/*/*kernel_stats.csv $O/r05a_kernel_stats_lanes$L.csv
  rm -rf $O/r05a_prof_l$L
  echo "== lanes $L"; grep "reduce\|accumulate\|fixup" $O/r05a_kernel_stats_lanes$L.csv | cut -c1-60,100-400 | head
done
