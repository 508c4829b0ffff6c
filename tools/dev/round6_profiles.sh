# Development helper (GPU box): round 6's rocprofv3 evidence under gpurun_out/<TAG>_*.  PART=verify: kernel statistics and counters (SQ_INSTS_VALU,
# GRBM_GUI_ACTIVE, SQ_WAVES; one counter per run, as MI355X_MICROARCH.md prescribes) for the verifier-side calls round 5 added last
# (k_g1_scale_oct, the scaled Miller loop, dgpu_legogroth16_verify_batch, one proof verified).  PART=msm: the headline set of round 5 again (kernel
# statistics with 1 and 6 calls in flight, G2 loop, witness map, the G1 / G2 counters and the traffic file bench.py reads, with the commit passed in as
# COMMIT=<short hash>: the box has no .git).  PART=all: both.
set -x
TAG=${TAG:-r06}
COMMIT=${COMMIT:-unknown}
PART=${PART:-all}
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
V="python /root/repo/tools/dev/verify_loop.py"
if [ "$PART" = verify ] || [ "$PART" = all ]; then
  for W in batch scaled scale one miller; do
    WHAT=$W timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$W -- $V 2>/dev/null | grep "per call" > $O/${TAG}_verify_loop_${W}_under_rocprof.txt
    cp $O/${TAG}_prof_$W/*/*kernel_stats.csv $O/${TAG}_kernel_stats_verify_$W.csv
    for C in SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES; do
      WHAT=$W K=6 timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_${W}_$C -- $V > /dev/null 2>&1
    done
    python /root/repo/tools/pmc_summary.py $O/${TAG}_pmc_${W}_* > $O/${TAG}_pmc_summary_verify_$W.txt
    rm -rf $O/${TAG}_prof_$W $O/${TAG}_pmc_${W}_*
  done
  for W in batch scaled scale one miller; do WHAT=$W K=40 $V 2>/dev/null | grep "per call"; done > $O/${TAG}_verify_loop_no_profiler.txt
fi
if [ "$PART" = msm ] || [ "$PART" = all ]; then
  B="python /root/repo/bench.py --steps 20 --no-secondary --no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof1 -- $B --inflight 1 > $O/${TAG}_bench_inflight1_under_rocprof.json 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof6 -- $B > $O/${TAG}_bench_inflight6_under_rocprof.json 2>/dev/null
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_profg2 -- python /root/repo/tools/dev/g2_loop.py > $O/${TAG}_g2_loop.txt 2>/dev/null
  cp $O/${TAG}_prof1/*/*kernel_stats.csv $O/${TAG}_kernel_stats_inflight1.csv
  cp $O/${TAG}_prof6/*/*kernel_stats.csv $O/${TAG}_kernel_stats_default_inflight6.csv
  cp $O/${TAG}_profg2/*/*kernel_stats.csv $O/${TAG}_kernel_stats_g2.csv
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES; do
    timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_g1_$C -- python /root/repo/bench.py --inflight 1 --steps 8 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
    K=6 timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_g2_$C -- python /root/repo/tools/dev/g2_loop.py > /dev/null 2>&1
  done
  python /root/repo/tools/pmc_summary.py $O/${TAG}_pmc_g1_* > $O/${TAG}_pmc_summary_g1.txt
  python /root/repo/tools/pmc_summary.py $O/${TAG}_pmc_g2_* > $O/${TAG}_pmc_summary_g2.txt
  python /root/repo/tools/traffic_json.py $O/${TAG}_pmc_g1_FETCH_SIZE $O/${TAG}_pmc_g1_WRITE_SIZE $COMMIT 20 > $O/${TAG}_traffic_accumulate.json
  rm -rf $O/${TAG}_prof1 $O/${TAG}_prof6 $O/${TAG}_profg2 $O/${TAG}_pmc_g1_* $O/${TAG}_pmc_g2_*
fi
ls -la $O | grep ${TAG}
