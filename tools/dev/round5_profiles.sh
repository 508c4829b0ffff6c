# Development helper (GPU box): the round's rocprofv3 evidence under gpurun_out/<TAG>_*: kernel statistics (G1 headline loop with 1 and 6 calls in
# flight, G2 loop, Miller loops, small MSMs) and the PMC passes (one counter per run, as MI355X_MICROARCH.md prescribes) for G1 and G2, and the
# traffic file bench.py reads (with the commit it was measured at: COMMIT=<short hash> is passed in, the box has no .git).
set -x
TAG=${TAG:-r05}
COMMIT=${COMMIT:-unknown}
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --steps 20 --no-secondary --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof1 -- $B --inflight 1 > $O/${TAG}_bench_inflight1_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof6 -- $B > $O/${TAG}_bench_inflight6_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_profg2 -- python /root/repo/tools/dev/g2_loop.py > $O/${TAG}_g2_loop.txt 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_profml -- python /root/repo/tools/dev/ml_loop.py 2>/dev/null | grep -v "^[EW]2026" > $O/${TAG}_ml_loop.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_profsm -- python /root/repo/tools/dev/small_msm_prof.py > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_profagg -- python /root/repo/tools/dev/agg_time.py 2>/dev/null | grep "^n =" > $O/${TAG}_agg_time_under_rocprof.txt
cp $O/${TAG}_prof1/*/*kernel_stats.csv $O/${TAG}_kernel_stats_inflight1.csv
cp $O/${TAG}_prof6/*/*kernel_stats.csv $O/${TAG}_kernel_stats_default_inflight6.csv
cp $O/${TAG}_profg2/*/*kernel_stats.csv $O/${TAG}_kernel_stats_g2.csv
cp $O/${TAG}_profml/*/*kernel_stats.csv $O/${TAG}_kernel_stats_miller.csv
cp $O/${TAG}_profsm/*/*kernel_stats.csv $O/${TAG}_kernel_stats_small_msm.csv
cp $O/${TAG}_profagg/*/*kernel_stats.csv $O/${TAG}_kernel_stats_aggregation.csv
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES; do
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_g1_$C -- python /root/repo/bench.py --inflight 1 --steps 8 --warmup 1 --no-secondary --no-cpu-baseline > /dev/null 2>&1
  K=6 timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/${TAG}_pmc_g2_$C -- python /root/repo/tools/dev/g2_loop.py > /dev/null 2>&1
done
python /root/repo/tools/pmc_summary.py $O/${TAG}_pmc_g1_* > $O/${TAG}_pmc_summary_g1.txt
python /root/repo/tools/pmc_summary.py $O/${TAG}_pmc_g2_* > $O/${TAG}_pmc_summary_g2.txt
python /root/repo/tools/traffic_json.py $O/${TAG}_pmc_g1_FETCH_SIZE $O/${TAG}_pmc_g1_WRITE_SIZE $COMMIT 20 > $O/${TAG}_traffic_accumulate.json
rm -rf $O/${TAG}_prof1 $O/${TAG}_prof6 $O/${TAG}_profg2 $O/${TAG}_profml $O/${TAG}_profsm $O/${TAG}_profagg $O/${TAG}_pmc_g1_* $O/${TAG}_pmc_g2_*
ls -la $O | grep ${TAG}
