# Development helper (GPU box): the round's evidence under gpurun_out/ — full GPU test suite, bench lines, rocprofv3 kernel stats, NTT counters, timing tables.
set -x
TAG=${TAG:-r02d}
O=/root/repo/gpurun_out
python -m pytest tests -q -m gpu 2>&1 | tail -3 > $O/${TAG}_pytest_gpu.txt
python bench.py > $O/${TAG}_bench_default.json 2> /dev/null
python bench.py --no-table > $O/${TAG}_bench_no_table.json 2> /dev/null
python tests/perf/dist_perf.py 2>&1 | grep -E "^(plain|table)" > $O/${TAG}_scalar_distributions.txt
python tests/perf/qap_perf.py 2>&1 | grep "witness map" > $O/${TAG}_witness_map.txt
for l in 12 16 18 21; do LOG2N=$l python tests/perf/qap_perf.py 2>&1 | grep "witness map" >> $O/${TAG}_witness_map.txt; done
python tests/perf/prove_perf.py 2>&1 | grep "witness map" > $O/${TAG}_prove.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof1 -- python /root/repo/bench.py --inflight 1 > $O/${TAG}_bench_inflight1_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_wm -- python /root/repo/tests/perf/qap_perf.py > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_wm -- python /root/repo/tests/perf/qap_perf.py > /dev/null 2>&1
cp $O/${TAG}_prof1/*/*kernel_stats.csv $O/${TAG}_kernel_stats_inflight1.csv
cp $O/${TAG}_prof_wm/*/*kernel_stats.csv $O/${TAG}_kernel_stats_witness_map.csv
cp $O/${TAG}_pmc_wm/*/*counter_collection.csv $O/${TAG}_pmc_witness_map_raw.csv
ls -la $O | tail -20
