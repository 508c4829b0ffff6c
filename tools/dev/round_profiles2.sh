set -x
TAG=${TAG:-r02d}
O=/root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof1b -- python /root/repo/bench.py --inflight 1 --steps 20 --no-secondary --no-cpu-baseline > $O/${TAG}_bench_inflight1_under_rocprof.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof6b -- python /root/repo/bench.py --steps 20 --no-secondary --no-cpu-baseline > /dev/null 2>&1
cp $O/${TAG}_prof1b/*/*kernel_stats.csv $O/${TAG}_kernel_stats_inflight1.csv
cp $O/${TAG}_prof6b/*/*kernel_stats.csv $O/${TAG}_kernel_stats_default_inflight6.csv
