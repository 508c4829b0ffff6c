cd /tmp && export TMPDIR=/tmp
for N in 600 4096; do
  D=/root/repo/gpurun_out/prof_sm_$N
  N=$N timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python /root/repo/tools/dev/small_msm_loop.py 2>/dev/null | grep "per resident"
  python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('$D/*/*kernel_stats.csv')[0])):
    print('   %-70s calls %4s avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
"
  rm -rf $D
done
