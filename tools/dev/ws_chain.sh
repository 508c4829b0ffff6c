cd /tmp && export TMPDIR=/tmp
for M in 7 15; do for N in 3 1024 4096; do
  D=/root/repo/gpurun_out/prof_wc_${N}_$M
  MLMODE=$M N=$N timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python /root/repo/tools/dev/ws_chain_time.py > /dev/null 2>&1
  python3 -c "
import csv,sys,glob
for r in csv.DictReader(open(glob.glob('$D/*/*kernel_stats.csv')[0])):
    if 'k_miller_lines' in r['Name']: print('mode $M n $N:', r['Name'].split('(')[1][-24:], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3))
"
  rm -rf $D
done; done
