"""Development helper: the kernels of the LAST proof of a rocprofv3 kernel trace of tests/perf/prove_perf.py (tools/dev/prove_trace.sh), one line per
kernel of at least MIN_US microseconds: start, duration, hardware queue, stream, host thread.  Usage: python tools/dev/prove_timeline.py trace.csv [MIN_US]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tail = rows[-int(__import__('os').environ.get('TAIL', '160')):]
b = 0
for i in range(len(tail) - 1, 0, -1):
    if int(tail[i]['Start_Timestamp']) - max(int(x['End_Timestamp']) for x in tail[:i]) > int(__import__('os').environ.get('GAP_NS', '100000')):
        b = i; break
tail = tail[b:]
t0 = int(tail[0]['Start_Timestamp'])
print(sorted(set((r['Queue_Id'], r['Stream_Id'], r['Thread_Id']) for r in tail)))
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '').replace('msm::', '')
    if (e - s) / 1e3 >= min_us or 'ps_count' in n:
        print("%8.1f %8.1f q%-2s s%-3s t%-5s %-40s" % ((s - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], r['Stream_Id'], r['Thread_Id'], n[:40]))
print("span %.1f us" % ((max(int(r['End_Timestamp']) for r in tail) - t0) / 1e3))
