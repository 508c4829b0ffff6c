"""Summarise rocprofv3 --pmc output: per kernel and counter, number of dispatches and the mean counter value.
Usage: python tools/pmc_summary.py <dir with *counter_collection.csv> [...] > profiles/<round>_pmc_summary.txt"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:48], row.get("Counter_Name", "?"))
                acc[k][0] += 1
                acc[k][1] += float(row.get("Counter_Value", 0) or 0)
for (kern, ctr), (n, s) in sorted(acc.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print("%-50s %-18s calls=%d avg=%.1f" % (kern, ctr, n, s / n))
