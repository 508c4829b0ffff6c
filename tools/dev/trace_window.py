# tools/dev/trace_window.py <kernel_trace.csv> — the last aggregation's kernels as a timeline (start offset, duration, name) for overlap inspection
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last 400 dispatches
key = sys.argv[3] if len(sys.argv) > 3 else None
if key:
    last = max(i for i, r in enumerate(rows) if key in r["Kernel_Name"])
    rows = rows[max(0, last - int(sys.argv[2])):last + 3]
else:
    rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -260:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
