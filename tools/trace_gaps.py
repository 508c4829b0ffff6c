import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
last = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows
t0 = int(last[0]["Start_Timestamp"])
prev_end = None
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%10.1f us  dur %9.1f us  gap %9.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:60]))
    prev_end = e
