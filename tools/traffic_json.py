"""profiles/traffic_accumulate.json from the rocprofv3 --pmc passes of `bench.py --inflight 1` (FETCH_SIZE and WRITE_SIZE, one counter per
run): HBM bytes per k_accumulate launch = 2 x FETCH_SIZE + WRITE_SIZE (kB -> B; the doubling of FETCH_SIZE is the gfx950 correction
/opt/skills/guides/MI355X_MICROARCH.md prescribes for 16-byte-per-lane gathers of whole 128-byte records).  bench.py copies the figure into
`roofline.traffic` together with the commit it was measured at.
Usage: python tools/traffic_json.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> <commit> <log2n> > profiles/traffic_accumulate.json"""
import csv, glob, json, os, sys


def mean(d, counter, needle):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter and needle in row.get("Kernel_Name", ""):
                tot += float(row.get("Counter_Value", 0) or 0); n += 1
    return (tot / n, n) if n else (None, 0)


fetch, nf = mean(sys.argv[1], "FETCH_SIZE", "k_accumulate<msm::G1S, false>")
write, nw = mean(sys.argv[2], "WRITE_SIZE", "k_accumulate<msm::G1S, false>")
assert fetch is not None and write is not None, "no k_accumulate rows in the counter files"
print(json.dumps({"log2n": int(sys.argv[4]), "table": True, "kernel": "msm::k_accumulate<msm::G1S, false>", "commit": sys.argv[3],
                  "fetch_size_kb_raw": round(fetch, 1), "write_size_kb": round(write, 1), "launches_averaged": [nf, nw],
                  "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
                  "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `bench.py --inflight 1 --steps 8` (tools/dev/round4_profiles.sh); "
                            "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane gathers of whole 128-B records"}, indent=1))
