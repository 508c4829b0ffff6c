"""Development helper: per-proof timeline of the large kernels from a rocprofv3 kernel trace of tests/perf/prove_perf.py.
Usage: python tools/dev/prove_timeline.py <kernel_trace.csv> [n_proofs]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_csr_eval' in r['Kernel_Name']]
starts = idx[0::3]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
for pi in range(len(starts) - n - 1, len(starts) - 1):
    a, b = starts[pi], starts[pi + 1]
    t0 = int(rows[a]['Start_Timestamp'])
    ev = []
    for r in rows[a:b]:
        full = r['Kernel_Name']
        st = (int(r['Start_Timestamp']) - t0) / 1e6; en = (int(r['End_Timestamp']) - t0) / 1e6
        g = "G2" if 'G2' in full else "G1"
        for key, tag in (('k_accumulate', 'acc'), ('k_reduce_top', 'top'), ('k_reduce_l0', 'l0'), ('k_ps_bucket', 'psb'), ('k_fixup_heavy_ranges', 'hvy')):
            if key in full and en - st > 0.05:
                ev.append((tag + (g if tag != 'psb' else ''), st, en))
        if 'k_ntt_r4' in full and en - st > 0.3:
            ev.append(("ntt", st, en))
    last = max((int(r['End_Timestamp']) - t0) / 1e6 for r in rows[a:b])
    print("proof %d: next %.2f last %.2f | " % (pi, (int(rows[b]['Start_Timestamp']) - t0) / 1e6, last) + " ".join("%s[%.1f-%.1f]" % e for e in ev))
