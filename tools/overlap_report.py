"""Read a rocprofv3 kernel_trace.csv and report, for each long single-workgroup kernel (concat / adam),
which other kernels ran while it was running (overlap in ms) — checks cross-stream concurrency."""
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0 = rows[0][0]
slow = [r for r in rows if "concat_reselect" in r[2] or "adam" in r[2]]
for s, e, name, q in slow[-8:]:
    ov = collections.Counter()
    for s2, e2, n2, q2 in rows:
        if q2 == q or e2 <= s or s2 >= e: continue
        ov[n2.split("(")[0][-40:]] += (min(e, e2) - max(s, s2)) / 1e6
    top = ", ".join(f"{k}:{v:.2f}" for k, v in ov.most_common(4))
    print(f"{name.split('(')[0][-32:]:32s} q{q} start {(s - t0) / 1e6:9.2f} ms dur {(e - s) / 1e6:6.2f} ms | overlapped: {top}")
