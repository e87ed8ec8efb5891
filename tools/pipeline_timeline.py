"""Kernel-trace view of the pipelined bench (bench.py, depth 2): per step, how long the encoder's and the generator's launches take
alone (sum of durations) and as spans, and how much of the generator ran beside the encoder.
    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --timed-only
    python tools/pipeline_timeline.py OUT/*/*kernel_trace.csv"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id") or r.get("Queue_Id") or "?"))
rows.sort()
by_stream = collections.Counter()
for s, e, n, q in rows:
    by_stream[q] += e - s
main = by_stream.most_common(1)[0][0]
MATCH = ("concat_reselect", "adam_reg", "gram_kernel", "f0_rerank", "log_f0", "shift_f0", "weighted_gather", "knn_", "row_norms", "split_weight2")
def kind(n, q):
    if q == main: return "front"
    return "match" if any(m in n for m in MATCH) else "voc"
starts = [s for s, e, n, q in rows if "conv0_ln_gelu" in n and q == main]
print(f"streams: {dict(by_stream.most_common(6))}; main = {main}; {len(starts)} steps seen")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    return tot + (ce - cs if cs is not None else 0)
for k in range(max(1, len(starts) - 5), len(starts) - 1):
    a, b = starts[k], starts[k + 1]
    win = [(s, e, n, q) for s, e, n, q in rows if s >= a and s < b]
    parts = collections.defaultdict(list)
    for s, e, n, q in win: parts[kind(n, q)].append((s, e))
    f, v, m = parts["front"], parts["voc"], parts["match"]
    span = lambda iv: (max(e for s, e in iv) - min(s for s, e in iv)) / 1e6 if iv else 0.0
    dur = lambda iv: sum(e - s for s, e in iv) / 1e6
    both = union(f + v) / 1e6
    print(f"step {k}: wall {(b - a) / 1e6:6.2f} ms | front: sum {dur(f):6.2f} busy {union(f) / 1e6:6.2f} | generator: {len(v)} launches sum {dur(v):5.2f} "
          f"busy {union(v) / 1e6:5.2f} span {span(v):5.2f} | match: sum {dur(m):5.2f} span {span(m):5.2f} | front U generator busy {both:6.2f} "
          f"(overlap {union(f) / 1e6 + union(v) / 1e6 - both:5.2f})")
