"""Timeline of ONE generator graph replay from a rocprofv3 kernel trace of tools/vocoder_replay.py: per stage (delimited by the
mean3 kernels) wall time, sum of kernel durations and the gaps in which no kernel runs.
    python tools/voc_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:48])
        for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last complete replay: from the last additive_synth before the final 'conv ... tanh' to the end
synth = [i for i, r in enumerate(rows) if "additive_synth" in r[2]]
i0 = synth[-1]
# the head fork starts before the synthesiser: back up to the previous kernel gap > 20 us
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 20000 and "additive_synth" not in rows[i0 - 1][2]:
    i0 -= 1
seg = rows[i0:]
t0 = seg[0][0]
marks = [(0, "start")] + [((e - t0) / 1e3, "mean3") for s, e, n in seg if "mean3" in n] + [((seg[-1][1] - t0) / 1e3, "end")]
print(f"one replay: {(seg[-1][1] - t0) / 1e3:.1f} us, {len(seg)} kernels")
prev = 0.0
for t, name in marks[1:]:
    ks = [(s, e, n) for s, e, n in seg if (s - t0) / 1e3 >= prev - 1e-9 and (e - t0) / 1e3 <= t + 1e-9]
    busy = 0; cur_s = cur_e = None
    for s, e, n in sorted(ks):
        if cur_e is None: cur_s, cur_e = s, e
        elif s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    if cur_e is not None: busy += cur_e - cur_s
    print(f"  until {name:6s} at {t:8.1f} us: wall {t - prev:7.1f} us, {len(ks):3d} kernels, sum of durations {sum(e - s for s, e, n in ks) / 1e3:8.1f} us, some kernel running {busy / 1e3:7.1f} us")
    prev = t
