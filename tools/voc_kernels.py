"""Per-kernel list of ONE generator graph replay (merged branch grids) from a rocprofv3 kernel trace of tools/vocoder_replay.py.
    python tools/voc_kernels.py <kernel_trace.csv>"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70],
         int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r.get("Grid_Size_Y", 1) or 1)) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
synth = [i for i, r in enumerate(rows) if "additive_synth" in r[2]]
i0 = synth[-1]
while i0 > 0 and rows[i0][0] - rows[i0 - 1][1] < 20000 and "additive_synth" not in rows[i0 - 1][2]:
    i0 -= 1
seg = rows[i0:]
t0 = seg[0][0]
print(f"one replay: {(seg[-1][1] - t0) / 1e3:.1f} us, {len(seg)} kernels")
for s, e, n, gx, gy in seg:
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f} us  grid {gx:6d} x {gy}  {n}")
