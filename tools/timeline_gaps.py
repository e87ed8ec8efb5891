"""Idle analysis of a rocprofv3 kernel trace of the pipelined bench: over the last WINDOW ms, the time during which no
chip-wide kernel (>= MINWG workgroups, default 256; env MINWG) is executing, and the largest such gaps with the kernels around them.
    python tools/timeline_gaps.py <kernel_trace.csv> [window_ms]"""
import csv, os, sys
MINWG = int(os.environ.get("MINWG", "256"))
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    wg = (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) * max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))) * \
         max(1, int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_Z"])))
    nm = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), nm.split("(")[0][:60], wg))
rows.sort()
win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 150e6
t1 = rows[-1][1]; t0 = t1 - win
wide = [(max(s, t0), e, n) for s, e, n, wg in rows if wg >= MINWG and e > t0]
busy = 0; cur_s = cur_e = None; gaps = []; last_name = None
for s, e, n in wide:
    if cur_e is None: cur_s, cur_e, last_name = s, e, n; continue
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append((s - cur_e, last_name, n, (cur_e - t0) / 1e6)); cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
    if e >= cur_e: last_name = n
busy += cur_e - cur_s
tot_wide = sum(e - s for s, e, n in wide)
print(f"window {win / 1e6:.1f} ms: some chip-wide kernel running {busy / 1e6:.2f} ms ({100 * busy / win:.1f} %), "
      f"sum of chip-wide kernel durations {tot_wide / 1e6:.2f} ms (concurrency {tot_wide / busy:.2f}x)")
print(f"idle of chip-wide work: {(win - busy) / 1e6:.2f} ms in {len(gaps)} gaps; largest:")
for g, a, b, at in sorted(gaps, reverse=True)[:12]:
    print(f"  {g / 1e3:8.1f} us at {at:8.2f} ms   after {a}   before {b}")
