"""Print the top kernels of a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [divide_by] [top]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = 0.0
rows = list(csv.DictReader(open(f)))
for r in rows:
    tot += float(r["TotalDurationNs"])
for r in rows[:top]:
    print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} us avg {float(r['TotalDurationNs']) / div / 1e3:10.1f} us/unit  {r['Name'][:100]}")
print(f"total {tot / div / 1e3:.1f} us/unit")
