"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as MI355X_MICROARCH.md prescribes)
into per-launch HBM traffic of the dominant kernel.  FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at
64 B for wide streaming reads), WRITE_SIZE is taken as is; both are reported by rocprofv3 in KiB.

    python tools/pmc_traffic.py gpurun_out/pmcf/p_counter_collection.csv gpurun_out/pmcw/p_counter_collection.csv \
        > profiles/r01_pmc_traffic.json
"""
import csv, json, sys

def avg(path, counter, key):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"]
            if key in name:
                vals.setdefault(name, []).append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in vals.items()}

KEY = sys.argv[3] if len(sys.argv) > 3 else "conv_gemm2_kernel<Gemm2Tile<128, 128"
f = avg(sys.argv[1], "FETCH_SIZE", KEY)
w = avg(sys.argv[2], "WRITE_SIZE", KEY)
# every instantiation whose name matches (e.g. the A2 = true / false variants of the 128x128 kernel), launch-weighted
n = sum(c for _v, c in f.values())
fk = sum(v * c for v, c in f.values()) / n
wk = sum(v * c for v, c in w.values()) / max(1, sum(c for _v, c in w.values()))
out = dict(kernel=" | ".join(sorted(f)), launches=n, fetch_size_kib_avg=round(fk, 1), write_size_kib_avg=round(wk, 1),
           hbm_bytes_per_launch=int((2 * fk + wk) * 1024),
           note="2 x FETCH_SIZE + WRITE_SIZE, averaged over every launch of the kernel in a bench.py run (--steps 2 --warmup 1, plus the "
                "setup, latency and eager roofline passes bench.py adds); separate --pmc passes")
print(json.dumps(out, indent=1))
