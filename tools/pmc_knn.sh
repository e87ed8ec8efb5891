#!/bin/bash
# GPU box: counted MFMA utilisation of the kNN's kernels at one search size (VERDICT r4 #2d).
#   tools/pmc_knn.sh <tag> [NQ NP]      ->  gpurun_out/<tag>_pmc_knn.txt / .json
# ONE rocprofv3 --pmc pass (with --kernel-trace only; the program goes directly after --) over tools/knn_prof_one.py.
# utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x shader cycles), shader cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs;
# MI355X_MICROARCH.md "DVFS give-back").
tag=$1; NQ=${2:-1500}; NP=${3:-30000}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_knn_$tag; rm -rf $O; mkdir -p $O
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY \
    --kernel-trace --output-format csv -d $O -- python3 $R/tools/knn_prof_one.py $NQ $NP 8 > $O/run.log 2>&1
python3 - "$O" "$R/gpurun_out/${tag}_pmc_knn" $NQ $NP <<'PY'
import csv, glob, sys, collections, json, re
root, out, nq, npool = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{root}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "knn_" not in k and "row_norms" not in k and "split" not in k: continue
        m = re.search(r"(knn_\w+|row_norms_kernel|\w*split\w*)(<[^>]*>)?", k)
        k = (m.group(1) + (m.group(2) or "")) if m else k[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
dur = collections.defaultdict(list)
for f in glob.glob(f"{root}/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(knn_\w+|row_norms_kernel|\w*split\w*)(<[^>]*>)?", r["Kernel_Name"])
        if m: dur[m.group(1) + (m.group(2) or "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {"nq": nq, "np": npool, "kernels": {}}
lines = [f"kNN search {nq} x {npool}: counters per launch (rocprofv3 --pmc, one pass), utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)"]
for k, d in sorted(acc.items()):
    per = {c: v / cnt[(k, c)] for c, v in d.items()}
    util = per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (128.0 * per["GRBM_GUI_ACTIVE"]) if per.get("GRBM_GUI_ACTIVE") else None
    us = sorted(dur.get(k, [0.0]))
    med = us[len(us) // 2]
    clk = per.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / med / 1e3 if med else None
    res["kernels"][k] = dict(per, launches=cnt[(k, "GRBM_GUI_ACTIVE")], median_us=med, mfma_util=util, clock_ghz=clk)
    lines.append(f"{k}\n   launches {cnt[(k, 'GRBM_GUI_ACTIVE')]}, median {med:.1f} us (under the profiler), shader clock ~{clk:.2f} GHz, MFMA utilisation {util:.3f}")
    for c, v in sorted(per.items()):
        lines.append(f"   {c:28s} {v:.4g}")
scr = [v for k, v in res["kernels"].items() if k.startswith("knn_screen")]
if scr:
    busy = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"] * v["launches"] for v in scr); act = sum(v["GRBM_GUI_ACTIVE"] * v["launches"] for v in scr)
    res["screen_mfma_util"] = busy / (128.0 * act)
    allk = list(res["kernels"].values())
    res["search_mfma_util"] = busy / (128.0 * sum(v["GRBM_GUI_ACTIVE"] * v["launches"] for v in allk))
    lines.append(f"knn_screen_kernel (all epochs): MFMA utilisation {res['screen_mfma_util']:.3f}; over every kernel of the search (norms, splits, screens, refines, re-score): {res['search_mfma_util']:.3f}")
open(out + ".txt", "w").write("\n".join(lines) + "\n"); json.dump(res, open(out + ".json", "w"), indent=1)
print("\n".join(lines))
PY
