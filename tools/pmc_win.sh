#!/bin/bash
# GPU box: where the generator's windowed convolutions spend their cycles — LDS conflicts, waits, instruction mix, per kernel.
#   tools/pmc_win.sh <tag>      ->  gpurun_out/<tag>_pmc_win.txt
# Separate rocprofv3 --pmc passes (with --kernel-trace only; the program goes directly after --) over tools/vocoder_replay.py.
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_win_$tag; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
           "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAIT_ANY SQ_LEVEL_WAVES"; do
    i=$((i + 1))
    timeout -k 10 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/vocoder_replay.py > $O/run$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/run$i.log; }
done
python3 - "$O" "$R/gpurun_out/${tag}_pmc_win.txt" <<'PY'
import csv, glob, sys, collections, re
root, out = sys.argv[1], sys.argv[2]
def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"(\w+)(<.*>)?\(", k)
    return (m.group(1) + (m.group(2) or "")) if m else k[:80]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
for f in glob.glob(f"{root}/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if "conv_" not in k: continue
        g = r.get("Grid_Size", "")
        k = f"{k} grid {g}"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
lines = ["generator replay at 1500 frames: counters per launch (rocprofv3 --pmc, six passes), summed over the chip"]
for k, d in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0.0) / max(1, cnt[(kv[0], "SQ_BUSY_CYCLES")]) * cnt[(kv[0], "SQ_BUSY_CYCLES")]):
    per = {c: v / cnt[(k, c)] for c, v in d.items()}
    n = cnt[(k, "GRBM_GUI_ACTIVE")]
    lines.append(f"{k}   x {n}")
    g = per.get("GRBM_GUI_ACTIVE", 0.0)
    if g:
        cyc = g / 8.0
        lines.append(f"   shader cycles per launch {cyc:.4g}; MFMA busy / (1024 SIMD x cycles) = {per.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (128 * g):.3f}")
    wc = per.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        lines.append(f"   of wave-cycles: waiting for any instruction {per.get('SQ_WAIT_INST_ANY', 0) / wc:.3f}, for LDS {per.get('SQ_WAIT_INST_LDS', 0) / wc:.3f}; LDS instruction active {per.get('SQ_ACTIVE_INST_LDS', 0) / wc:.3f}")
    ia = per.get("SQ_LDS_IDX_ACTIVE", 0.0)
    if ia:
        lines.append(f"   LDS: bank-conflict cycles / active cycles = {per.get('SQ_LDS_BANK_CONFLICT', 0) / ia:.3f}; active cycles per CU / shader cycles = {ia / 256 / (g / 8.0) if g else 0:.3f}")
    for c, v in sorted(per.items()):
        lines.append(f"      {c:28s} {v:.4g}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:80]))
PY
