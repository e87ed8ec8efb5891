#!/bin/bash
# usage: tools/pmc_conv.sh <tag> <conv_bench args...>   (GPU box).  rocprofv3 --pmc passes over tools/conv_bench.py
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
            "SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM"; do
  n=$(echo $pass | md5sum | cut -c1-6)
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/tools/${PMC_TOOL:-conv_bench.py} "$@" > $R/gpurun_out/pmc_${tag}_$n.log 2>&1
done
python3 - "$R/gpurun_out" "$tag" <<'PY'
import csv, glob, sys, collections, os
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(f"{root}/pmc_{tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if os.environ.get("KFILTER", "conv_gemm") not in k: continue
        k = k.split("(")[0][-60:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(tag, k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} {v / cnt[(k, c)]:.4g} per launch")
PY
