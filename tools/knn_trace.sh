#!/bin/bash
# GPU box: per-kernel median durations of tools/knn_bench.py by grid size (rocprofv3 --kernel-trace)
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/knn_kt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/knn_kt -- python3 $GRAFT_REPO_ROOT/tools/knn_bench.py > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/knn_kt/*/*kernel_trace.csv")[0]
agg=collections.OrderedDict()
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    key=("gemm" if "conv_gemm2" in n else "select" if "knn_select" in n else "split" if "split_weight" in n else None)
    if not key: continue
    g=(key,int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])), r.get("VGPR_Count"), r.get("LDS_Block_Size"))
    agg.setdefault(g,[]).append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for g,v in agg.items():
    v=sorted(v); print(g, len(v), "median us", round(v[len(v)//2],1))
PY
