#!/bin/bash
# GPU box: per-kernel durations of ONE kNN search size (rocprofv3 --kernel-trace): tools/knn_trace2.sh NQ NP OUT
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/$3; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/tools/knn_prof_one.py $1 $2 12 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/*/*kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
starts=[i for i,n in enumerate(names) if "row_norms" in n]
i0=starts[-2]
t0=int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    n=r["Kernel_Name"].split("(")[0][-60:]
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} us  +{(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid {int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])):6d}  {n}')
PY
