#!/bin/bash
# GPU box: chip-idle analysis of the pipelined bench, plain and under a ONE-rank RCCL group (environment-made: no launcher between
# rocprofv3 and python), timed region only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in plain group; do
  O=$R/gpurun_out/kt_rank1_$mode; rm -rf $O
  if [ $mode = group ]; then export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29877; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --gpus 1 --steps 10 --warmup 2 --timed-only > $R/gpurun_out/kt_rank1_$mode.log 2>&1
  echo "== $mode: $(grep 'timed region' $R/gpurun_out/kt_rank1_$mode.log)"
  python3 $R/tools/timeline_gaps.py $(ls $O/*/*kernel_trace.csv | head -1) 200
done
