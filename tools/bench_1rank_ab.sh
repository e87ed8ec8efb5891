#!/bin/bash
# GPU box: bench.py under a ONE-rank RCCL group (the code path of every N > 1 run) for several environment variants, against the plain
# N = 1 run: how much of the stream pipeline's overlap survives the process group's extra streams.
O=$GRAFT_REPO_ROOT/gpurun_out/rank1_ab; mkdir -p $O
cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  f=$(echo "$name" | tr -c 'A-Za-z0-9_=' '_')
  env "$@" timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
      bench.py --gpus 1 --no-cpu-baseline --no-other-configs > "$O/$f.out" 2> "$O/$f.err"
  python3 - "$O/$f.out" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:32s} {d['value']:8.1f} xRT  {d['ms_per_step']:7.2f} ms/step  sequential {d['config']['sequential_ms_per_step']:6.2f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-other-configs > $O/plain.out 2> $O/plain.err
python3 -c "
import json; d=json.loads(open('$O/plain.out').read().strip().splitlines()[-1]); print(f\"{'plain (no process group)':32s} {d['value']:8.1f} xRT  {d['ms_per_step']:7.2f} ms/step  sequential {d['config']['sequential_ms_per_step']:6.2f}\")"
run "rccl group, default" X=1
for v in "$@"; do run "rccl group, $v" $v; done
