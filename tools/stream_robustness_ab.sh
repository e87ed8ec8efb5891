#!/bin/bash
# GPU box: the bench line (K = 10, W = 2) with n unrelated streams created first, and under a ONE-rank RCCL process group (the code path
# of every N > 1 run): VERDICT r4 #3 asks that the figure does not depend on either.  tools/stream_robustness_ab.sh [tree]
T=${1:-.}
cd $GRAFT_REPO_ROOT/$T
B="--no-cpu-baseline --no-other-configs"
P='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print("%-44s %8.1f xRT  %7.2f ms/step  sequential %6.2f" % (sys.argv[2], d["value"], d["ms_per_step"], d["config"]["sequential_ms_per_step"]))'
O=$GRAFT_REPO_ROOT/gpurun_out/robust_$(echo $T | tr -c 'A-Za-z0-9' '_'); mkdir -p $O
for n in 0 1 2 3 4 7; do
  KNNSVC_BENCH_DUMMY_STREAMS=$n timeout -k 10 300 python3 bench.py $B > $O/d$n.out 2> $O/d$n.err; python3 -c "$P" $O/d$n.out "plain, $n dummy streams first"
done
for n in 0 2 3; do
  KNNSVC_BENCH_DUMMY_STREAMS=$n timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) \
      bench.py --gpus 1 $B > $O/r$n.out 2> $O/r$n.err; python3 -c "$P" $O/r$n.out "one-rank RCCL group, $n dummy streams first"
done
