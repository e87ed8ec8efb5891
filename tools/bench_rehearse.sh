#!/bin/bash
# Run ON THE (one-GPU) BOX: bench.py's N > 1 paths with every rank on cuda:0 (gloo, host-staged collectives).  Not a measurement —
# it shows that the weak- and strong-scaling code paths run end to end before the driver runs them on a real node.
N=${1:-2}
O=$GRAFT_REPO_ROOT/gpurun_out/rehearse; mkdir -p $O
cd $GRAFT_REPO_ROOT
export KNNSVC_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for mode in weak strong; do
  timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 500)) \
      bench.py --gpus $N --steps 2 --warmup 1 --scaling $mode --no-cpu-baseline --no-other-configs > $O/bench_${mode}_n$N.out 2> $O/bench_${mode}_n$N.err \
      && tail -1 $O/bench_${mode}_n$N.out | cut -c1-600 || { echo "$mode FAILED"; tail -20 $O/bench_${mode}_n$N.err; exit 1; }
done
