cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>gpurun_out/b1rank.err | tail -1 | cut -c1-260
