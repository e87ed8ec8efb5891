# same-box A/B of the bench line: the round-4 tree (git worktree add ab_r04 e419d31 && make -C ab_r04/knn_svc_amd/csrc) against this tree
set -e
B="--no-cpu-baseline --no-other-configs"
P='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], d["value"], d["ms_per_step"], d["config"]["sequential_ms_per_step"], d["roofline"]["frac"], d["knn"]["ms"])'
for i in 1 2; do
  if [ -d ab_r04 ]; then (cd ab_r04 && timeout -k 10 400 python bench.py $B > ../gpurun_out/ab_r04_$i.json 2>/dev/null); python -c "$P" gpurun_out/ab_r04_$i.json; fi
  timeout -k 10 400 python bench.py $B > gpurun_out/ab_r05_$i.json 2>/dev/null; python -c "$P" gpurun_out/ab_r05_$i.json
done
KNNSVC_MERGE_BRANCHES=0 timeout -k 10 400 python bench.py $B > gpurun_out/ab_r05_serial_branches.json 2>/dev/null; python -c "$P" gpurun_out/ab_r05_serial_branches.json
