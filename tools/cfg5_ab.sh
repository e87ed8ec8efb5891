for v in "X=1" "KNNSVC_QUAD_MIN_TILES=448" "KNNSVC_KNN_FUSED=0" "KNNSVC_KNN_PIPE_BLOCKS=0" "KNNSVC_KNN_PIPE_BLOCKS=128" "KNNSVC_KNN_GROUP_FRAMES=6000"; do
  echo -n "[$v] "; env $v python tools/cfg5_bench.py --sources 32 --pool-minutes 60 --reps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['xRT'], d['ms_per_source'])"
done
