for v in "X=1" "KNNSVC_KNN_FUSED=0" "KNNSVC_FUSED_PAIR=0" "KNNSVC_MATCH_LANES=1" "KNNSVC_KNN_GROUP_FRAMES=1000000000" "KNNSVC_KNN_PIPE_BLOCKS=0"; do
  echo "[$v]"; env $v NS=16 NP=60 python tools/determinism_pipeline.py 2>&1 | grep -E "vs" | cut -c1-150
done
