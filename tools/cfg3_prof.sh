cd $GRAFT_REPO_ROOT
python -m cProfile -o gpurun_out/cfg3.prof tools/cfg3_bench.py --speakers 3 --utts 60 --passes 2 2>/dev/null | cut -c1-300
python - <<'PY'
import pstats
p = pstats.Stats("gpurun_out/cfg3.prof"); p.sort_stats("tottime").print_stats(28)
PY
