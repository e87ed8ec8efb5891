cd $GRAFT_REPO_ROOT
for L in 3 4 6 8; do
  echo -n "lanes $L cfg3: "; KNNSVC_MATCH_LANES=$L python tools/cfg3_bench.py --speakers 4 --utts 80 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print([p['xrt'] for p in d['passes']])"
  echo -n "lanes $L cfg5: "; KNNSVC_MATCH_LANES=$L python tools/cfg5_bench.py --sources 32 --pool-minutes 60 --reps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['xRT'], d['ms_per_source'])"
done
