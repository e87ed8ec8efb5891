# dataset-mode kNN grouping (KNNSVC_KNN_GROUP_FRAMES): cfg 5 share and cfg 3 at several group sizes
cd $GRAFT_REPO_ROOT
for G in 1500 3000 8192 100000000; do
  echo -n "group $G cfg5: "; KNNSVC_KNN_GROUP_FRAMES=$G python tools/cfg5_bench.py --sources 32 --pool-minutes 60 --reps 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['xRT'], d['ms_per_source'])"
  echo -n "group $G cfg3: "; KNNSVC_KNN_GROUP_FRAMES=$G python tools/cfg3_bench.py --speakers 4 --utts 80 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print([p['xrt'] for p in d['passes']])"
done
