# end-to-end bench under the dispatch variants of the 16x16x32 quad kernel
cd $GRAFT_REPO_ROOT
for v in "KNNSVC_QUAD16=0" "KNNSVC_QUAD16=1" "KNNSVC_QUAD16=1 KNNSVC_QUAD_KMIN=1024" "KNNSVC_QUAD16=1 KNNSVC_QUAD_KMIN=1024 KNNSVC_QUAD_GELU=1" "KNNSVC_QUAD16=1 KNNSVC_QUAD_KMIN=512"; do
  echo -n "[$v] "; env $v python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['config']['sequential_ms_per_step'])"
done
