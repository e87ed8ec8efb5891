cd $GRAFT_REPO_ROOT
run() { echo -n "$1: "; env $2 A2=1 WARM=60 python tools/gemm_bench.py 31500 4096 1024 60 2>/dev/null | grep -E "TFLOP|kernel" | tr '\n' ' '; echo; }
run "FFN1 gelu+split" "ACT=gelu OSPLIT=1"
run "FFN1 split only" "OSPLIT=1"
run "FFN1 plain     " "X=1"
run "FFN1 gelu only " "ACT=gelu"
