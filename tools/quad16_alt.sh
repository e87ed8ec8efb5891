# A/B of an alternative library build (knn_svc_amd/libknnsvc_alt.so) on the quad shapes: correctness first, then speed
cd $GRAFT_REPO_ROOT
KNNSVC_LIB=$PWD/knn_svc_amd/libknnsvc_alt.so timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "quad" 2>&1 | tail -2 || exit 1
run() { echo -n "$1 $2: "; env $3 A2=1 WARM=60 python tools/gemm_bench.py $4 60 2>/dev/null | grep -E "TFLOP|kernel" | tr '\n' ' '; echo; }
for v in "KNNSVC_LIB=$PWD/knn_svc_amd/libknnsvc_alt.so" "KNNSVC_QUAD=1"; do
  run "FFN2 resid     " "[$v]" "RESID=1 $v" "31500 1024 4096"
  run "FFN1 gelu+split" "[$v]" "ACT=gelu OSPLIT=1 $v" "31500 4096 1024"
  run "QKV            " "[$v]" "X=1 $v" "31500 3072 1024"
  run "conv k3 s2     " "[$v]" "$v" "504000 512 1536"
done
