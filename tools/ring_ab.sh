set -e
for shape in "31500 4096 1024" "31500 1024 4096" "31500 3072 1024" "31500 1024 1024" "1500 30000 1024"; do
  echo "== $shape"
  for r in ${RINGS:-0 1 3 4}; do
    echo -n "ring=$r  "; A2=1 KNNSVC_RING=$r timeout -k 10 120 python tools/gemm_bench.py $shape 30 2>&1 | grep -E "TFLOP|err" | tr '\n' ' '; echo
  done
done
