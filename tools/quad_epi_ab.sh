# quad vs 128x128 kernel on the encoder's shapes WITH their real epilogues, sustained load (WARM=60 launches before timing)
set -e
run() { echo -n "$1 quad=$2: "; env $3 A2=1 WARM=60 KNNSVC_QUAD=$2 python tools/gemm_bench.py $4 60 2>/dev/null | grep -E "TFLOP|kernel" | tr '\n' ' '; echo; }
for q in 2 0; do
  run "FFN1 gelu+split" $q "ACT=gelu OSPLIT=1" "31500 4096 1024"
  run "FFN2 resid     " $q "RESID=1" "31500 1024 4096"
  run "QKV            " $q "X=1" "31500 3072 1024"
  run "out-proj resid " $q "RESID=1" "31500 1024 1024"
done
