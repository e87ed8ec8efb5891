# 16x16x32 (Gemm2QuadS) vs 32x32x16 (Gemm2QuadR) vs the 128x128 kernel, real epilogues, sustained load
run() { echo -n "$1 $2: "; env $3 A2=1 WARM=60 python tools/gemm_bench.py $4 60 2>/dev/null | grep -E "TFLOP|kernel" | tr '\n' ' '; echo; }
for v in "KNNSVC_QUAD=2 KNNSVC_QUAD16=1" "KNNSVC_QUAD=2 KNNSVC_QUAD16=0" "KNNSVC_QUAD=0"; do
  run "FFN2 resid     " "[$v]" "RESID=1 $v" "31500 1024 4096"
  run "FFN1 gelu+split" "[$v]" "ACT=gelu OSPLIT=1 $v" "31500 4096 1024"
  run "QKV            " "[$v]" "X=1 $v" "31500 3072 1024"
done
