#!/bin/bash
# specialised (KNNSVC_QUAD_EPI=1, default) against generic (=0) epilogue of the quad kernel on the encoder's shapes, sustained load
run() { echo -n "$1 [$2]: "; env $2 $3 A2=1 WARM=60 KNNSVC_QUAD=2 python tools/gemm_bench.py $4 60 2>/dev/null | grep -E "TFLOP|kernel" | tr '\n' ' '; echo; }
for cfg in "KNNSVC_QUAD_EPI=0" "KNNSVC_QUAD_EPI=1"; do
  run "FFN1 gelu+split" "$cfg" "ACT=gelu OSPLIT=1" "31500 4096 1024"
  run "FFN2 resid     " "$cfg" "RESID=1" "31500 1024 4096"
  run "QKV            " "$cfg" "X=1" "31500 3072 1024"
  run "out-proj resid " "$cfg" "RESID=1" "31500 1024 1024"
  run "conv stack     " "$cfg" "X=1" "48007 512 1536"
done
