# A/B of the quad kernel (KNNSVC_QUAD=2) against the 128x128 kernel (KNNSVC_QUAD=0) on the encoder's big shapes, A2 input.
# LIB=path selects an A/B build of the library (e.g. the main-loop-only what-if build).
set -e
for shape in "31500 4096 1024" "31500 1024 4096" "31500 3072 1024" "31500 1024 1024" "1008651 512 1536"; do
  echo "== $shape"
  for q in 0 2; do
    echo -n "quad=$q  "; KNNSVC_LIB=${LIB:+$PWD/$LIB} A2=1 KNNSVC_QUAD=$q timeout -k 10 120 python tools/gemm_bench.py $shape 20 2>&1 | grep -E "TFLOP|err" | tr '\n' ' '; echo
  done
done
