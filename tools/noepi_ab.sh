set -e
run() { for shape in "31500 4096 1024" "31500 1024 4096" "31500 1024 1024"; do A2=1 timeout -k 10 120 python tools/gemm_bench.py $shape 30 2>&1 | grep "TFLOP"; done
  for s in "120000 128 128 3 1" "240000 64 64 3 1" "480000 32 32 3 1" "120000 128 128 11 5"; do timeout -k 10 60 python tools/conv_bench.py $s 1 2>&1 | grep -v amdgpu | tail -1; done; }
echo "== base"; run
cp knn_svc_amd/libknnsvc_hip_ne.so knn_svc_amd/libknnsvc_hip.so; echo "== no epilogue"; run
