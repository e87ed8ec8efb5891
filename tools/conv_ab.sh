set -e
run() { for s in "120000 128 128 3 1" "120000 128 128 11 5" "240000 64 64 3 1" "480000 32 32 3 1" "15000 256 256 3 1"; do
  timeout -k 10 60 python tools/conv_bench.py $s $1 2>&1 | grep -v amdgpu | tail -1; done; }
cp knn_svc_amd/libknnsvc_hip.so /tmp/base.so
echo "== base resid=0"; run 0
cp knn_svc_amd/libknnsvc_hip_ROT.so knn_svc_amd/libknnsvc_hip.so
echo "== ROT resid=0"; run 0
cp knn_svc_amd/libknnsvc_hip_NOMFMA.so knn_svc_amd/libknnsvc_hip.so
echo "== NOMFMA resid=0"; run 0
