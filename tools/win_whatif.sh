# Where a windowed convolution's time goes: what-if builds (timing aids, wrong results) next to the product library.
#   cd knn_svc_amd/csrc && make BUILD=build_win_NOBLOAD OUT=../libknnsvc_win_NOBLOAD.so EXTRA=-DKN_WIN_NOBLOAD
#                          make BUILD=build_win_NOBAR OUT=../libknnsvc_win_NOBAR.so EXTRA="-DKN_WIN_NOBLOAD -DKN_WIN_NOBAR"
#                          make BUILD=build_win_NOMFMA OUT=../libknnsvc_win_NOMFMA.so EXTRA=-DKN_WHATIF_NOMFMA
cd $GRAFT_REPO_ROOT
for shape in "15000 256 256 11 1" "15000 256 256 3 1" "120000 128 128 11 1" "120000 128 128 3 1" "120000 128 128 7 3"; do
  echo "== $shape"
  echo -n "product : "; python tools/conv_bench.py $shape 1 50 2>/dev/null | tail -1
  for v in NOBLOAD NOBAR NOMFMA; do
    echo -n "$v : "; KNNSVC_LIB=$GRAFT_REPO_ROOT/knn_svc_amd/libknnsvc_win_$v.so python tools/conv_bench.py $shape 1 50 2>/dev/null | tail -1
  done
done
