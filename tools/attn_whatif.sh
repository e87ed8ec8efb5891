# Where the attention kernel's time goes: what-if builds (timing aids, wrong results) next to the product library.
#   cd knn_svc_amd/csrc && for v in NOBIAS NOEXP NOS NOPV NOSTAGE; do make BUILD=build_att_$v OUT=../libknnsvc_att_$v.so EXTRA=-DKN_ATT_$v; done
cd $GRAFT_REPO_ROOT
echo "product:"; python tools/attn_bench.py 2>&1 | grep "pre-split"
for v in NOBIAS NOEXP NOS NOPV NOSTAGE; do
  echo "$v:"; KNNSVC_LIB=$GRAFT_REPO_ROOT/knn_svc_amd/libknnsvc_att_$v.so python tools/attn_bench.py 2>&1 | grep "pre-split"
done
