# A/B of the windowed kernel's weight prefetch depth on the shapes that take the 64-row tile (KNNSVC_WIN_DEEP=0: one step ahead)
cd $GRAFT_REPO_ROOT
for shape in "15000 256 256 11 1" "15000 256 256 7 3" "15000 256 256 3 1" "3750 256 256 11 1" "3750 256 256 3 5" "30000 128 128 11 1" "30000 128 128 3 1"; do
  for d in 0 1; do
    echo -n "DEEP=$d : "; KNNSVC_WIN_DEEP=$d python tools/conv_bench.py $shape 1 50 2>/dev/null | tail -1
  done
done
