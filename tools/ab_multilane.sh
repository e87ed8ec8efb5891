# same box: BASELINE cfg 5 share and cfg 3 through the product entries, round-4 tree against this tree
cd $GRAFT_REPO_ROOT
for t in ab_r04 . ab_r04 .; do
  [ -d $t ] || continue
  echo "== tree $t"; (cd $t && python3 tools/cfg5_product_bench.py 2>/dev/null | tail -1 | cut -c1-120; timeout 600 python3 tools/cfg3_bench.py --speakers 4 --utts 80 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"cfg3\", d[\"value\"], [p[\"xrt\"] for p in d[\"passes\"]])")
done
