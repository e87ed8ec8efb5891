cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python bench.py > $O/r03_bench_default_line.json 2> /dev/null
python3 tools/cfg5_product_bench.py 2>/dev/null | tail -1 > $O/r03_cfg5_share_product_entry_1gpu.json
python3 tools/cfg5_bench.py --sources 32 --pool-minutes 60 --reps 3 2>/dev/null | tail -1 > $O/r03_cfg5_share_1gpu.json
timeout 600 python3 tools/cfg3_bench.py --speakers 4 --utts 80 2>/dev/null | tail -1 > $O/r03_cfg3_1gpu.json
(for q in 1 2; do echo QB=$q; KNNSVC_ATT_QB=$q python tools/attn_bench.py; KNNSVC_ATT_QB=$q python tools/attn_bench.py 3 777 16; done) 2>&1 | grep "QB\|checksum\|pre-split" > $O/r03_attention_qb.txt
(for t in 1 2 3 4; do echo "KNNSVC_TAILS=$t"; KNNSVC_TAILS=$t python3 tools/cfg5_product_bench.py 2>/dev/null | tail -1; KNNSVC_TAILS=$t timeout 600 python3 tools/cfg3_bench.py --speakers 4 --utts 80 2>/dev/null | tail -1 | cut -c1-330; done) > $O/r03_tail_streams_ab.txt 2>&1
MINWG=150 bash tools/timeline_cmd.sh 1000 python3 /root/repo/tools/cfg5_product_bench.py > $O/r03_cfg5_timeline_three_tails.txt 2>&1
