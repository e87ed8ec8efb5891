#!/bin/bash
# Run ON THE GPU BOX (gpurun): regenerates everything under profiles/ for the current round tag ($1, e.g. r01).
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/profiles_$tag; mkdir -p $O
cd $R
timeout 900 python3 bench.py --steps 10 --warmup 2 --stages > $O/bench.out 2> $O/bench.err
tail -1 $O/bench.out > $O/${tag}_bench_n1.json
grep '^\[stage\]' $O/bench.err > $O/${tag}_bench_n1_stages.txt
grep '^\[gemm\]' $O/bench.err > $O/${tag}_bench_n1_gemm_shapes.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/kt.log 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/${tag}_bench_n1_kernel_stats.csv
cp $(ls $O/kt/*/*domain_stats.csv | head -1) $O/${tag}_bench_n1_domain_stats.csv
# the same command with conversions run one at a time: per-kernel durations without cross-conversion overlap
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --pipeline-depth 1 > $O/kts.log 2>&1
cp $(ls $O/kts/*/*kernel_stats.csv | head -1) $O/${tag}_bench_n1_sequential_kernel_stats.csv
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmcf.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/pmcw.log 2>&1
python3 $R/tools/pmc_traffic.py $(ls $O/pmcf/*/*counter_collection.csv | head -1) $(ls $O/pmcw/*/*counter_collection.csv | head -1) > $O/${tag}_pmc_traffic.json
cd $R
A2=1 tools/pmc_cycles.sh f128_$tag 31500 4096 1024 5 > $O/${tag}_pmc_cycles_f128.txt 2>&1
rm -rf $O/kt $O/kts $O/pmcf $O/pmcw
ls -la $O
tail -1 $O/bench.out | cut -c1-400
cat $O/${tag}_pmc_traffic.json
