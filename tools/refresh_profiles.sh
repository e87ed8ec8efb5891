#!/bin/bash
# Run ON THE GPU BOX (gpurun): regenerates everything under profiles/ for the current round tag ($1, e.g. r01).
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/profiles_$tag; mkdir -p $O
cd $R
if [ "${PART:-all}" != "2" ]; then
timeout 900 python3 bench.py --steps 10 --warmup 2 --stages > $O/bench.out 2> $O/bench.err
timeout 900 python3 bench.py > $O/${tag}_bench_default_line.json 2> /dev/null
tail -1 $O/bench.out > $O/${tag}_bench_n1.json
grep '^\[stage\]' $O/bench.err > $O/${tag}_bench_n1_stages.txt
grep '^\[gemm\]' $O/bench.err > $O/${tag}_bench_n1_gemm_shapes.txt
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs > $O/kt.log 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/${tag}_bench_n1_kernel_stats.csv
cp $(ls $O/kt/*/*domain_stats.csv | head -1) $O/${tag}_bench_n1_domain_stats.csv
# the same command with conversions run one at a time: per-kernel durations without cross-conversion overlap
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kts -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs --pipeline-depth 1 > $O/kts.log 2>&1
cp $(ls $O/kts/*/*kernel_stats.csv | head -1) $O/${tag}_bench_n1_sequential_kernel_stats.csv
# the counter passes run the SAME command line as the driver's bench (default K = 10, W = 2): the launch mix the bench line reports on
timeout 1100 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $O/pmcf.log 2>&1
timeout 1100 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw -- python3 $R/bench.py --no-cpu-baseline --no-other-configs > $O/pmcw.log 2>&1
python3 $R/tools/pmc_traffic.py $(ls $O/pmcf/*/*counter_collection.csv | head -1) $(ls $O/pmcw/*/*counter_collection.csv | head -1) "conv_gemm2quad_kernel<Gemm2QuadS" > $O/${tag}_pmc_traffic.json
python3 $R/tools/pmc_traffic.py $(ls $O/pmcf/*/*counter_collection.csv | head -1) $(ls $O/pmcw/*/*counter_collection.csv | head -1) > $O/${tag}_pmc_traffic_f128.json
rm -rf $O/kt $O/kts $O/pmcf $O/pmcw
fi
cd $R
[ "${PART:-all}" = "1" ] && { ls -la $O; tail -1 $O/bench.out | cut -c1-400; cat $O/${tag}_pmc_traffic.json; exit 0; }
A2=1 KNNSVC_QUAD=0 tools/pmc_cycles.sh f128_$tag 31500 4096 1024 5 > $O/${tag}_pmc_cycles_f128.txt 2>&1
A2=1 KNNSVC_QUAD=2 tools/pmc_cycles.sh quad_$tag 31500 1024 4096 5 > $O/${tag}_pmc_cycles_quad.txt 2>&1
python3 tools/knn_bench.py > $O/${tag}_knn_bench.txt 2>&1
KNNSVC_KNN_FUSED=0 python3 tools/knn_bench.py 2>&1 | tail -1 >> $O/${tag}_knn_bench.txt
bash tools/quad_fastepi_ab.sh 2>&1 | grep -v amdgpu > $O/${tag}_quad_specialised_vs_generic_epilogue.txt
# (libknnsvc_prof.so = -DKN_QUAD_PROF -DKN_KNN_PROF, built in the build container: make -C knn_svc_amd/csrc BUILD=build_prof
#  OUT=../libknnsvc_prof.so PROBE=../libknnsvc_prof_probe.so EXTRA="-DKN_QUAD_PROF -DKN_KNN_PROF")
bash tools/quad_prof.sh 2>&1 | grep -v amdgpu > $O/${tag}_quad_phase_trace.txt
( export KNNSVC_LIB=$R/knn_svc_amd/libknnsvc_prof.so; for a in "bench" "1500 30000" "300 30000" "24000 180000"; do python3 tools/knn_prof.py $a 2>&1 | grep -v amdgpu; done ) > $O/${tag}_knn_phase_trace.txt
bash tools/knn_trace2.sh 1500 30000 kt_$tag > $O/${tag}_knn_kernel_timeline_1500x30000.txt 2>&1
bash tools/pmc_generator.sh $tag > $O/${tag}_pmc_generator_summary.txt 2>&1; cp $R/gpurun_out/pmcgen_$tag/${tag}_pmc_*.json $O/
python3 tools/concat_race.py 40 2>&1 | grep -v amdgpu > $O/${tag}_concat_race_product.txt
KNNSVC_LIB=$R/knn_svc_amd/libknnsvc_slpprobe.so python3 tools/concat_race.py 40 2>&1 | grep -v amdgpu > $O/${tag}_concat_race_slp_probe.txt
python3 tools/gemm_zero.py 31500 1024 4096 > $O/${tag}_gemm_zero_vs_random.txt 2>&1
python3 tools/layer_error.py 6 6 > $O/${tag}_layer_error.txt 2>&1
( python3 tools/vocoder_replay.py; KNNSVC_RANGE_SLOTS=0 python3 tools/vocoder_replay.py ) > $O/${tag}_vocoder_range_slots.txt 2>&1
python3 tools/cfg5_bench.py --sources 32 --pool-minutes 60 --reps 3 2>/dev/null | tail -1 > $O/${tag}_cfg5_share_1gpu.json
python3 tools/cfg5_product_bench.py 2>/dev/null | tail -1 > $O/${tag}_cfg5_share_product_entry_1gpu.json
timeout 600 python3 tools/cfg3_bench.py --speakers 4 --utts 80 2>/dev/null | tail -1 > $O/${tag}_cfg3_1gpu.json
rm -rf $O/kt $O/kts $O/pmcf $O/pmcw
ls -la $O
tail -1 $O/bench.out | cut -c1-400
cat $O/${tag}_pmc_traffic.json
