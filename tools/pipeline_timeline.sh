# GPU box: tools/pipeline_timeline.sh <tree dir> <tag>   (tree = . or ab_r04)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=$R/$1; O=$R/gpurun_out/kt_$2; rm -rf $O
cd $T && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $T/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --timed-only > /dev/null 2>&1
python3 $R/tools/pipeline_timeline.py $(ls $O/*/*kernel_trace.csv | head -1)
