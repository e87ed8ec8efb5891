cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/kt_tl
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_tl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-configs --timed-only > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $(ls $GRAFT_REPO_ROOT/gpurun_out/kt_tl/*/*kernel_trace.csv | head -1) 300
