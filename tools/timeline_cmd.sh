# GPU idle analysis (tools/timeline_gaps.py) of any command:  tools/timeline_cmd.sh <window_ms> python3 /root/repo/tools/x.py args...
win=$1; shift
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/kt_cmd; rm -rf $O
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O -- "$@" > $GRAFT_REPO_ROOT/gpurun_out/timeline_cmd.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline_gaps.py $(ls $O/*/*kernel_trace.csv | head -1) $win
rm -rf $O
