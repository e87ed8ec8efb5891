cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/kt_voc; rm -rf $O
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $GRAFT_REPO_ROOT/tools/vocoder_replay.py > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/voc_kernels.py $(ls $O/*/*kernel_trace.csv | head -1)
