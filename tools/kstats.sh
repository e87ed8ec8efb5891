#!/bin/bash
# kernel-time table of a command on the GPU box:  tools/kstats.sh <divide_by> <top> python3 tools/x.py args...
# (durations summed over the run and divided by <divide_by>, e.g. the iteration count)
div=$1; top=$2; shift 2
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/kstats_tmp; rm -rf $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- "$@" > $R/gpurun_out/kstats.log 2>&1
python3 $R/tools/kstats.py $O $div $top
rm -rf $O
