#!/bin/bash
# Run ON THE GPU BOX: per-kernel times of the Harvest f0 front end on the two 60 s sample clips -> gpurun_out/harvest_prof/
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/harvest_prof; mkdir -p $O
cd $R
python3 tools/harvest_check.py > $O/${tag}_harvest_check.txt 2>&1
cd /tmp && export TMPDIR=/tmp
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 tools/harvest_check.py > $O/kt.log 2>&1
cp $(ls $O/kt/*/*kernel_stats.csv | head -1) $O/${tag}_harvest_kernel_stats.csv
rm -rf $O/kt
cat $O/${tag}_harvest_check.txt
head -20 $O/${tag}_harvest_kernel_stats.csv
