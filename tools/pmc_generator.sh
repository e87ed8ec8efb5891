#!/bin/bash
# GPU box: HBM traffic of the generator's kernels from the PMC counters (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, as
# MI355X_MICROARCH.md prescribes) next to their algorithmic bytes and their durations from a plain kernel trace.
#   bash tools/pmc_generator.sh r04      -> gpurun_out/pmcgen_r04/{r04_pmc_generator.json, r04_pmc_synth.json}
tag=${1:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmcgen_$tag; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/pmc_generator.py pmcgen_$tag > $O/kt.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 $R/tools/pmc_generator.py pmcgen_$tag > $O/pf.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 $R/tools/pmc_generator.py pmcgen_$tag > $O/pw.log 2>&1 &&
python3 $R/tools/pmc_generator_report.py $O $tag
