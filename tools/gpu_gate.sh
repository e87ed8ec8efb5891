#!/bin/bash
# The gate to run AFTER the last commit of a session: the whole GPU suite exactly as the driver runs it.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/gpu_gate.sh'
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu 2>&1 | tee gpurun_out/gpu_gate.log
