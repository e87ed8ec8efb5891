#!/bin/bash
# Per-kernel register / LDS / spill figures of one HIP source, as the compiler reports them (no GPU needed):
#   tools/kernel_regs.sh knn_svc_amd/csrc/conv_gemm.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | grep "remark:" |
  sed -E 's/.*remark: +//; s/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/^Function Name/ {if (n) print n, v; n=$3; v=""; next} /^(VGPRs|AGPRs|ScratchSize|Occupancy|VGPRs Spill|LDS Size)/ {v=v " | " $0} END {print n, v}' |
  while read -r name rest; do echo "$(echo "$name" | c++filt | sed -E 's/\(anonymous namespace\):://g; s/\(ConvArgs\)//') $rest"; done
