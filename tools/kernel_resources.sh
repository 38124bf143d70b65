#!/bin/bash
# Register / LDS / scratch usage of every kernel of one translation unit (compile-only, no GPU needed):
#   tools/kernel_resources.sh xcorr_ws32 [extra hipcc flags]
cd "$(dirname "$0")/../torchpiv_amd/csrc" || exit 1
unit=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=fast-honor-pragmas -fno-slp-vectorize \
  -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-result "$@" \
  -Rpass-analysis=kernel-resource-usage -c $unit.hip -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|AGPRs|Spill|ScratchSize|LDS Size|Occupancy" |
  sed -e 's/.*remark: [^ ]* *//' | paste - - - - - - - - | sed -e 's/\[-Rpass-analysis=kernel-resource-usage\]//g' | cut -c1-260
