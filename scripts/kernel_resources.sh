#!/bin/bash
# Registers, spills, scratch and occupancy of every kernel in a source file of raytracing_amd/csrc, as the compiler
# reports them (-Rpass-analysis=kernel-resource-usage).  usage: kernel_resources.sh [file.hip] [grep pattern] [extra flags]
cd "$(dirname "$0")/../raytracing_amd/csrc"
/opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC -ffp-contract=off -fno-slp-vectorize --offload-arch=gfx950 -Wno-unused-function ${3:-} \
  -Rpass-analysis=kernel-resource-usage -c "${1:-kernels.hip}" -o /dev/null 2>&1 |
  grep -E "Function Name|VGPRs:|Spill|Occupancy|SGPRs:|LDS Size|ScratchSize" | paste - - - - - - - - |
  sed 's/[^ ]*\.hip:[0-9]*:[0-9]*: remark: //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g' | awk '{$1=$1};1' | grep -E "${2:-.}"
