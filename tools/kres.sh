#!/bin/bash
# usage: tools/kres.sh <file.hip>   -> one line per kernel: name, VGPRs, spills, LDS, occupancy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I "$(dirname "$0")/../include" -c "$1" -o /tmp/kres.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Function Name|    VGPRs:|VGPRs Spill|LDS Size|Occupancy" \
  | sed 's/\[-Rpass-analysis=kernel-resource-usage\]//g; s/.*remark: *//' | paste - - - - - | sort -u
