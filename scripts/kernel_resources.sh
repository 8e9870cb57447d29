#!/bin/bash
# VGPRs / scratch / occupancy of every kernel of one translation unit (cross-compiled, no GPU needed):
#   scripts/kernel_resources.sh p7x_envelope.hip
cd "$(dirname "$0")/../pyhmmer_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -I ../../include -c "$1" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | sed 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | awk '
  /Function Name:/ { name=$NF }
  / VGPRs:/ { v=$NF }
  /ScratchSize/ { s=$NF }
  /Occupancy/ { o=$NF }
  /SGPRs Spill/ { ss=$NF }
  /LDS Size/ { printf "%-64s vgpr %4s scratch %5s sgpr-spill %4s occ %s\n", name, v, s, ss, o }' | c++filt 2>/dev/null | sed 's/p7x:://; s/(p7x::ArgRef)//'
