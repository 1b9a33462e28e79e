#!/bin/bash
# Compact per-kernel register / spill / scratch report for the gfx950 build (dev tool; needs no GPU): every stepper variant (the nine
# instantiation groups of stepper_inst.hip, compiled in parallel) and the small kernels of clothhip_api.hip.
#   bash tools/kernel_resources.sh > profiles/r06_kernel_resources.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/gym_cloth_amd/csrc" || exit 1
T=/tmp/clothhip_rsrc; mkdir -p $T
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w $EXTRA -Rpass-analysis=kernel-resource-usage --cuda-device-only"
for g in 0 1 2 3 4 5 6 7 8; do /opt/rocm/bin/hipcc $F -DCLOTHHIP_INST_GROUP=$g -c stepper_inst.hip -o $T/g$g.o > $T/g$g.log 2>&1 & done
/opt/rocm/bin/hipcc $F -c clothhip_api.hip -o $T/api.o > $T/api.log 2>&1 &
wait
cat $T/g?.log $T/api.log |
  grep -E "Function Name|Name:| VGPRs:|AGPRs:|VGPRs Spill|ScratchSize" |
  sed -E 's/^[^ ]+ remark: +//; s/ \[-Rpass.*$//' |
  awk '/Name:/ {if (line) print line; line=$NF; next} {gsub(/^ +/,""); line=line "  " $0} END {print line}' |
  while read -r name rest; do printf "%-70s %s\n" "$(echo "$name" | c++filt | cut -c1-68)" "$rest"; done | sort
