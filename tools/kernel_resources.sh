#!/bin/bash
# Compact per-kernel register / spill / scratch report for the gfx950 build (dev tool).
SRC=${1:-/root/repo/gym_cloth_amd/csrc/clothhip_api.hip}
mkdir -p /tmp/clothhip_rsrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -Rpass-analysis=kernel-resource-usage \
      -c "$SRC" -o /tmp/clothhip_rsrc/x.o 2>&1 |
  grep -E "Function Name|Name:| VGPRs:|AGPRs:|VGPRs Spill|ScratchSize" |
  sed -E 's/^[^ ]+ remark: +//; s/ \[-Rpass.*$//' |
  awk '/Name:/ {if (line) print line; line=$NF; next} {gsub(/^ +/,""); line=line "  " $0} END {print line}' |
  while read -r name rest; do printf "%-62s %s\n" "$(echo "$name" | c++filt | cut -c1-60)" "$rest"; done
