#!/bin/bash
# A/B of library builds on the states where the strain sweep's passes dominate (tools/microbench.py: harvested reference states, all phases):
# us per substep of the pulled / lifted / settled states, REPS times per library, alternating -- five times more sensitive to the sweep's pass
# loop than the bench workload.   tools/ab_micro.sh [REPS] lib1.so lib2.so ...
REPS=${1:-2}; shift
for r in $(seq 1 $REPS); do
  for lib in "$@"; do
    echo "$lib run $r: $(CLOTHHIP_LIB=$PWD/$lib python3 tools/microbench.py --masks 15 2>&1 | awk '/mask 15/ {printf "%s %s  ", $1, $4}')"
  done
done
