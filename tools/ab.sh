#!/bin/bash
# A/B of production builds on the GPU box: runs the headline bench (no companions, no CPU baseline) for every library given,
# alternating, REPS times each, and prints value / kernel-only rate per run.
#   tools/ab.sh [REPS] lib1.so lib2.so ...        (paths relative to the repo root; extra bench flags through BENCH_FLAGS)
REPS=${1:-2}; shift
for r in $(seq 1 $REPS); do
  for lib in "$@"; do
    out=$(CLOTHHIP_LIB=$PWD/$lib python3 bench.py --no-extra --no-cpu-baseline $BENCH_FLAGS 2>&1 | tail -1)
    echo "$lib run $r: $(echo "$out" | python3 -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d["roofline"]
    print("value (actions only) %.3f M/s  blended %.3f M/s  blended kernel-only %.3f M/s  frac %.4f  steps_eq %.1f" % (d["value"]/1e6, d["config"]["blended_substeps_per_s"]/1e6, r["substeps_per_launch"]/r["kernel_ms_avg"]/1e3, r["frac"], d["config"]["steps_equivalent"]))
except Exception as e:
    print("FAILED", e)')"
  done
done
