#!/bin/bash
# Collect the rocprofv3 evidence that profiles/ summarises (run on the GPU box from the repo root, e.g. through
# gpurun). One rocprofv3 run per counter group: --pmc is never combined with tracing. Outputs under gpurun_out/.
set -u
OUT=gpurun_out/prof_${1:-final}
mkdir -p "$OUT"
export TMPDIR=/tmp
T="timeout 300"
# 1. kernel trace + stats of the bench command
$T rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o run -- python3 bench.py > "$OUT/bench_traced.json" 2> "$OUT/trace.log"
# 2. the same command untraced (the number quoted)
$T python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"
# 3. HBM traffic counters, separate passes
for C in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o run -- python3 bench.py --no-cpu-baseline --steps 2 > "$OUT/pmc_$C.out" 2> "$OUT/pmc_$C.log"
done
# 4. instruction mix / wait counters over one real action split in six launches
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  $T rocprofv3 --pmc $C -d "$OUT/pmc_$N" -o run -- python3 tools/phase_profile.py > "$OUT/pmc_$N.out" 2> "$OUT/pmc_$N.log"
done
# 5. phase profiles (fp32, fp64) and per-cloth diagnostics
$T python3 tools/phase_profile.py > "$OUT/phase_f32.txt" 2>&1
$T python3 tools/phase_profile.py --precision f64 > "$OUT/phase_f64.txt" 2>&1
$T python3 tools/bench_diag.py > "$OUT/bench_diag.txt" 2>&1
find "$OUT" -name "*.csv" | head -40
