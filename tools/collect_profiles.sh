#!/bin/bash
# Collect the rocprofv3 evidence that profiles/ summarises (run on the GPU box from the repo root, e.g. through
# gpurun). One rocprofv3 run per counter group: --pmc is never combined with tracing. Outputs under gpurun_out/.
#   bash tools/collect_profiles.sh r04          then, anywhere:  python tools/summarize_profiles.py gpurun_out/prof_r04 profiles/r04
set -u
OUT=gpurun_out/prof_${1:-final}
mkdir -p "$OUT"
export TMPDIR=/tmp
T="timeout -k 10 900"                      # (-k: a python child that ignores the signal is killed, not left holding the GPU)
HEAD="--no-extra --no-cpu-baseline"        # the headline configuration only (512 cloths, fp32, fused time slices)
# STAGES="bench trace traffic writerep l2 sq sq2 phases census action large n50 ablation" selects stages (default: all)
want() { [[ -z "${STAGES:-}" || " $STAGES " == *" $1 "* ]]; }
if want bench; then
# 0. the full default bench line (what the driver runs), untraced
$T python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.log"
fi
if want trace; then
# 1. kernel trace + stats of the headline bench command
$T rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o run -- python3 bench.py $HEAD > "$OUT/bench_traced.json" 2> "$OUT/trace.log"
fi
if want traffic; then
# 2. HBM traffic counters, separate passes, for every configuration bench.py reports a roofline for:
#    headline (tier 1), tier-2 companion, 50x50 companion. FETCH_SIZE three times for the headline: one dispatch in some passes
#    has shown a 10x fetch burst (profiles/README.md); TCC_EA0_RDREQ_sum is the request counter FETCH_SIZE derives from.
for C in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
  $T rocprofv3 --pmc $C -d "$OUT/pmc_$C" -o run -- python3 bench.py $HEAD > "$OUT/pmc_$C.out" 2> "$OUT/pmc_$C.log"
done
for K in 2 3; do
  $T rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmcrep${K}_FETCH_SIZE" -o run -- python3 bench.py $HEAD > "$OUT/pmcrep${K}_FETCH_SIZE.out" 2> "$OUT/pmcrep${K}_FETCH_SIZE.log"
done
fi
if want writerep; then
# the same burst has shown up in WRITE_SIZE: two more passes, and one with the request counter it derives from in the SAME pass
for K in 2 3; do
  $T rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmcrep${K}_WRITE_SIZE" -o run -- python3 bench.py $HEAD > "$OUT/pmcrep${K}_WRITE_SIZE.out" 2> "$OUT/pmcrep${K}_WRITE_SIZE.log"
done
$T rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum -d "$OUT/pmcx1_WRITE_SIZE_TCC_EA0_WRREQ_sum" -o run -- python3 bench.py $HEAD > "$OUT/pmcx1_WRITE_SIZE_TCC_EA0_WRREQ_sum.out" 2> "$OUT/pmcx1_WRITE_SIZE_TCC_EA0_WRREQ_sum.log"
fi
if want traffic; then
for C in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --pmc $C -d "$OUT/t2pmc_$C" -o run -- python3 bench.py $HEAD --init tier2 --steps 10 > "$OUT/t2pmc_$C.out" 2> "$OUT/t2pmc_$C.log"
  $T rocprofv3 --pmc $C -d "$OUT/n50pmc_$C" -o run -- python3 bench.py $HEAD --n-side 50 --envs 1024 --steps 5 --warmup 0 --fuse 5 --step-ms 250 > "$OUT/n50pmc_$C.out" 2> "$OUT/n50pmc_$C.log"
done
fi
if want l2; then
# 2b. L2 hits / misses beside the fetch counter over the headline command (the counter bursts of rounds 2 and 3: profiles/README.md)
$T rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d "$OUT/pmcl2_FETCH_SIZE_TCC" -o run -- python3 bench.py $HEAD > "$OUT/pmcl2.out" 2> "$OUT/pmcl2.log"
fi
if want sq; then
# 3. instruction mix / wait counters over the headline command
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  $T rocprofv3 --pmc $C -d "$OUT/sq_$N" -o run -- python3 bench.py $HEAD --steps 10 > "$OUT/sq_$N.out" 2> "$OUT/sq_$N.log"
done
fi
if want sq2; then
# 3b. latencies as the sequencer sees them: instruction fetch (LEVEL / count), vector memory, LDS; branches; cycles waiting for an instruction
for C in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_ANY" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | tr ' ' '_')
  $T rocprofv3 --pmc $C -d "$OUT/sq_$N" -o run -- python3 bench.py $HEAD --steps 10 > "$OUT/sq_$N.out" 2> "$OUT/sq_$N.log"
done
fi
if want phases; then
# 4. where the time of the bench workload goes: balance of a launch, then the per-phase stamps (profiling build)
$T python3 tools/fused_profile.py > "$OUT/fused_balance.txt" 2>&1
if [ -f gym_cloth_amd/libclothhip_stamps.so ]; then
  CLOTHHIP_LIB=$PWD/gym_cloth_amd/libclothhip_stamps.so CLOTHHIP_DEBUG_PHASES=47 $T python3 tools/fused_profile.py > "$OUT/fused_phases.txt" 2>&1
  CLOTHHIP_LIB=$PWD/gym_cloth_amd/libclothhip_stamps.so CLOTHHIP_DEBUG_PHASES=47 PREC=f64 SLICE_MS=1500 LAUNCHES=2 $T python3 tools/fused_profile.py > "$OUT/fused_phases_f64.txt" 2>&1
fi
fi
if want census; then
# 4b. census of the bench workload (counter build): active collision cells, frozen substeps, skippable windows, hits per visit
if [ -f gym_cloth_amd/libclothhip_cnt.so ]; then
  CLOTHHIP_LIB=$PWD/gym_cloth_amd/libclothhip_cnt.so $T python3 tools/cell_counters.py > "$OUT/census.txt" 2>&1
fi
fi
if want action; then
# 5. one real pick-and-place (the reference's oracle action) phase by phase, fp32 and fp64; the sweep's passes (sweep-stamps build)
$T python3 tools/phase_profile.py > "$OUT/phase_f32.txt" 2>&1
$T python3 tools/phase_profile.py --precision f64 > "$OUT/phase_f64.txt" 2>&1
if [ -f gym_cloth_amd/libclothhip_sweepstamps.so ]; then
  CLOTHHIP_LIB=$PWD/gym_cloth_amd/libclothhip_sweepstamps.so CLOTHHIP_DEBUG_PHASES=47 $T python3 tools/phase_profile.py > "$OUT/sweepstamps_f32.txt" 2>&1
fi
fi
if want large; then
# 5b. kernel trace of the large-batch companions: 1 536 cloths (six per CU, one generation) and 2 048 (four per CU: two generations, one dispatch each)
for E in 1536 2048; do
  $T rocprofv3 --kernel-trace --stats -d "$OUT/trace$E" -o run -- python3 bench.py $HEAD --envs $E --steps 10 --fuse 5 > "$OUT/bench${E}_traced.json" 2> "$OUT/trace$E.log"
done
fi
if want n50; then
# 6. rocprofv3 trace of the 50x50 companion (configs[4]) alone
$T rocprofv3 --kernel-trace --stats -d "$OUT/trace50" -o run -- python3 bench.py $HEAD --n-side 50 --envs 1024 --steps 5 --warmup 0 --fuse 5 --step-ms 250 > "$OUT/bench50_traced.json" 2> "$OUT/trace50.log"
fi
if want ablation; then
# 7. phase ablation on harvested reference states (settled / rest / pull / lift): all phases, without the strain limit, without
#    strain limit and self-collision, Hooke + Verlet only -- what the exact-order sweeps cost, and what is left without them
$T python3 tools/microbench.py --masks 15,7,5,1 > "$OUT/ablation.txt" 2>&1
fi
find "$OUT" -name "*.db" | head -60
