#!/bin/bash
# GPU box: run the GPU parity suite against the four MUTANT builds (make -C gym_cloth_amd/csrc mutants: one ordering rule of the reference
# broken in each) and record which tests reject which mutant -> gpurun_out/mutants/summary.txt (copied to profiles/r06_mutation.txt).
#   bash tools/run_mutants.sh            every mutant must fail at least one test; a surviving mutant means a missing test
OUT=gpurun_out/mutants; mkdir -p $OUT
SEL='tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_lean.py tests/test_gpu_large2.py tests/test_gpu_soak.py tests/test_gpu_env.py tests/test_gpu_fused.py tests/test_gpu_refpins.py'
DESC=("" "1: ONE particle adds two of its incident springs in swapped list order (Hooke, cloth.pyx:221-237)"
         "2: the first two visits of a collision cell swapped (cloth.pyx:324-343)"
         "3: the second over-stretched spring of a pass committed with the first whether or not it depends on it (cloth.pyx:265-296)"
         "4: the both-pinned skip of the strain limit dropped (cloth.pyx:268)")
: > $OUT/summary.txt
echo "# pytest -m gpu (parity files) against libclothhip_mut{1..4}.so: tests that FAIL = tests that reject the mutant. f32-only = the failing tests whose id carries f32 / lean." >> $OUT/summary.txt
echo "# control: the production library passes all of them (GPUTEST)." >> $OUT/summary.txt
for k in 1 2 3 4; do
  lib=$PWD/gym_cloth_amd/libclothhip_mut$k.so
  [ -f $lib ] || { echo "mutant $k: library missing" >> $OUT/summary.txt; continue; }
  CLOTHHIP_LIB=$lib timeout -k 10 900 python -m pytest $SEL -m gpu -q -p no:cacheprovider -o addopts="" --timeout 600 > $OUT/mut$k.log 2>&1
  nf=$(grep -c "^FAILED" $OUT/mut$k.log); np=$(tail -1 $OUT/mut$k.log)
  echo "== mutant ${DESC[$k]}" >> $OUT/summary.txt
  echo "   $np" >> $OUT/summary.txt
  echo "   failing tests: $nf, of them fp32: $(grep "^FAILED" $OUT/mut$k.log | grep -ci "f32\|lean\|float")" >> $OUT/summary.txt
  grep "^FAILED" $OUT/mut$k.log | sed 's/ - .*//' | sed 's/^/     /' >> $OUT/summary.txt
done
cat $OUT/summary.txt
