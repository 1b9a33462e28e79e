#!/bin/bash
# Dev tool: build a development library (fast build: the 25x25-class variants only) under a name, from any directory.
#   tools/devbuild.sh NAME [extra -D flags]   ->  gym_cloth_amd/libx_NAME.so          (FULL=1: every variant)
# and the resource usage + ISA of the headline kernel: /tmp/k_NAME.s (KERNEL="float, 512, 2, 2, true, 1" selects another instantiation)
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
KERNEL=${KERNEL:-float, 512, 2, 2, true, 1}
cd "$ROOT/gym_cloth_amd/csrc" || exit 1
FL=fast; LIB=../libclothhip_fast.so
[ -n "${FULL:-}" ] && { FL=prod; LIB=../libclothhip.so.dev; }
rm -rf obj_$FL
if [ -n "${FULL:-}" ]; then
  make -j8 -s ../libclothhip.so EXTRA="$*" 2>&1 | grep -i "error\|warning" | head; cp ../libclothhip.so ../libx_$NAME.so
else
  make -s fast EXTRA="$*" 2>&1 | grep -i "error\|warning" | head; cp ../libclothhip_fast.so ../libx_$NAME.so
fi
rm -rf obj_$FL                      # (the next plain `make` must not pick up objects built with other flags)
printf '#include <hip/hip_runtime.h>\n#include "episode_loop.hpp"\ntemplate __global__ void clothhip::k_run_schedule<%s>(clothhip::StepArgs<%s>);\n' "$KERNEL" "${KERNEL%%,*}" > /tmp/one_$NAME.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -w -I. $* --cuda-device-only -S -Rpass-analysis=kernel-resource-usage -o /tmp/k_$NAME.s /tmp/one_$NAME.hip 2>&1 |
  grep -E " VGPRs:|VGPRs Spill|ScratchSize|SGPRs Spill" | sed -E 's/^.*remark: +//; s/ \[-Rpass.*$//' | tr '\n' ' '
echo; echo "$NAME: $(wc -l < /tmp/k_$NAME.s) lines, scratch_load $(grep -c scratch_load /tmp/k_$NAME.s), scratch_store $(grep -c scratch_store /tmp/k_$NAME.s), s_barrier $(grep -c s_barrier /tmp/k_$NAME.s)"
