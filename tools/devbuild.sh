#!/bin/bash
# Dev tool: build a development library (fast build: the 25x25 variants only) AND its device ISA, from any directory.
#   tools/devbuild.sh NAME [extra -D flags]   ->  gym_cloth_amd/libx_NAME.so, /tmp/x_NAME.s, /tmp/k_NAME.s (the headline kernel)
# KERNEL (mangled-name fragment) selects the kernel cut out into /tmp/k_NAME.s; FULL=1 builds every variant.
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
KERNEL=${KERNEL:-k_run_scheduleIfLi512ELi2ELi2ELb1ELi1E}
F="-O3 -std=c++17 -fPIC -ffp-contract=off -w $*"
[ -z "${FULL:-}" ] && F="$F -DCLOTHHIP_FAST_BUILD"
cd "$ROOT/gym_cloth_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -S --cuda-device-only -o /tmp/x_$NAME.s clothhip_api.hip &
/opt/rocm/bin/hipcc --offload-arch=gfx950 $F -shared -o ../libx_$NAME.so clothhip_api.hip || { wait; echo BUILD FAILED; exit 1; }
wait
awk -v k="$KERNEL" 'index($0, "_ZN8clothhip14" k) == 1 && /:/ {p=1} p && /s_endpgm/ {p=0} p' /tmp/x_$NAME.s > /tmp/k_$NAME.s
echo "$NAME: $(wc -l < /tmp/k_$NAME.s) lines, scratch_load $(grep -c scratch_load /tmp/k_$NAME.s), scratch_store $(grep -c scratch_store /tmp/k_$NAME.s), s_barrier $(grep -c s_barrier /tmp/k_$NAME.s)"
grep -A30 "^\s*.amdhsa_kernel _ZN8clothhip14$KERNEL" /tmp/x_$NAME.s | grep -E "next_free_vgpr|next_free_sgpr|private_segment_fixed_size" | tr -d '\t' | tr '\n' ' '; echo
