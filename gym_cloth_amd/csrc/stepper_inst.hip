// stepper_inst.hip -- explicit instantiations of k_run_schedule, one GROUP of variants per object file (stepper_variants.hpp):
//   hipcc -DCLOTHHIP_INST_GROUP=g -c stepper_inst.hip -o inst_g.o      for g = 0 .. CLOTHHIP_INST_GROUPS - 1
#include <hip/hip_runtime.h>

#ifndef CLOTHHIP_INST_GROUP
#error "compile with -DCLOTHHIP_INST_GROUP=<group>"
#endif
#include "stepper_variants.hpp"

#if CLOTHHIP_INST_GROUP == 0
CLOTH_GROUP_0(CLOTH_DEFN)
#elif CLOTHHIP_INST_GROUP == 1
CLOTH_GROUP_1(CLOTH_DEFN)
#elif CLOTHHIP_INST_GROUP == 2
CLOTH_GROUP_2(CLOTH_DEFN)
#elif CLOTHHIP_INST_GROUP == 3
CLOTH_GROUP_3(CLOTH_DEFN)
#elif CLOTHHIP_INST_GROUP == 4
CLOTH_GROUP_4(CLOTH_DEFN)
CLOTH_RELAXED()
#elif CLOTHHIP_INST_GROUP == 5
CLOTH_GROUP_5(CLOTH_DEFN)
#elif CLOTHHIP_INST_GROUP == 6
CLOTH_SPEC_A(CLOTH_DEFN_S)
#elif CLOTHHIP_INST_GROUP == 7
CLOTH_SPEC_B(CLOTH_DEFN_S)
#elif CLOTHHIP_INST_GROUP == 8
#ifndef CLOTHHIP_FAST_BUILD
CLOTH_SPEC_C_F32(CLOTH_DEFN_S) CLOTH_SPEC_C_F64(CLOTH_DEFN_S)
#endif
#else
#error "unknown CLOTHHIP_INST_GROUP"
#endif
