// stepper_variants.hpp -- the compile-time variants of k_run_schedule (episode_loop.hpp) and the translation unit each one is compiled in.
//
// The library has ~80 instantiations of one kernel template; compiled in one translation unit they took three minutes. They are split into
// GROUPS, one object file each (stepper_inst.hip compiled with -DCLOTHHIP_INST_GROUP=g, in parallel by make); clothhip_api.hip sees them as
// `extern template` declarations and only takes their addresses (a kernel launch across translation units needs no relocatable device code:
// the host stub is an ordinary symbol, the device code is registered by the object that defines it).
//   X(T, NT, PPT, TAB, REST_REG): threads per cloth, particles per thread, table mode, rest lengths in registers / LEAN palette (episode_loop.hpp)
#pragma once

#include "episode_loop.hpp"

// standard arithmetic (fp32 and fp64): the 25x25 class, then the large grids
#define CLOTH_VARIANTS_SMALL(X, T) X(T, 512, 2, 1, false) X(T, 512, 2, 0, false) X(T, 256, 3, 1, true) X(T, 256, 3, 1, false) X(T, 256, 3, 0, false)
#ifdef CLOTHHIP_FAST_BUILD           // development builds: the 25x25-class variants only (make fast)
#define CLOTH_VARIANTS_LARGE(X, T)
#else
#define CLOTH_VARIANTS_LARGE(X, T) X(T, 512, 5, 0, false) X(T, 512, 5, 1, false) X(T, 1024, 3, 0, false) X(T, 1024, 4, 0, false)
#endif
#define CLOTH_VARIANTS(X, T) CLOTH_VARIANTS_SMALL(X, T) CLOTH_VARIANTS_LARGE(X, T)
// the LEAN builds (fp32 only; 25x25 class: three to six cloths per CU, and eight waves per cloth at two per CU; the large grids: the whole CU
// for a cloth, or two 512 x 5 cloths per CU)
#define CLOTH_VARIANTS_LEAN_SMALL(X, T) X(T, 256, 3, 0, true) X(T, 256, 3, -1, true) X(T, 256, 3, -2, true) X(T, 256, 3, -3, true) X(T, 512, 2, 2, true)
#ifdef CLOTHHIP_FAST_BUILD
#define CLOTH_VARIANTS_LEAN_LARGE(X, T)
#else
#define CLOTH_VARIANTS_LEAN_LARGE(X, T) X(T, 1024, 3, 3, true) X(T, 1024, 4, 3, true) X(T, 512, 5, 4, true)
#endif
#define CLOTH_VARIANTS_LEAN(X, T) CLOTH_VARIANTS_LEAN_SMALL(X, T) CLOTH_VARIANTS_LEAN_LARGE(X, T)
// the fp64 LEAN build (25x25 class, eight waves per cloth; rest lengths = palette value + per-spring ulp offset)
#define CLOTH_VARIANTS_LEAN64(X, T) X(T, 512, 2, 0, true)

// the grid-specialised builds (NS = 25 / 50: a BASELINE grid at compile time, cloth_common.hpp spec_*); XS(T, NT, PPT, TAB, RR, NS)
//   A, B: the fp32 LEAN variants of the 25x25 class (the headline; three to six cloths per CU)
//   C: tier 2 at 25x25 (configs[3]'s per-GPU shape: standard arithmetic, per-env rest tables), the fp64 LEAN build at 25x25, 50x50 at two cloths per CU (configs[4])
#define CLOTH_SPEC_A(XS) XS(float, 512, 2, 2, true, 25) XS(float, 256, 3, 0, true, 25) XS(float, 256, 3, -1, true, 25)
#define CLOTH_SPEC_B(XS) XS(float, 256, 3, -2, true, 25) XS(float, 256, 3, -3, true, 25)
#define CLOTH_SPEC_C_F32(XS) XS(float, 512, 2, 1, false, 25) XS(float, 512, 5, 4, true, 50)
#define CLOTH_SPEC_C_F64(XS) XS(double, 512, 2, 0, true, 25)
#ifdef CLOTHHIP_FAST_BUILD
#define CLOTH_SPEC_F32(XS) CLOTH_SPEC_A(XS) CLOTH_SPEC_B(XS)
#define CLOTH_SPEC_F64(XS)
#else
#define CLOTH_SPEC_F32(XS) CLOTH_SPEC_A(XS) CLOTH_SPEC_B(XS) CLOTH_SPEC_C_F32(XS)
#define CLOTH_SPEC_F64(XS) CLOTH_SPEC_C_F64(XS)
#endif

// every variant exists for FUSED = 0 (one external schedule), 1 (episodes, flat tiers), 2 (episodes incl. tier-2 resets and the cold policies)
#define CLOTH_FUSED3(KW, T, NT, PPT, TAB, RR)                                                        \
    KW template __global__ void clothhip::k_run_schedule<T, NT, PPT, TAB, RR, 0>(clothhip::StepArgs<T>);  \
    KW template __global__ void clothhip::k_run_schedule<T, NT, PPT, TAB, RR, 1>(clothhip::StepArgs<T>);  \
    KW template __global__ void clothhip::k_run_schedule<T, NT, PPT, TAB, RR, 2>(clothhip::StepArgs<T>);
#define CLOTH_DECL(T, NT, PPT, TAB, RR) CLOTH_FUSED3(extern, T, NT, PPT, TAB, RR)
#define CLOTH_DEFN(T, NT, PPT, TAB, RR) CLOTH_FUSED3(, T, NT, PPT, TAB, RR)
#define CLOTH_FUSED3_S(KW, T, NT, PPT, TAB, RR, NS_)                                                        \
    KW template __global__ void clothhip::k_run_schedule<T, NT, PPT, TAB, RR, 0, NS_>(clothhip::StepArgs<T>);  \
    KW template __global__ void clothhip::k_run_schedule<T, NT, PPT, TAB, RR, 1, NS_>(clothhip::StepArgs<T>);  \
    KW template __global__ void clothhip::k_run_schedule<T, NT, PPT, TAB, RR, 2, NS_>(clothhip::StepArgs<T>);
#define CLOTH_DECL_S(T, NT, PPT, TAB, RR, NS_) CLOTH_FUSED3_S(extern, T, NT, PPT, TAB, RR, NS_)
#define CLOTH_DEFN_S(T, NT, PPT, TAB, RR, NS_) CLOTH_FUSED3_S(, T, NT, PPT, TAB, RR, NS_)

// The groups (object files). CLOTHHIP_INST_GROUPS of them; stepper_inst.hip defines group CLOTHHIP_INST_GROUP, everybody else declares.
//   0 fp32 standard small   1 fp64 standard small   2 fp32 standard large   3 fp64 standard large   4 LEAN small (+ the relaxed-order companion)   5 LEAN large + fp64 LEAN   6, 7, 8 the grid-specialised builds (CLOTH_SPEC_A / _B / _C)
#define CLOTHHIP_INST_GROUPS 9
#define CLOTH_GROUP_0(M) CLOTH_VARIANTS_SMALL(M, float)
#define CLOTH_GROUP_1(M) CLOTH_VARIANTS_SMALL(M, double)
#define CLOTH_GROUP_2(M) CLOTH_VARIANTS_LARGE(M, float)
#define CLOTH_GROUP_3(M) CLOTH_VARIANTS_LARGE(M, double)
#define CLOTH_GROUP_4(M) CLOTH_VARIANTS_LEAN_SMALL(M, float)
#define CLOTH_GROUP_5(M) CLOTH_VARIANTS_LEAN_LARGE(M, float) CLOTH_VARIANTS_LEAN64(M, double)
#define CLOTH_RELAXED(KW) KW template __global__ void clothhip::k_run_schedule<float, 512, 2, 2, true, 3>(clothhip::StepArgs<float>);

#ifndef CLOTHHIP_INST_GROUP          // a user of the kernels (clothhip_api.hip): nothing is instantiated here
CLOTH_GROUP_0(CLOTH_DECL) CLOTH_GROUP_1(CLOTH_DECL) CLOTH_GROUP_2(CLOTH_DECL) CLOTH_GROUP_3(CLOTH_DECL) CLOTH_GROUP_4(CLOTH_DECL) CLOTH_GROUP_5(CLOTH_DECL)
CLOTH_SPEC_F32(CLOTH_DECL_S) CLOTH_SPEC_F64(CLOTH_DECL_S)
CLOTH_RELAXED(extern)
#endif
