// cloth_kernels.hpp -- gfx950 device code of the cloth stepper.
//
// One workgroup steps ONE cloth through a whole schedule (up to ~2000 substeps) with the particle state
// resident in LDS; HBM is touched once on entry and once on exit.  Every phase keeps the reference's
// evaluation order (cloth.pyx:169-214), so the double instantiation reproduces the reference bit for bit
// (compiled with -ffp-contract=off) and the float instantiation is the same algorithm in fp32.
//
//   phase            reference                parallelisation (exact-order preserving)
//   adjust/release   gripper.pyx:55-73        per point (owner thread)
//   gravity+Hooke    cloth.pyx:216-237        per-point gather of <=12 springs in ascending list index
//   Verlet           cloth.pyx:239-256        per point (fused with the gather)
//   spatial map      cloth.pyx:298-311        LDS hash table keyed by the exact cell key + occupied-cell list; members
//                                             of a cell stored contiguously (rank from the counting atomic)
//   self-collision   cloth.pyx:313-343        parallel seed test, then an exact Gauss-Seidel sweep of the cells that
//                                             have a seed: to-visit set = seeds + later neighbours of members that moved;
//                                             four small cells per wave / one wave per large cell, LDS tickets
//   plane            cloth.pyx:345-370        per point (owner thread: it holds the previous position)
//   strain limit     cloth.pyx:258-296        dependency levels packed into 64-slot windows (one spring per lane, lane order ==
//                                             level order) walked by ONE wave from the first over-stretched spring to the last
//                                             window its corrections can reach; a pass finishes every spring none of whose
//                                             predecessors in the window (static per-slot dependency masks) is over-stretched
// DESIGN.md section 4 has the exactness argument of every phase.
// The code is split by phase: cloth_common.hpp (shared types and wave primitives), phase_strain.hpp, phase_collide.hpp,
// cloth_metrics.hpp, episode_loop.hpp (k_run_schedule), cloth_aux_kernels.hpp.
#pragma once

#include "cloth_common.hpp"
#include "phase_strain.hpp"
#include "phase_collide.hpp"
#include "cloth_metrics.hpp"
#include "episode_loop.hpp"
#include "cloth_aux_kernels.hpp"
