#!/usr/bin/env python3
"""Counters of the self-collision phase over the bench workload (dev tool, GPU box). Needs the counter build:
    cd gym_cloth_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DCLOTHHIP_CELL_COUNTERS \
        -shared -o ../libclothhip_cnt.so clothhip_api.hip
    CLOTHHIP_LIB=$PWD/gym_cloth_amd/libclothhip_cnt.so python tools/cell_counters.py
Wave 0 of every cloth counts what IT did (about a quarter of the cells): big cells (> 16 members) swept, their members and
visits, small-cell tickets (up to four cells each), the pre-check's trip bound, active / occupied cells."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gym_cloth_amd.envs import ClothVecEnv  # noqa: E402

E = 512
env = ClothVecEnv(bench.bench_cfg(25, 0.02), n_envs=E, precision="f32")
env.seed(1000); env.reset()
rs = [np.random.RandomState(2000 + e) for e in range(E)]
tot = np.zeros(16); nsub = 0.0
for it in range(6):
    acts = np.stack([np.stack([r.uniform(-1, 1, 4) for r in rs]) for _ in range(40)])
    out = env.step_many(actions=acts, time_budget_ms=800.0)
    if it >= 1:
        tot += env.batch.debug_stats().astype(np.float64).sum(0)
        nsub += float(out["executed"].sum() + out["reset_substeps"].sum())
# stats[4 + q] = counter q of the kernel's tph[] array
big, nmem, vis, trips, smt, na, nocc = (tot[4 + q] for q in (4, 5, 6, 7, 8, 9, 10))
noseed, frozen, fixed, skipw, hits = (tot[4 + q] for q in (0, 1, 2, 3, 11))
print("census, fraction of all cloth-substeps: no active collision cell %.3f | no strain sweep %.3f | frozen (no adjust, every "
      "unpinned particle restored by the plane, no over-stretched spring) %.3f | no particle changed its collision cell since "
      "the substep before %.3f" % (noseed / nsub, 1.0 - tot[0] / nsub, frozen / nsub, fixed / nsub))
print("strain sweep per cloth-substep: windows walked %.1f, passes %.1f, correcting %.1f | windows without a flagged spring and beyond "
      "every correction's reach (skippable) %.1f" % (tot[1] / nsub, tot[2] / nsub, tot[3] / nsub, skipw / nsub))
print("big cells: hits per visit %.2f" % (hits / max(vis, 1)))
print("substeps %.0f | per cloth-substep: strain sweeps %.3f" % (nsub, tot[0] / nsub))
print("wave 0, per cloth-substep: big cells %.2f (mean %.1f members, %.1f visits each) | small-cell tickets %.2f | "
      "pre-check member bound %.1f | active cells %.1f of %.1f occupied" %
      (big / nsub, nmem / max(big, 1), vis / max(big, 1), smt / nsub, trips / nsub, na / nsub, nocc / nsub))
