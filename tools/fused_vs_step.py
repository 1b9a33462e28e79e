#!/usr/bin/env python3
"""The same pick-and-place action on E identical cloths through the two kernel variants (dev tool, GPU box): the per-step path
(ClothVecEnv.step: the plain stepper) and the episode launch (step_many: the fused stepper), kernel time per substep. Identical
cloths run in lock step, so this isolates what the episode variant's code costs from what the bench's workload mix costs."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gym_cloth_amd.envs import ClothVecEnv  # noqa: E402

E = int(os.environ.get("ENVS", "512"))
cfg = bench.bench_cfg(25, 0.02)
acts = np.random.RandomState(5).uniform(-1, 1, size=(6, 4))
acts[:, :2] *= 0.6                                      # pick points on the cloth
for mode in ("step", "fused"):
    env = ClothVecEnv(cfg, n_envs=E, precision=os.environ.get("PREC", "f32"), consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000)  # identical cloths
    env.reset()
    for t in range(len(acts)):
        a = np.repeat(acts[t][None, :], E, axis=0)
        if mode == "step":
            env.step(a, auto_reset=False)
            n = int(env.last_executed[0])
        else:
            out = env.step_many(a[None], auto_reset=False)
            n = int(out["executed"][0, 0])
        ms = env.batch.last_kernel_ms
        if n:
            print("%-5s action %d: %5d substeps %8.2f ms kernel -> %6.2f us/substep (%5.2f M substeps/s)" %
                  (mode, t, n, ms, ms * 1e3 / n, E * n / ms / 1e3), flush=True)
    env.close()
