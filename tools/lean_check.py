#!/usr/bin/env python3
"""LEAN stepper variant against the standard fp32 variant (dev tool, GPU box): same seeds, same actions -> bit-identical records
and particles (the two differ in where they keep the gather stencil and the rest lengths, not in any arithmetic)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from gym_cloth_amd.envs import ClothVecEnv  # noqa: E402

E, T = int(os.environ.get("ENVS", "64")), 6
outs = {}
for lean in ("0", "1"):
    os.environ["CLOTHHIP_DEBUG_LEAN"] = lean
    env = ClothVecEnv(bench.bench_cfg(25, 0.02), n_envs=E, precision="f32", consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000 + e)
    env.reset()
    acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1)
    out = env.step_many(acts, auto_reset=True)
    a = np.stack([np.random.RandomState(3000 + e).uniform(-1, 1, size=4) for e in range(E)])
    o2 = env.step(a, auto_reset=False)                         # the per-step path (plain stepper variant) as well
    outs[lean] = (out["rew"].copy(), out["executed"].copy(), out["obs"].copy(), o2[0].copy(), o2[1].copy(), env.last_executed.copy())
    env.close()
ok = all(np.array_equal(x, y) for x, y in zip(outs["0"], outs["1"]))
print("substeps", int(outs["0"][1].sum()), "+", int(outs["0"][5].sum()), "| lean == standard:", ok)
sys.exit(0 if ok else 1)
