#!/usr/bin/env python3
"""dev tool: the same work (the reference's oracle action of seed 1337 on E cloths) through the plain per-step kernel and
through the episode kernel (one action slot, no reset): kernel milliseconds of both. Shows what the episode machinery costs
the substep loop (register pressure of the cold code)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle
from gym_cloth_amd.envs import ClothVecEnv

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_env import base_cfg

E = int(os.environ.get("ENVS", "512"))
prec = os.environ.get("PREC", "f32")
g = pyoracle.load_golden("g_env_tier1_1337.npz")
k = int(g["n_reset_calls"])
act = g["act"][k]
res = {}
for mode in ("plain", "fused", "plain", "fused"):
    v = ClothVecEnv(base_cfg("tier1", 1337), n_envs=E, precision=prec, consume_domrand_draws=False)
    v.batch.set_state(g["act_pos0"][k][None], g["act_prev0"][k][None], g["act_pin0"][k][None])
    cov, vinv, _, _ = v.batch.metrics()
    v._prev_reward[:] = cov; v._start_coverage[:] = cov; v._start_variance_inv[:] = vinv
    if mode == "plain":
        v.step(np.tile(act, (E, 1)))
        ex = v.last_executed
    else:
        out = v.step_many(np.tile(act, (1, E, 1)), auto_reset=False)
        ex = out["executed"][0]
    ms = v.batch.last_kernel_ms
    print("%-6s %s: %d substeps per cloth, kernel %.2f ms -> %.2f us/substep, %.2f M substeps/s" %
          (mode, prec, ex[0], ms, ms * 1e3 / ex[0], E * ex[0] / ms / 1e3))
    v.close()
