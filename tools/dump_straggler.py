#!/usr/bin/env python3
"""Dump the start state + action of the heaviest env of the bench workload (dev tool, GPU): gpurun_out/straggler.npz."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_cloth_amd.envs import ClothVecEnv

E = 512
cfg = bench.bench_cfg(25, 0.02)
env = ClothVecEnv(cfg, n_envs=E, precision="f64", consume_domrand_draws=False)
for e in range(E):
    env.np_randoms[e] = np.random.RandomState(1000 + e)
env.reset()
acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(4, 4)) for e in range(E)], axis=1)
env.step(acts[0])
pos0, prev0, pin0 = env.batch.get_state()
env.step(acts[1])
st = env.batch.debug_stats()
ex = env.last_executed
order = np.argsort(-np.where(ex > 0, st[:, 2], 0))[:4]
d = env.decode_actions(acts[1])
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed("gpurun_out/straggler.npz", envs=order, pos=pos0[order], prev=prev0[order], pin=pin0[order],
                    x=d["x"][order], y=d["y"][order], bounds=d["bounds"][order], dx=d["x_dir_r"][order], dy=d["y_dir_r"][order],
                    executed=ex[order])
print("dumped envs", order, "executed", ex[order])
