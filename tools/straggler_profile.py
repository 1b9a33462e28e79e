#!/usr/bin/env python3
"""Replicate the heaviest env of the bench workload over all E cloths and profile its action phase by phase (dev tool)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_cloth_amd import make_schedules
from gym_cloth_amd.envs import ClothVecEnv

E = 512
cfg = bench.bench_cfg(25, 0.02)
env = ClothVecEnv(cfg, n_envs=E, precision="f32", consume_domrand_draws=False)
for e in range(E):
    env.np_randoms[e] = np.random.RandomState(1000 + e)
env.reset()
acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(4, 4)) for e in range(E)], axis=1)
env.step(acts[0])
pos0, prev0, pin0 = env.batch.get_state()
env.step(acts[1])
st = env.batch.debug_stats()
ex = env.last_executed
w = int(np.argmax(np.where(ex > 0, st[:, 15], 0)))   # the cloth whose schedule took the most shader clocks
print("worst env", w, "executed", ex[w], "levels/substep %.1f" % (st[w, 2] / ex[w]), "iters_pull", env.last_iters_pull[w])
b = env.batch
b.set_state(pos0[w], prev0[w], pin0[w] * 0)
d = env.decode_actions(np.tile(acts[1][w], (E, 1)))
n = b.grab_top(np.stack([d["x"], d["y"]], 1))
bd = d["bounds"][0]
ip = int(d["iters_pull"][0])
phases = [("lift", 50, dict(n_up_end=50, n_uprest_end=50, n_pull_end=50, n_griprest_end=50, n_total=50, dz_up=0.0025)),
          ("uprest", 80, dict(n_griprest_end=80, n_total=80)),
          ("pull", ip, dict(n_pull_end=ip, n_griprest_end=ip, n_total=ip, dx_pull=d["x_dir_r"][0], dy_pull=d["y_dir_r"][0])),
          ("griprest", 300, dict(n_griprest_end=300, n_total=300)),
          ("rest0-200", 200, dict(n_total=200)), ("rest200-1000", 800, dict(n_total=800))]
tot = 0.0
for name, nn, f in phases:
    if nn == 0:
        continue
    s = make_schedules(E, active=1, break_on_tear=1, **f)
    b.run(s)
    ms = b.last_kernel_ms
    q = b.debug_stats()[0] / float(nn)
    tot += ms
    print("%-13s %4d substeps %8.2f ms  %7.2f us/substep | sweeps %.2f dense %.2f levels %6.1f corrected %5.1f" %
          (name, nn, ms, ms * 1e3 / nn, q[0], q[1], q[2], q[3]))
    if q[4:14].sum() > 0:       # stamps build (make -C gym_cloth_amd/csrc stamps) + CLOTHHIP_DEBUG_PHASES=47
        names = ["adjust", "hooke", "insert", "ranges", "fill", "precheck", "cells-wait", "reset+plane", "prepass", "sweep", "big cells", "small cells"]
        print("      cycles/substep: " + "  ".join("%s %.0f" % (n, q[4 + i] * 64) for i, n in enumerate(names)))
print("total %.1f ms" % tot)
