#!/usr/bin/env python3
"""Per-env diagnostics of the bench workload (dev tool): which envs dominate the kernel time?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_cloth_amd.envs import ClothVecEnv

E = 512
cfg = bench.bench_cfg(25, 0.02)
env = ClothVecEnv(cfg, n_envs=E, precision="f32", consume_domrand_draws=False)
for e in range(E):
    env.np_randoms[e] = np.random.RandomState(1000 + e)
env.reset()
acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(4, 4)) for e in range(E)], axis=1)
for t in range(3):
    obs, rew, done, info = env.step(acts[t])
    st = env.batch.debug_stats()
    ex = env.last_executed
    lev = st[:, 2]
    cov, vinv, oob, tear = env.batch.metrics()
    pos = env.batch.positions()
    ext = np.abs(pos).max(axis=(1, 2))
    clk = st[:, 15].astype(float) * 1024.0           # shader clocks each cloth's schedule took
    order = np.argsort(-clk)[:8]
    print("   schedule clocks: max %.0f M  p99 %.0f M  p90 %.0f M  median %.0f M  (kernel %.1f ms -> %.2f GHz if the slowest cloth spans it)" %
          (clk.max() / 1e6, np.percentile(clk, 99) / 1e6, np.percentile(clk, 90) / 1e6, np.median(clk) / 1e6, env.batch.last_kernel_ms,
           clk.max() / env.batch.last_kernel_ms / 1e6))
    print("step %d kernel %.1f ms | executed: mean %.0f max %d zero %d | levels/substep: mean %.1f p90 %.1f max %.1f | oob %d tear %d" %
          (t, env.batch.last_kernel_ms, ex.mean(), ex.max(), (ex == 0).sum(),
           (lev / np.maximum(ex, 1)).mean(), np.percentile(lev / np.maximum(ex, 1), 90), (lev / np.maximum(ex, 1)).max(),
           oob.sum(), tear.sum()))
    for e in order:
        print("   env %3d %5.0f Mclk (%5.1f kclk/substep) executed %4d levels/substep %6.1f corrected/substep %5.1f iters_pull %3d oob %d tear %d max|coord| %.2f" %
              (e, clk[e] / 1e6, clk[e] / 1e3 / max(ex[e], 1), ex[e], lev[e] / max(ex[e], 1), st[e, 3] / max(ex[e], 1), env.last_iters_pull[e], oob[e], tear[e], ext[e]))
