#!/usr/bin/env python3
"""Where the time of the BENCH workload goes, phase by phase (dev tool, GPU box). Runs bench.py's fused workload (512 cloths,
tier-1 start, random actions, in-kernel episode resets) for a few time slices with the profiling build of the library and
sums the per-phase shader-cycle stamps of every cloth (wave 0's view):
    make -C gym_cloth_amd/csrc stamps
    CLOTHHIP_LIB=$PWD/gym_cloth_amd/libclothhip_stamps.so CLOTHHIP_DEBUG_PHASES=47 python tools/fused_profile.py
Without the stamps build it prints the balance of the launch only (per-cloth schedule clocks, stats[15])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_cloth_amd.envs import ClothVecEnv

E = int(os.environ.get("ENVS", "512"))
slice_ms = float(os.environ.get("SLICE_MS", "1000"))
launches = int(os.environ.get("LAUNCHES", "3"))
prec = os.environ.get("PREC", "f32")
cfg = bench.bench_cfg(25, 0.02)
env = ClothVecEnv(cfg, n_envs=E, precision=prec, consume_domrand_draws=False)
for e in range(E):
    env.np_randoms[e] = np.random.RandomState(1000 + e)
env.reset()
slots = 24
streams = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(slots * (launches + 2), 4)) for e in range(E)])
cnt = np.zeros(E, dtype=np.int64)
names = ["adjust", "hooke + verlet", "hash insert", "cell ranges", "cell fill", "collision pre-check", "cell sweeps: tickets/wait", "table reset + plane",
         "strain pre-pass", "strain sweep", "cell sweeps: big cells", "cell sweeps: small cells"]
tot_ph = np.zeros(12)
tot_sub = 0
for w in range(launches + 1):
    tbl = streams[np.arange(E)[None, :], cnt[None, :] + np.arange(slots)[:, None]]
    out = env.step_many(tbl, time_budget_ms=slice_ms, max_resets=12)
    cnt += out["ran"].sum(axis=0)
    st = env.batch.debug_stats().astype(np.float64)
    sub = int(out["executed"].sum() + out["reset_substeps"].sum() + out.get("tail_reset_substeps", np.zeros(1)).sum())
    kms = env.batch.last_kernel_ms
    print("launch %d: %.0f ms kernel, %d substeps -> %.2f M/s; env-steps %d, resets %d" %
          (w, kms, sub, sub / kms / 1e3, int(out["ran"].sum()), int((out["reset_before"] > 0).sum())))
    if w == 0:
        continue                                     # warm-up
    if st[:, 4:16].sum() > 0 and os.environ.get("CLOTHHIP_DEBUG_PHASES"):
        tot_ph += st[:, 4:16].sum(axis=0) * 64
        tot_sub += sub
    else:
        clk = st[:, 15] * 1024
        print("   per-cloth schedule clocks: min %.3g median %.3g mean %.3g max %.3g -> balance (mean/max) %.3f" %
              (clk.min(), np.median(clk), clk.mean(), clk.max(), clk.mean() / clk.max()))
if tot_sub:
    per = tot_ph / tot_sub
    total = per.sum()                                 # the twelve stamps partition wave 0's time in the substep loop
    print("cycles per cloth-substep, averaged over the whole workload (%d substeps): total %.0f" % (tot_sub, total))
    for n, v in zip(names, per):
        print("   %-28s %8.0f  %5.1f %%" % (n, v, 100 * v / total))
    print("   %-28s %8.0f  %5.1f %%" % ("(all cell sweeps)", per[6] + per[10] + per[11], 100 * (per[6] + per[10] + per[11]) / total))
    st = env.batch.debug_stats().astype(np.float64)
    print("strain sweep: %.3f sweeps/substep, windows walked/substep %.1f, passes/substep %.1f, corrected/substep %.1f" %
          (st[:, 0].sum() / tot_sub * launches, st[:, 1].sum() / tot_sub * launches, st[:, 2].sum() / tot_sub * launches,
           st[:, 3].sum() / tot_sub * launches))
env.close()
