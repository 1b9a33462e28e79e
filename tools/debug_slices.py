#!/usr/bin/env python3
"""dev tool: find where time-sliced launches diverge from one unsliced launch (env by env, action by action)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_cloth_amd.envs import ClothVecEnv

def mk(E):
    v = ClothVecEnv(bench.bench_cfg(25, 0.02), n_envs=E, precision="f64", consume_domrand_draws=False)
    for e in range(E):
        v.np_randoms[e] = np.random.RandomState(1000 + e)
    v.reset()
    return v

E, N = 4, 5
budget = float(os.environ.get("BUDGET", "25"))
streams = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(N, 4)) for e in range(E)])
a = mk(E)
ref = a.step_many(np.ascontiguousarray(streams.transpose(1, 0, 2)), max_resets=8, want_obs=True)
b = mk(E)
cnt = np.zeros(E, dtype=np.int64)
slots = 4
for launch in range(400):
    if not (cnt < N).any():
        break
    idx = np.minimum(cnt[None, :] + np.arange(slots)[:, None], N - 1)
    tbl = streams[np.arange(E)[None, :], idx]
    out = b.step_many(tbl, max_resets=8, time_budget_ms=budget, want_obs=True)
    for e in range(E):
        n_e = int(out["ran"][:, e].sum())
        for t in range(n_e):
            k = int(cnt[e]) + t
            if k >= N:
                break
            same_obs = np.array_equal(out["obs_t"][t, e], ref["obs_t"][k, e])
            same_rew = out["rew"][t, e] == ref["rew"][k, e]
            tag = "" if (same_obs and same_rew) else "   <<<<<< DIFF obs=%s rew=%s (%.6f vs %.6f) cov %.6f vs %.6f start_cov %.6f vs %.6f" % (
                same_obs, same_rew, out["rew"][t, e], ref["rew"][k, e], out["actual_coverage"][t, e], ref["actual_coverage"][k, e],
                out["start_coverage"][t, e], ref["start_coverage"][k, e])
            print("launch %3d env %d action %d: executed %d (ref %d) reset_before %d (ref %d)%s" % (
                launch, e, k, out["executed"][t, e], ref["executed"][k, e], out["reset_before"][t, e], ref["reset_before"][k, e], tag))
        cnt[e] += n_e
print("launches", launch)
