#!/usr/bin/env python3
"""Dev tool (GPU box): run a fixed episode workload (tests/test_gpu_lean.py's) with whatever library / variant the environment
selects (CLOTHHIP_LIB, CLOTHHIP_DEBUG_LEAN, ...) and dump records + final particles to an .npz; `--cmp a.npz b.npz` compares two
dumps (bit-identity, or max |delta| per array).   tools/variant_dump.py out.npz [tier] [E] [T]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = True
    for k in a.files:
        if np.array_equal(a[k], b[k], equal_nan=True):
            continue
        same = False
        x, y = a[k].astype(np.float64), b[k].astype(np.float64)
        print("  %-10s differs: max |delta| %.3e (of max |value| %.3e), %d of %d entries" %
              (k, np.nanmax(np.abs(x - y)), np.nanmax(np.abs(x)), int((x != y).sum()), x.size))
    print("%s vs %s: %s" % (sys.argv[2], sys.argv[3], "BIT-IDENTICAL" if same else "DIFFERENT"))
    sys.exit(0)

import bench
from gym_cloth_amd.envs import ClothVecEnv

tier = sys.argv[2] if len(sys.argv) > 2 else "tier1"
E = int(sys.argv[3]) if len(sys.argv) > 3 else 48
T = int(sys.argv[4]) if len(sys.argv) > 4 else 5
prec = os.environ.get("PREC", "f32")
env = ClothVecEnv(bench.bench_cfg(25, 0.02, tier), n_envs=E, precision=prec, consume_domrand_draws=False)
for e in range(E):
    env.np_randoms[e] = np.random.RandomState(1000 + e)
env.reset()
acts = np.ascontiguousarray(np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1))
out = env.step_many(acts, auto_reset=True)
a = np.stack([np.random.RandomState(3000 + e).uniform(-1, 1, size=4) for e in range(E)])
obs, rew, done, info = env.step(a, auto_reset=False)
st = env.batch.get_state()
np.savez(sys.argv[1], rew=out["rew"], executed=out["executed"], done=out["done"], cov=out["actual_coverage"], obs=out["obs"],
         obs2=obs, rew2=rew, exec2=env.last_executed, pos=st[0], prev=st[1], pin=st[2])
print("dumped %s: %d action substeps, %d per-step substeps" % (sys.argv[1], int(out["executed"].sum()), int(env.last_executed.sum())))
env.close()
