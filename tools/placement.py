#!/usr/bin/env python3
"""Where and when the workgroups of ONE time-sliced episode launch ran (dev tool, GPU box; needs a library built with
-DCLOTHHIP_DIAG_PLACEMENT: tools/devbuild.sh with FULL=1). Prints, per XCC, how many workgroups started in each slice-long
interval after the launch's first start, and how many distinct (SE, CU) pairs hosted them.
    CLOTHHIP_LIB=$PWD/gym_cloth_amd/libx_diag.so python tools/placement.py [n_side envs slice_ms]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gym_cloth_amd.envs import ClothVecEnv

n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 50
E = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
slice_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 800.0
cfg = bench.bench_cfg(n_side, 0.0095 if n_side == 50 else 0.02)
env = ClothVecEnv(cfg, n_envs=E, precision="f32", consume_domrand_draws=False)
for e in range(E):
    env.np_randoms[e] = np.random.RandomState(1000 + e)
env.reset()
slots = 24
tbl = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(slots, 4)) for e in range(E)], axis=1)
for rep in range(int(os.environ.get("REPS", "3"))):
    out = env.step_many(tbl, time_budget_ms=slice_ms, max_resets=12)
    st = env.batch.debug_stats().astype(np.int64)
    t0 = st[:, 14] & 0x7fffffff
    t1 = st[:, 11] & 0x7fffffff
    base = t0.min()
    start_ms = (t0 - base) / 1e5
    end_ms = ((t1 - base) % (1 << 31)) / 1e5
    xcc = st[:, 12] & 0xF
    hw = st[:, 13]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    gen = np.floor(start_ms / (0.5 * slice_ms) + 0.25).astype(int) // 2
    print("launch %d: kernel %.0f ms, %s; workgroups started in slice 0 / 1 / 2 / 3+: %s" %
          (rep, env.batch.last_kernel_ms, env.batch.last_variant()["name"], [int((gen == g).sum()) for g in range(3)] + [int((gen >= 3).sum())]))
    late = np.where(gen >= 2)[0]
    for e in late.tolist():
        key = (xcc[e], se[e], sh[e], cu[e])
        same = np.where((xcc == key[0]) & (se == key[1]) & (sh == key[2]) & (cu == key[3]))[0]
        print("   LATE workgroup %d on xcc %d se %d sh %d cu %d (wave-0 simd %d slot %d): that CU's workgroups (id, start, end ms, simd, slot): %s" %
              (e, key[0], key[1], key[2], key[3], (hw[e] >> 4) & 3, hw[e] & 15,
               [(int(j), round(float(start_ms[j]), 1), round(float(end_ms[j]), 1), int((hw[j] >> 4) & 3), int(hw[j] & 15)) for j in same]))
        # which CU of that XCC hosted only one workgroup during slice 1?
        m1 = (xcc == key[0]) & (gen == 1)
        cnt1 = {}
        for j in np.where(m1)[0]:
            k2 = (int(se[j]), int(sh[j]), int(cu[j])); cnt1.setdefault(k2, []).append(int(j))
        lone = {k2: v for k2, v in cnt1.items() if len(v) != 2}
        print("   CUs of xcc %d with other than two workgroups in slice 1: %s" % (key[0], lone))
        for k2 in lone:
            same2 = np.where((xcc == key[0]) & (se == k2[0]) & (sh == k2[1]) & (cu == k2[2]))[0]
            print("      cu %s: %s" % (k2, [(int(j), round(float(start_ms[j]), 2), round(float(end_ms[j]), 2), int((hw[j] >> 4) & 3), int(hw[j] & 15)) for j in same2]))
    for x in sorted(set(xcc.tolist())) if os.environ.get("VERBOSE") else []:
        m = xcc == x
        cus = {(int(a), int(b), int(c)) for a, b, c in zip(se[m], sh[m], cu[m])}
        g0 = m & (gen == 0)
        per_cu = {}
        for a, b, c in zip(se[g0], sh[g0], cu[g0]):
            per_cu[(int(a), int(b), int(c))] = per_cu.get((int(a), int(b), int(c)), 0) + 1
        hist = np.bincount(list(per_cu.values()), minlength=4)[:6].tolist() if per_cu else []
        print("   xcc %d: %4d workgroups, by slice %s, %2d CUs seen; slice 0: CUs hosting 1 / 2 / 3 workgroups: %s; first..last start %.1f..%.1f ms, last end %.1f" %
              (x, int(m.sum()), [int((m & (gen == g)).sum()) for g in range(4)], len(cus), hist[1:4], start_ms[m].min(), start_ms[m].max(), end_ms[m].max()))
