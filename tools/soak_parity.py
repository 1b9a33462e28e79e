#!/usr/bin/env python3
"""Soak test (GPU box; uses the CPU oracle as the checker, so it is a TEST tool, not product code): many envs of the
bench workload in fp64 through several consecutive env steps, every env compared bit for bit with the oracle replaying
the same step from the same pre-step state.

  python tools/soak_parity.py [--envs 512] [--steps 4] [--n-side 25] [--init tier1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from gym_cloth_amd.envs import ClothVecEnv, decode_actions  # noqa: E402
from oracle import pyoracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=512)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--n-side", type=int, default=25)
    ap.add_argument("--init", default="tier1")
    ap.add_argument("--seed0", type=int, default=1000)
    args = ap.parse_args()
    pyoracle.build()
    E = args.envs
    thickness = 0.02 if args.n_side <= 25 else 0.0095
    cfg = bench.bench_cfg(args.n_side, thickness, args.init)
    env = ClothVecEnv(cfg, n_envs=E, precision="f64", consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(args.seed0 + e)
    rests = None
    if args.init == "tier2":        # tier 2 gives every env its own rest lengths (cloth.pyx:417): same draws as reset() makes
        rests = []
        for e in range(E):
            rng = np.random.RandomState(args.seed0 + e)
            side = rng.rand() > 0.5
            rests.append(env.batch.init_grid(2, side, rng.rand(env.P))[1])
    env.reset()
    c, ev = cfg["cloth"], cfg["env"]
    ocfg = {"n_side": args.n_side, "width": c["width"], "height": c["height"], "density": c["density"], "ks": c["ks"],
            "damping": c["damping"], "thickness": c["thickness"], "plane_friction": c["plane_friction"],
            "tear_thresh": c["tear_thresh"], "frames_per_sec": cfg["frames_per_sec"],
            "simulation_steps": cfg["simulation_steps"], "gravity": -9.8, "minimum_z": 0.0, "grip_radius": ev["grip_radius"]}
    threads = min(len(os.sched_getaffinity(0)), pyoracle.lib().oracle_max_threads())
    bad_total = 0
    for t in range(args.steps):
        acts = np.stack([np.random.RandomState(args.seed0 + 1000 + e).uniform(-1, 1, size=(args.steps, 4))[t] for e in range(E)])
        pos0, prev0, pin0 = env.batch.get_state()
        tear0 = np.array(env.batch.tear).copy()
        env.step(acts)
        pos1, prev1, pin1 = env.batch.get_state()
        ex = env.last_executed.copy()
        d = decode_actions(acts, [-1.] * 4, [1.] * 4, True, True, ev["reduce_factor"], ev["iters_up"], ev["iters_up_rest"],
                           ev["iters_pull_max"], ev["iters_grip_rest"], ev["iters_rest"])
        idx = [e for e in range(E) if ex[e] > 0]
        cloths, sched, delta = [], np.zeros((len(idx), 5), dtype=np.int32), np.zeros((len(idx), 3))
        for k, e in enumerate(idx):
            oc = pyoracle.OracleCloth(ocfg)
            oc.set_state(pos0[e], prev0[e], pin0[e], None if rests is None else rests[e])
            oc.have_tear = bool(tear0[e])
            ng = oc.grab_top(float(d["x"][e]), float(d["y"][e]))
            sched[k] = d["bounds"][e] if ng > 0 else 0
            delta[k] = (0.0025, d["x_dir_r"][e], d["y_dir_r"][e])
            cloths.append(oc)
        t0 = time.perf_counter()
        exo = pyoracle.batch_run_schedule(cloths, sched, delta, True, threads) if cloths else np.zeros(0)
        dt = time.perf_counter() - t0
        bad = []
        for k, e in enumerate(idx):
            op, oq, opin = cloths[k].get_state()
            if exo[k] != ex[e] or not (np.array_equal(pos1[e], op) and np.array_equal(prev1[e], oq)):
                bad.append((e, int(ex[e]), int(exo[k]), float(np.abs(pos1[e] - op).max())))
        st = env.batch.debug_stats()
        print("step %d: %d envs active, %d substeps, oracle %.1f s on %d threads | strain sweeps %d | mismatches %d %s" %
              (t, len(idx), int(ex.sum()), dt, threads, int(st[idx, 0].sum()), len(bad), bad[:3]),
              flush=True)
        bad_total += len(bad)
    print("SOAK", "OK" if bad_total == 0 else "FAILED (%d)" % bad_total)
    env.close()
    sys.exit(0 if bad_total == 0 else 1)


if __name__ == "__main__":
    main()
