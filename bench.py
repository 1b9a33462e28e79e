#!/usr/bin/env python3
"""bench.py -- cloth-substeps/s of the HIP stepper on BASELINE.json's batched workload.

Workload (BASELINE.json configs[2], SURVEY.md 8d "C3"; weak-scaled to configs[3]'s 512 envs per GPU):
  E = 512 cloths of 25x25 per GPU, tier-1 start (flat grid + two scripted reset pulls drawn per env as in
  cloth_env.py:851-877 from RandomState(1000+e)), then one random pick-and-place action per env per step,
  a ~ U(-1,1)^4 in clip space from RandomState(2000+e), delta actions.
A "step" = ClothVecEnv.step over the whole batch = Gripper.grab_top + the fused schedule kernel
(~1430 + iters_pull substeps per env, cloth_env.py:472-515) + metrics.  State is resident in HBM; the timed
region contains no state upload.  value = executed Cloth.update()-equivalents (all envs, all ranks) / wall time.

`--init tier2` starts every env from the reference's tier-2 reset instead (BASELINE configs[3] names it; the default
keeps ONE workload at every GPU count so that the per-N values are comparable).

Multi-GPU (one process per GPU, torch.distributed 'nccl' = RCCL over xGMI): env blocks are sharded, rank 0's
action table is broadcast every step and per-env results are all-gathered; there is no other collective
because cloths never interact.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def bench_cfg(n_side, thickness, tier="tier1"):
    return {
        "cloth": {"damping": 2.0, "density": 200.0, "ks": 10000.0, "width": 1, "height": 1,
                  "num_width_points": n_side, "num_height_points": n_side, "thickness": thickness,
                  "pin_cond": "y=0", "color_pts": "None", "plane_friction": 1.0, "tear_thresh": 2.0},
        "frames_per_sec": 30, "simulation_steps": 30,
        "env": {"max_actions": 10, "max_z_threshold": 5, "iters_up": 50, "iters_up_rest": 80,
                "iters_pull_max": 400, "iters_grip_rest": 300, "iters_rest": 1000, "updates_per_move": 1,
                "reduce_factor": 0.002, "grip_radius": 0.003, "reward_type": "coverage-delta",
                "force_grab": False, "clip_act_space": True, "delta_actions": True, "obs_type": "1d",
                "oracle_reveal": "False", "use_depth": "False", "use_dom_rand": "False", "use_rgbd": "False"},
        "init": {"type": tier, "debug_matplotlib": False, "render_opengl": False},
        "log": {"level": "info", "file": "logs/bench.log"}, "seed": 1000}


def cpu_baseline(cfg, acts0, states, budget_s=15.0):
    """The CPU oracle (oracle/, exact-order fp64 C port of the reference) timed on this box's host cores on a
    bounded sample of the SAME workload: the first timed bench step of the first n envs, started from the very
    cloth states the GPU path started that step from (downloaded before the timed region)."""
    from oracle import pyoracle
    pyoracle.build()
    cores = len(os.sched_getaffinity(0))
    threads = min(cores, pyoracle.lib().oracle_max_threads())
    c = cfg["cloth"]
    ocfg = {"n_side": c["num_width_points"], "width": c["width"], "height": c["height"], "density": c["density"],
            "ks": c["ks"], "damping": c["damping"], "thickness": c["thickness"],
            "plane_friction": c["plane_friction"], "tear_thresh": c["tear_thresh"],
            "frames_per_sec": cfg["frames_per_sec"], "simulation_steps": cfg["simulation_steps"],
            "gravity": -9.8, "minimum_z": 0.0, "grip_radius": cfg["env"]["grip_radius"]}
    pos, prev, pin = states
    # ~70-150 us per 25x25 substep per core -> size the sample for ~budget_s of wall time
    per_sub = 100e-6 * (c["num_width_points"] ** 2) / 625.0
    n = int(max(threads, min(len(acts0), budget_s * threads / (1200 * per_sub))))
    n = min(max(threads, (n // threads) * threads), len(acts0), len(pos))
    from gym_cloth_amd.envs import decode_actions
    e = cfg["env"]
    d = decode_actions(acts0[:n], [-1.] * 4, [1.] * 4, True, True, e["reduce_factor"], e["iters_up"], e["iters_up_rest"],
                       e["iters_pull_max"], e["iters_grip_rest"], e["iters_rest"])
    cloths, sched, delta = [], np.zeros((n, 5), dtype=np.int32), np.zeros((n, 3))
    for k in range(n):
        oc = pyoracle.OracleCloth(ocfg)
        oc.set_state(pos[k], prev[k], pin[k])
        ng = oc.grab_top(float(d["x"][k]), float(d["y"][k]))
        sched[k] = d["bounds"][k] if ng > 0 else 0
        delta[k] = (0.0025, d["x_dir_r"][k], d["y_dir_r"][k])
        cloths.append(oc)
    t0 = time.perf_counter()
    ex = pyoracle.batch_run_schedule(cloths, sched, delta, True, threads)
    dt = time.perf_counter() - t0
    k1 = int(np.argmax(ex))                                    # single-core figure: the busiest of those cloths again
    oc = pyoracle.OracleCloth(ocfg)
    oc.set_state(pos[k1], prev[k1], pin[k1])
    oc.grab_top(float(d["x"][k1]), float(d["y"][k1]))
    t1 = time.perf_counter()
    ex1 = oc.run_schedule(sched[k1], 0.0025, float(d["x_dir_r"][k1]), float(d["y_dir_r"][k1]), True)
    dt1 = time.perf_counter() - t1
    return {"value": float(ex.sum() / dt), "unit": "cloth-substeps/s", "cores": int(threads), "kind": "port",
            "sample": "first timed step of the first %d envs (same start states and actions as the GPU run, %d substeps "
                      "in total), OpenMP one cloth per thread" % (n, int(ex.sum())),
            "single_core_value": float(ex1 / dt1) if dt1 > 0 and ex1 > 0 else None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--envs", type=int, default=512, help="cloths per GPU")
    ap.add_argument("--n-side", type=int, default=25)
    ap.add_argument("--thickness", type=float, default=None)
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--init", default="tier1", choices=["tier1", "tier2", "tier3"],
                    help="start state of every env (BASELINE configs[3] names tier2; reset draws come from RandomState(1000+e))")
    ap.add_argument("--gather-obs", action="store_true", help="all-gather the '1d' observations every step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:     # launched by torch.distributed.run: one rank per GPU
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from gym_cloth_amd.envs import ClothVecEnv

    E = args.envs
    thickness = args.thickness if args.thickness is not None else (0.02 if args.n_side <= 25 else 0.0095)
    cfg = bench_cfg(args.n_side, thickness, args.init)
    env = ClothVecEnv(cfg, n_envs=E, device=local_rank, precision=args.precision, consume_domrand_draws=False)
    g0 = rank * E                                            # first global env index of this rank
    for e in range(E):                                       # reset draws from RandomState(1000+e)
        env.np_randoms[e] = np.random.RandomState(1000 + g0 + e)
    env.reset()
    total_steps = args.warmup + args.steps
    P = env.P
    if rank == 0:                                            # actions for ALL envs of the job, RandomState(2000+e)
        acts_all = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(total_steps, 4))
                             for e in range(world * E)], axis=1)          # [steps, world*E, 4]
    else:
        acts_all = None

    from gym_cloth_amd.dist import StepExchange
    if dist is not None:
        import torch
        dev = torch.device("cuda", local_rank)
    else:
        dev = None
    xch = StepExchange(E, obs_dim=(3 * P if args.gather_obs else 0), device=dev) if dist is not None else None

    def one_step(t):
        # RCCL broadcast of the action table (rank 0 -> all), each rank keeps its env block
        a = xch.broadcast_actions(acts_all[t] if rank == 0 else None) if xch else acts_all[t]
        obs, rew, done, info = env.step(a)
        kms = env.batch.last_kernel_ms
        if xch:
            xch.gather_results(rew, done, info["actual_coverage"], env.last_executed)   # RCCL all-gather
            if args.gather_obs:
                env.batch.write_obs_f32_device(xch.obs_loc.data_ptr())
                env.batch.sync(False)
                xch.gather_obs()
        return int(env.last_executed.sum()), kms

    def fence():
        env.batch.sync(False)
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for t in range(args.warmup):
        one_step(t)
    fence()
    cpu_states = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # start states of the first timed step (untimed copy)
        ncpu = min(E, 512)
        cpu_states = env.batch.get_state(0, ncpu)
        fence()
    t0 = time.perf_counter()
    n_sub, k_ms, k_sub = 0, 0.0, 0
    for t in range(args.warmup, total_steps):
        s, kms = one_step(t)
        n_sub += s
        k_ms += kms
        k_sub += s
    fence()
    dt = time.perf_counter() - t0
    if xch:
        dt = xch.max_over_ranks(dt)
        n_sub_all = xch.sum_over_ranks(n_sub)
    else:
        n_sub_all = float(n_sub)

    if rank == 0:
        b_alg = 49 * P                                        # algorithmic bytes per cloth-substep (fp32)
        ach = (k_sub * b_alg / 1e9) / (k_ms / 1e3) if k_ms > 0 else 0.0
        out = {
            "metric": "cloth substeps/sec (25x25 grid, batched envs)" if args.n_side == 25 else
                      "cloth substeps/sec (%dx%d grid, batched envs)" % (args.n_side, args.n_side),
            "value": n_sub_all / dt, "unit": "cloth-substeps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "%d batched %dx%d cloths per GPU, %s start, random pick-and-place actions "
                                   "(BASELINE configs[2]; configs[3] = 8 x this)" % (E, args.n_side, args.n_side,
                                                                                    args.init.replace("tier", "tier-")),
                       "envs_per_gpu": E, "n_side": args.n_side, "exact_order": True,
                       "env_steps_per_s": world * E * args.steps / dt,
                       "substeps_per_env_step": n_sub_all / (world * E * args.steps)},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_run_schedule", "kernel_ms_avg": k_ms / max(args.steps, 1),
                         "alg_bytes_per_substep": b_alg},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, acts_all[args.warmup], cpu_states)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    env.close()


if __name__ == "__main__":
    main()
