#!/usr/bin/env python3
"""bench.py -- cloth-substeps/s of the HIP stepper on BASELINE.json's batched workload.

Workload (BASELINE.json configs[2], SURVEY.md 8d "C3"; weak-scaled to configs[3]'s 512 envs per GPU):
  E = 512 cloths of 25x25 per GPU, tier-1 start (flat grid + the scripted reset pulls drawn per env as in
  cloth_env.py:851-877 from RandomState(1000+e)), then one random pick-and-place action per env per step,
  a ~ U(-1,1)^4 in clip space from RandomState(2000+e), delta actions, EPISODES AS IN THE REFERENCE'S LOOP
  (examples/analytic.py:872-882): an env whose episode ends (out of bounds, tear, coverage > 0.92, max_actions = 10) is
  reset and goes on, so every env executes an action in every step.
A "step" = ClothEnv.step for every env of the batch = decode + Gripper.grab_top + the substep loop
(~1430 + iters_pull Cloth.update() per env, cloth_env.py:472-515) + metrics + terminal test.

Two execution modes (both reported; `value` is the fused one unless --mode step):
  fused  ClothVecEnv.step_many: one kernel launch (clothhip_run_actions) is a TIME SLICE in which every env executes its
         own sequence of actions and episode resets back to back (the resets' Cloth.update() calls are executed by the same
         kernel, inside the timed region, and are counted). Envs never wait for each other: the launch ends when the
         slice is used up, not when the env with the most work has finished a fixed number of actions, so after K
         "steps" the envs have executed K actions on average, not each exactly K (config.env_steps_executed says how
         many). The result of an action does not depend on which launch executes it.
  step   ClothVecEnv.step: one launch sequence per step (grab, schedule kernel, metrics), the clock is STOPPED around the
         host-driven episode resets between steps (SURVEY 8d excludes reset from the timed region).
value = SURVEY.md 8d's metric: Cloth.update()-equivalents executed by ACTIONS (grab, substep loop, release, metrics), all envs and
ranks, / the part of the timed wall time the envs spent in actions. 8d excludes create / reset from the timed region; in the fused
mode the episode resets run inside the same launches (an env resets and goes on while the others step), so the kernel accounts
every env's launch time to {actions, reset pulls, reset settling, other} with the 100 MHz s_memrealtime clock and the wall time is
split by those shares (config.action_time_frac; config.timed_region_s is the whole wall time). The blended figure -- ALL executed
update() calls, reset pulls and settling included, / the whole wall time -- is config.blended_substeps_per_s (rounds 1-4 printed it
as `value`). State is resident in HBM; the timed region contains no state upload.

Multi-GPU: one process per GPU (launched by the driver with torch.distributed.run, or by this script itself with
--gpus N when no launcher environment is present); env blocks are sharded, rank 0's action table is broadcast and the
per-env results are all-gathered with RCCL bound directly through ctypes (gym_cloth_amd/rccl.py) -- no torch.
Cloths never interact, so there is no other collective.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def tcp_port():
    """Port of the TCP transport (tests, --allow-tcp-fallback, --dry-run): CLOTH_BENCH_TCP_PORT if the launcher probed one, else MASTER_PORT + 17."""
    return int(os.environ.get("CLOTH_BENCH_TCP_PORT") or int(os.environ.get("MASTER_PORT", "29500")) + 17)


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
    bounded sample of the SAME workload: one bench action of the first n envs, started from the very cloth states the
    GPU path held at the start of the timed region (downloaded before it)."""
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
    k1 = int(np.argmax(ex))                                    # single-core figure: the busiest of those cloths again, three
    singles = []                                               # times over (a host core's rate varies from run to run): the median
    for _ in range(3):
        oc = pyoracle.OracleCloth(ocfg)
        oc.set_state(pos[k1], prev[k1], pin[k1])
        oc.grab_top(float(d["x"][k1]), float(d["y"][k1]))
        t1 = time.perf_counter()
        ex1 = oc.run_schedule(sched[k1], 0.0025, float(d["x_dir_r"][k1]), float(d["y_dir_r"][k1]), True)
        dt1 = time.perf_counter() - t1
        if dt1 > 0 and ex1 > 0:
            singles.append(float(ex1 / dt1))
    return {"value": float(ex.sum() / dt), "unit": "cloth-substeps/s", "cores": int(threads), "kind": "port",
            "sample": "one bench action of the first %d envs from the GPU run's own start states (%d substeps), OpenMP one cloth per thread"
                      % (n, int(ex.sum())),
            "single_core_value": float(np.median(singles)) if singles else None,
            "single_core_samples": singles}


TRAFFIC_FILES = ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json")       # the newest committed PMC record wins


def load_traffic(mode, E, n_side, precision, init, substeps_per_launch, b_alg=None):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r04_traffic.json, produced by
    tools/collect_profiles.sh: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, FETCH_SIZE doubled per the gfx950
    correction of MI355X_MICROARCH.md). The PMC run's launches are time slices like this run's, but not necessarily of the same
    length: the record's bytes PER SUBSTEP are scaled by THIS run's substeps per launch. None when no record matches this
    configuration (mode, envs, grid, precision, init)."""
    for name in TRAFFIC_FILES:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as fh:
                for r in json.load(fh)["records"]:
                    if (r["mode"], r["envs"], r["n_side"], r["precision"], r.get("init", "tier1")) == (mode, E, n_side, precision, init):
                        per_sub = r.get("hbm_bytes_per_substep")
                        if per_sub is None and r.get("substeps_per_launch"):
                            per_sub = r["hbm_bytes_per_launch"] / r["substeps_per_launch"]
                        if per_sub is None and r.get("algorithmic_bytes_per_launch") and b_alg:
                            per_sub = r["hbm_bytes_per_launch"] / (r["algorithmic_bytes_per_launch"] / b_alg)
                        if per_sub is None:
                            return r["hbm_bytes_per_launch"], "profiles/" + name     # (old records: bytes of THAT run's launches)
                        return per_sub * substeps_per_launch, "profiles/" + name
        except (OSError, ValueError, KeyError):
            pass
    return None, None


def run_workload(n_side, E, precision, init, mode, steps, warmup, fuse_max, rank, world, local_rank, thickness=None,
                 want_cpu=False, step_ms=170.0, slots=0, max_resets=0, allow_tcp_fallback=False, relaxed=False):
    """One bench configuration on this rank's GPU; returns the result record (rank 0) or None.
    relaxed=True: the relaxed-order companion kernel (self-collision in Jacobi order, strain limit in coloured order: NOT the
    reference's trajectories, no parity claim) -- what the exact order costs, as a measured figure. Never `value`."""
    return _run_workload(n_side, E, precision, init, mode, steps, warmup, fuse_max, rank, world, local_rank, thickness, want_cpu,
                         step_ms, slots, max_resets, allow_tcp_fallback, relaxed)


def _run_workload(n_side, E, precision, init, mode, steps, warmup, fuse_max, rank, world, local_rank, thickness, want_cpu,
                  step_ms, slots, max_resets, allow_tcp_fallback, relaxed):
    from gym_cloth_amd.dist import LocalTransport, RcclTransport, SocketTransport, StepExchange
    from gym_cloth_amd.envs import ClothVecEnv
    thickness = thickness if thickness is not None else (0.02 if n_side <= 25 else 0.0095)
    cfg = bench_cfg(n_side, thickness, init)
    env = ClothVecEnv(cfg, n_envs=E, device=local_rank, precision=precision, consume_domrand_draws=False)
    if relaxed:
        env.batch.set_relaxed_order(True)                    # THIS handle's episode launches run the companion kernel (ABI 7: per handle, no env var)
    transport_name = "none (1 GPU)"
    if world > 1 and os.environ.get("CLOTH_BENCH_FORCE_TCP") == "1":
        # test hook (tests/test_gpu_dist.py): two ranks on ONE GPU cannot form an RCCL communicator; the rank logic around the exchange --
        # sharding, per-rank streams, sums / max over ranks, the value arithmetic -- is the same over the TCP transport. Never a bench result.
        transport = SocketTransport(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), tcp_port())
        transport_name = "TCP sockets (forced by CLOTH_BENCH_FORCE_TCP: test only)"
    elif world > 1:
        try:
            transport, transport_name = RcclTransport(rank, world, env.batch), "RCCL (ctypes binding, handle stream, device-resident tables)"
        except Exception as exc:
            # a multi-GPU record that did not run over RCCL must not pass for one: the TCP transport is opt-in, and then ALL ranks
            # must fail the same way (a rank that did join the communicator would wait in it forever otherwise)
            if not allow_tcp_fallback:
                print("bench: rank %d: RCCL transport failed (%s: %s); pass --allow-tcp-fallback to run over TCP sockets instead"
                      % (rank, type(exc).__name__, exc), file=sys.stderr)
                raise
            print("bench: rank %d: RCCL transport failed (%s: %s); using the TCP transport" % (rank, type(exc).__name__, exc),
                  file=sys.stderr)
            transport = SocketTransport(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"),
                                        tcp_port())
            transport_name = "TCP sockets (RCCL init failed: %s)" % type(exc).__name__
    else:
        transport = LocalTransport()
    xch = StepExchange(E, transport)
    g0 = rank * E                                            # first global env index of this rank
    for e in range(E):                                       # reset draws from RandomState(1000+e)
        env.np_randoms[e] = np.random.RandomState(1000 + g0 + e)
    env.reset()
    P = env.P
    fuse = 1
    if mode == "fused":
        fuse = max(1, min(fuse_max, steps))              # nominal steps per launch: the timed region is steps // fuse launches
        if not env.batch.fused_supported:                    # grid too large for the in-kernel metrics: report the step mode
            if rank == 0:
                print("bench: fused mode unavailable for %dx%d; using step mode" % (n_side, n_side), file=sys.stderr)
            mode, fuse = "step", 1
    slots0 = slots
    total = warmup + steps
    acts_all = None
    if rank == 0:                                            # actions for ALL envs of the job, RandomState(2000+e)
        acts_all = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(total, 4))
                             for e in range(world * E)], axis=1)          # [total, world*E, 4]

    def fence():
        env.batch.sync(False)
        xch.barrier()                                        # RCCL all-reduce on the handle's stream + stream sync
        env.batch.sync(False)

    stat = {"sub": 0, "act_sub": 0, "kms": 0.0, "launches": 0, "ran": 0, "grabbed": 0, "slots": 0, "resets": 0, "out_of_slots": 0,
            "op_ticks": np.zeros(4), "op_subs": np.zeros(4)}
    cpu_acts = None
    cpu_states = None
    t_timed = 0.0
    slice_ms = fuse * step_ms
    calib = None
    if mode == "step":
        auto_reset_host = init in ("tier1", "tier2", "tier3")
        for t in range(total):
            a = xch.broadcast_actions(acts_all[t] if rank == 0 else None)
            if t == warmup and want_cpu and rank == 0:
                cpu_states = env.batch.get_state(0, min(E, 512))
            fence()
            t0 = time.perf_counter()
            obs, rew, done, info = env.step(a)
            xch.gather_results(rew, done, info["actual_coverage"], env.last_executed)
            fence()
            dt = time.perf_counter() - t0
            if t >= warmup:
                t_timed += dt
                s = int(env.last_executed.sum())
                stat["sub"] += s; stat["act_sub"] += s; stat["kms"] += env.batch.last_kernel_ms; stat["launches"] += 1
                stat["ran"] += E; stat["grabbed"] += int((info["n_grabbed"] > 0).sum()); stat["slots"] += E
            if auto_reset_host and done.any():               # the reference's episode loop; clock stopped
                stat["resets"] += int(done.sum()) if t >= warmup else 0
                env.reset(mask=done)
    else:
        # every env consumes ITS OWN action stream (RandomState(2000+e)) at its own pace: per launch rank 0 builds the table
        # of each env's next `slots` actions from the per-env counters, broadcasts it, and gets the consumed counts back.
        # The launches are time slices of fuse x (duration of one env step). That duration is MEASURED here, outside the timed
        # region: a short calibration launch, then the warm-up launches (each refines it); the timed launches all get the same slice,
        # so that the timed region covers `--steps` env steps per env whatever kernel is being measured.
        n_warm, n_timed = (warmup + fuse - 1) // fuse, max(1, steps // fuse)
        est_ms = float(step_ms)                              # the caller's hint, replaced by measurements below
        cal_steps = 2.0
        mk_slots = lambda ms: slots0 if slots0 > 0 else max(4 * fuse, int(ms / 40.0), 8)   # a missed grab costs no time: be generous
        max_slots = mk_slots(16.0 * fuse * est_ms)
        n_stream = total + max_slots * (n_warm + n_timed + 2) * 4
        if rank == 0:
            streams = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(n_stream, 4)) for e in range(world * E)])
            cnt = np.zeros(world * E, dtype=np.int64)
        device_xch = world > 1 and hasattr(transport, "broadcast_device")
        op_ticks = np.zeros(4)
        op_subs = np.zeros(4)

        def launch(ms, n_slots):
            tbl = None
            if rank == 0:
                idx = (cnt[None, :] + np.arange(n_slots)[:, None]) % n_stream
                tbl = streams[np.arange(world * E)[None, :], idx]                     # [slots, world*E, 4]
                tbl = np.ascontiguousarray(tbl.reshape(n_slots, world, E, 4).transpose(1, 0, 2, 3))   # rank-major blocks
            # RCCL: the table is broadcast in place in device memory and the launch reads this rank's block there
            return xch.broadcast_action_blocks(tbl, n_slots, env.batch if device_xch else None)

        def run(blk, d_blk, ms, n_slots):
            out = env.step_many(blk, n_actions=n_slots, actions_device_ptr=d_blk, auto_reset=True, time_budget_ms=ms,
                                max_resets=max_resets if max_resets > 0 else min(n_slots, 250))
            res = xch.gather_summary(env.batch)              # [world*E, 4]: slots consumed, episode over, coverage, action substeps
            n_ran = out["ran"].sum(axis=0)
            assert np.array_equal(res[g0:g0 + E, 0].astype(np.int64), n_ran)            # the device's summary == the records
            if rank == 0:
                cnt[:] += res[:, 0].astype(np.int64)
            return out, res, n_ran

        def refine(ms, res):
            # every rank holds the gathered summary, so every rank derives the same estimate
            per_env = float(res[:, 0].mean())
            return ms / per_env if per_env >= 0.5 else None

        calib = {"hint_ms": est_ms, "launches_ms": [], "estimates_ms": []}
        ms0 = cal_steps * est_ms                             # calibration launch (about two env steps per env)
        blk, d_blk = launch(ms0, mk_slots(ms0))
        out, res, n_ran = run(blk, d_blk, ms0, mk_slots(ms0))
        e1 = refine(ms0, res)
        if e1 is not None:
            est_ms = min(max(e1, est_ms / 16.0), est_ms * 16.0)
        calib["launches_ms"].append(ms0); calib["estimates_ms"].append(est_ms)
        for w in range(n_warm + n_timed):
            if w <= n_warm:
                slice_ms = fuse * est_ms                     # (fixed from the first timed launch on)
                slots = min(mk_slots(slice_ms), max_slots)
            blk, d_blk = launch(slice_ms, slots)
            if w == n_warm:
                if want_cpu and rank == 0:
                    cpu_states = env.batch.get_state(0, min(E, 512))
                    cpu_acts = blk[0][:min(E, 512)].copy() if blk is not None else None
                fence()
                t0 = time.perf_counter()
            out, res, n_ran = run(blk, d_blk, slice_ms, slots)
            if w < n_warm:
                e2 = refine(slice_ms, res)
                if e2 is not None:
                    est_ms = min(max(e2, est_ms / 2.0), est_ms * 2.0)
                calib["launches_ms"].append(slice_ms); calib["estimates_ms"].append(est_ms)
            if w >= n_warm:
                a_sub = int(out["executed"].sum())
                r_sub = int(out["reset_substeps"].sum()) + int(out.get("tail_reset_substeps", np.zeros(1)).sum())
                stat["sub"] += a_sub + r_sub; stat["act_sub"] += a_sub
                stat["kms"] += env.batch.last_kernel_ms; stat["launches"] += 1
                stat["ran"] += int(n_ran.sum()); stat["grabbed"] += int((out["n_grabbed"] > 0).sum()); stat["slots"] += slots * E
                stat["resets"] += int((out["reset_before"] > 0).sum()) + int((out.get("tail_reset_substeps", np.zeros(1)) > 0).sum())
                stat["out_of_slots"] += int((n_ran == slots).sum())
                op_ticks += out["op_ticks"].sum(axis=0).astype(np.float64)
                op_subs += out["op_substeps"].sum(axis=0).astype(np.float64)
        fence()
        t_timed = time.perf_counter() - t0
        stat["op_ticks"], stat["op_subs"] = op_ticks, op_subs
    variant = env.batch.last_variant()
    rccl_nranks = transport.comm.nranks if hasattr(transport, "comm") else None
    dt = xch.max_over_ranks(t_timed)
    n_sub_all = xch.sum_over_ranks(stat["sub"])
    # the actions' update() calls and the actions' ticks are counted at the SAME boundary -- the kernel's per-launch class
    # counters (EpState::subs / ticks, class 0) -- not substeps by completed records against ticks by launch
    if mode == "fused":
        stat["act_sub"] = int(round(float(stat["op_subs"][0])))
        stat["sub"] = int(round(float(np.sum(stat["op_subs"]))))
        n_sub_all = xch.sum_over_ranks(stat["sub"])
    n_act_all = xch.sum_over_ranks(stat["act_sub"])
    n_env_steps = xch.sum_over_ranks(stat["ran"])
    # the share of the envs' launch time spent in actions (all ranks): the kernel's own per-env accounting. Step mode runs its
    # resets outside the clock (and keeps no such accounting): everything timed there is an action
    tk_act_all = xch.sum_over_ranks(float(stat["op_ticks"][0]))
    tk_all = xch.sum_over_ranks(float(np.sum(stat["op_ticks"])))
    act_frac = tk_act_all / tk_all if (mode == "fused" and tk_all > 0) else 1.0
    rec = None
    if rank == 0:
        b_alg32 = 49 * P                                      # algorithmic bytes per cloth-substep (fp32 state)
        b_alg = b_alg32 if precision == "f32" else 97 * P     # the same accounting at the f64 instantiation's state width
        # the dominant kernel's rate on the metric's definition: action substeps of this rank / the kernel time its envs spent in actions
        tk_r, kms_act = stat["op_ticks"], 0.0
        if stat["kms"] > 0:
            kms_act = stat["kms"] * (float(tk_r[0]) / float(np.sum(tk_r)) if (mode == "fused" and np.sum(tk_r) > 0) else 1.0)
        ach = (stat["act_sub"] * b_alg / 1e9) / (kms_act / 1e3) if kms_act > 0 else 0.0
        ach_blended = (stat["sub"] * b_alg / 1e9) / (stat["kms"] / 1e3) if stat["kms"] > 0 else 0.0
        traffic, traffic_src = load_traffic(mode, E, n_side, precision, init, stat["sub"] / max(stat["launches"], 1), b_alg)
        # SURVEY 8d's metric proper (reset excluded): action substeps / time spent in actions. The kernel accounts every env's
        # launch time to {actions, reset pulls, reset settling, rest}; with n_conc cloths stepping concurrently on the GPU the
        # rate of an actions-only workload is n_conc * sum(action substeps) / sum(env-seconds in actions).
        tk, sb = stat["op_ticks"], stat["op_subs"]
        # cloths stepping concurrently: what the library reports for the kernel that ran (resident cloths per CU x CUs), at most E
        n_conc = min(E, max(1, variant["cloths_per_cu"]) * max(1, variant["n_cus"]))
        act_only = n_conc * sb[0] / (tk[0] / 1e8) if tk[0] > 0 else None
        reset_only = n_conc * (sb[1] + sb[2]) / ((tk[1] + tk[2]) / 1e8) if (tk[1] + tk[2]) > 0 else None
        steps_eq = max(n_env_steps / (world * E), 1e-9)
        rec = {
            # action substeps / the wall time spent in actions; ms_per_step on the same footing (one env step's action)
            "value": n_act_all / (dt * act_frac), "ms_per_step": dt * act_frac / steps_eq * 1e3, "dtype": precision,
            "value_definition": ("action_substeps / (timed_region_s * action_time_frac): Cloth.update() calls of ClothEnv.step actions over the "
                                 "actions' in-kernel share of the wall time (SURVEY 8d: resets excluded); all substeps / wall = "
                                 "config.blended_substeps_per_s") if mode == "fused" else
                                "action_substeps / timed_region_s (step mode: resets run outside the clock)",
            "config": {"workload": "%d batched %dx%d cloths per GPU, %s start, random pick-and-place actions, episodes "
                                   "reset as in the reference's loop (BASELINE configs[2]; configs[3] = 8 x this)"
                                   % (E, n_side, n_side, init.replace("tier", "tier-")),
                       "mode": mode, "launches": stat["launches"],
                       "mode_note": ("fused: %d launches, each a %.0f ms time slice of back-to-back actions and episode resets per env "
                                     "(value: the actions' substeps over the actions' share of the time; resets excluded, SURVEY 8d)"
                                     % (stat["launches"], slice_ms))
                                    if mode == "fused" else "step: one launch sequence per step, clock stopped around host-driven resets",
                       "envs_per_gpu": E, "n_side": n_side, "init": init, "exact_order": not relaxed,
                       "parity": "none (relaxed order: Jacobi self-collision, coloured strain limit -- not the reference's trajectories)" if relaxed
                                 else "exact order (fp64 bit-exact, fp32 to the stated tolerances: tests/)",
                       "transport": transport_name,
                       "rccl_nranks": rccl_nranks,                                      # ncclCommCount of the communicator that ran (None: no RCCL)
                       "variant": variant["name"], "resident_cloths": n_conc,
                       "variant_short": variant["name"].split(":")[0].replace("k_run_schedule", ""),          # the kernel the last launch ran (clothhip_last_variant)
                       "slice_calibration": calib,                                      # fused: how the time slices were sized, in this run
                       "env_steps_executed": n_env_steps, "steps_equivalent": n_env_steps / (world * E),
                       "env_steps_per_s": n_env_steps / dt,
                       # the timed region: whole wall time (max over ranks), the share of it the envs spent in actions (in-kernel
                       # accounting, all ranks), and what rounds 1-4 printed as `value`: ALL update() calls / the whole wall time
                       "timed_region_s": dt, "action_time_frac": act_frac,
                       "wall_ms_per_env_step_incl_resets": dt / steps_eq * 1e3,
                       "blended_substeps_per_s": n_sub_all / dt,
                       "substeps_per_env_step": n_sub_all / max(n_env_steps, 1),
                       "action_substeps_per_env_step": n_act_all / max(n_env_steps, 1),
                       "action_substeps_per_s": n_act_all / dt,                         # action substeps / WHOLE time: a lower bound
                       # SURVEY 8d's definition (reset excluded), this rank: action substeps / time the envs spent in actions
                       "action_only_substeps_per_s": act_only, "reset_only_substeps_per_s": reset_only,
                       "reset_substep_frac": 1.0 - stat["act_sub"] / max(stat["sub"], 1),
                       "time_frac_by_class": dict(zip(("actions", "reset_pulls", "reset_settling", "other"),
                                                      (tk / max(tk.sum(), 1.0)).tolist())),
                       "slice_ms": slice_ms if mode == "fused" else None,
                       "grabbed_env_frac": stat["grabbed"] / max(stat["ran"], 1),       # env-steps whose pick point hit the cloth (else 0 substeps)
                       "envs_out_of_slots": stat["out_of_slots"],                       # fused: envs that used all their action slots of a launch
                       # fraction of (env, launch) pairs in which the env was busy for the whole launch (fused: it did not run out
                       # of action slots; step: it executed its action slot)
                       "active_env_frac": 1.0 - stat["out_of_slots"] / max(E * stat["launches"], 1) if mode == "fused" else 1.0,
                       "episode_resets_in_timed_region": stat["resets"]},
            # achieved = algorithmic bytes of the ACTION substeps / the kernel time spent in actions (HIP events on the handle's stream x
            # the in-kernel share), i.e. the same numerator as `value`; *_blended: every substep of the launch / the whole kernel time
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS,
                         "traffic": traffic,
                         "traffic_source": ("%s (separate rocprofv3 --pmc passes, scaled per substep; not measured in this run)" % traffic_src)
                                           if traffic is not None else None,
                         "kernel": "k_run_schedule", "kernel_ms_avg": stat["kms"] / max(stat["launches"], 1),
                         "kernel_ms_in_actions_avg": kms_act / max(stat["launches"], 1),
                         "launches": stat["launches"], "alg_bytes_per_substep": b_alg,
                         # a time-sliced launch over more cloths than are resident is issued as one kernel dispatch per generation
                         # (clothhip_api.hip launch_run): kernel_ms_avg spans them all, rocprofv3 lists them one by one
                         "dispatches_per_launch": variant["dispatches"],        # as issued by the library (clothhip_last_dispatches)
                         "frac_on_fp32_bytes": (stat["act_sub"] * b_alg32 / 1e9) / (kms_act / 1e3) / HBM_PEAK_GBS if kms_act > 0 else 0.0,
                         "achieved_blended": ach_blended, "frac_blended": ach_blended / HBM_PEAK_GBS,
                         "substeps_per_launch": stat["sub"] / max(stat["launches"], 1),
                         "action_substeps_per_launch": stat["act_sub"] / max(stat["launches"], 1)},
        }
        if want_cpu and cpu_states is not None:
            rec["cpu_baseline"] = cpu_baseline(cfg, (cpu_acts if cpu_acts is not None else acts_all[warmup])[:len(cpu_states[0])],
                                               cpu_states)
    if rank == 0 and os.environ.get("CLOTH_BENCH_HOST_PROFILE") and getattr(env, "host_prof", None):
        print("bench: host time in step_many [s]: %s" % {k: round(v, 3) for k, v in env.host_prof.items()}, file=sys.stderr)
    xch.barrier()
    xch.t.close()
    env.close()
    return rec


STEP_MS_HINT = 80.0    # one env step of the headline workload (512 cloths, 25x25, fp32), order of magnitude only: every run
                       # measures its own step time before the timed region (run_workload's calibration) and sizes its slices from that


def render_bench(E, local_rank, size=224, reps=3):
    """Companion record for the headless rasteriser: RGB (+ depth) images per second of E crumpled 25x25 cloths
    at size x size, wall time of ClothBatch.render incl. the download of the images (what an image-observation env pays)."""
    from gym_cloth_amd.envs import ClothVecEnv
    env = ClothVecEnv(bench_cfg(25, 0.02), n_envs=E, device=local_rank, precision="f32", consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000 + e)
    env.reset()                                               # the reset pulls leave every cloth folded somewhere
    rec = {}
    for label, kw in (("rgb", dict(want_rgb=True, want_depth=False)), ("rgbd", dict(want_rgb=True, want_depth=True))):
        env.batch.render(width=size, height=size, **kw)      # warm-up (allocations)
        t0 = time.perf_counter()
        for _ in range(reps):
            env.batch.render(width=size, height=size, **kw)
        dt = (time.perf_counter() - t0) / reps
        rec[label + "_images_per_s"] = E / dt
        rec[label + "_ms_per_batch"] = dt * 1e3
    env.close()
    rec.update({"value": rec["rgb_images_per_s"], "unit": "images/s", "dtype": "f32",
                "config": {"workload": "%d cloths of 25x25 (tier-1 post-reset states), %dx%d RGB images, k_render + download" % (E, size, size)}})
    return rec


def demo_bench(E, local_rank, episodes=4000):
    """Companion record for the demonstration writer: the reference's own use of the path, `examples/analytic.py
    oracle --tier=1` (analytic.py:825-910) -- whole episodes with the oracle-corner policy, here evaluated in the kernel, until
    `episodes` episodes have finished over E envs; wall time incl. cutting the records into the reference's episode dicts.
    Episodes are taken in order of completion and the launch that completes the last one is run to its end, so the rate is a slight
    under-statement. (The reference notes ~250 minutes for 400 such episodes on one core, Blender observations included:
    BASELINE.md section 1; its physics alone is 70-95 s per 10-action episode.)"""
    from gym_cloth_amd.demos import collect_demos
    from gym_cloth_amd.envs import ClothVecEnv
    env = ClothVecEnv(bench_cfg(25, 0.02), n_envs=E, device=local_rank, precision="f32")
    env.seed(1337)
    t0 = time.perf_counter()
    eps = collect_demos(env, "oracle_corner", max_episodes=episodes, time_budget_ms=400.0)
    dt = time.perf_counter() - t0
    n_act = sum(len(ep["act"]) for ep in eps)
    n_sub = sum(int(ep["info"][-1]["num_sim_steps"]) for ep in eps)
    cov = float(np.mean([ep["info"][-1]["actual_coverage"] for ep in eps]))
    env.close()
    return {"value": len(eps) / dt, "unit": "episodes/s", "dtype": "f32", "episodes": len(eps), "wall_s_collect": dt,
            "actions_per_episode": n_act / max(len(eps), 1), "action_substeps_of_the_kept_episodes": n_sub,
            "mean_final_coverage": cov,
            "config": {"workload": "%d finished tier-1 episodes of the in-kernel oracle-corner policy over %d envs "
                                   "(resets, policy, steps, metrics on the device; records cut into episode dicts on the host)"
                                   % (len(eps), E)}}


def self_launch(args):
    """--gpus N without a launcher environment: start one child process per GPU (fresh processes: nothing in THIS process
    has touched the GPU, and no process is ever replaced by exec), relay rank 0's JSON line. All children are supervised: when one
    exits non-zero the others are terminated and its code is returned (a rank that failed to join the communicator must not leave the
    rest waiting in it)."""
    import secrets
    import socket
    import tempfile
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    rdzv_dir = tempfile.mkdtemp(prefix="clothhip_rdzv_")          # private directory (0700): nobody else can plant an id file
    rdzv = os.path.join(rdzv_dir, "rccl.id")
    nonce = secrets.token_hex(16)
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CLOTHHIP_RDZV_FILE=rdzv, CLOTHHIP_RDZV_NONCE=nonce, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    try:
        if os.path.exists(rdzv):
            os.remove(rdzv)
        os.rmdir(rdzv_dir)
    except OSError:
        pass
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode())
    return rc


def dry_run(args, rank, world):
    """--dry-run: everything of a multi-rank bench run but the GPU -- launcher environment, rendezvous file protocol (a stand-in
    128-byte id travels exactly as the RCCL unique id does), the per-launch exchange pattern over the TCP transport and rank 0's
    JSON line. The CPU test suite drives bench.py's own launcher through this (tests/test_dist_sockets.py)."""
    from gym_cloth_amd import rccl
    from gym_cloth_amd.dist import LocalTransport, SocketTransport, StepExchange
    E = args.envs
    if os.environ.get("CLOTH_BENCH_DRY_FAIL_RANK") == str(rank):      # test hook: this rank dies before joining
        return 7
    head = rccl._MAGIC + rccl._nonce(rccl.rendezvous_path(), world) + int(world).to_bytes(4, "little")
    path = rccl.rendezvous_path()
    if world > 1:
        if rank == 0:
            tmp = "%s.tmp%d" % (path, os.getpid())
            with open(tmp, "wb") as fh:
                fh.write(head + bytes(range(128)))
            os.replace(tmp, path)
        else:
            t0 = time.time()
            while True:
                try:
                    raw = open(path, "rb").read()
                    if raw[:len(head)] == head and raw[len(head):] == bytes(range(128)):
                        break
                except OSError:
                    pass
                if time.time() - t0 > 60:
                    print("bench: rank %d: no rendezvous file" % rank, file=sys.stderr)
                    return 3
                time.sleep(0.01)
    t = SocketTransport(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), tcp_port()) \
        if world > 1 else LocalTransport()
    xch = StepExchange(E, t)
    slots, total = 3, 0.0
    for w in range(args.steps):
        tbl = None
        if rank == 0:
            tbl = np.arange(world * slots * E * 4, dtype=np.float64).reshape(world, slots, E, 4) + w
        blk, d_blk = xch.broadcast_action_blocks(tbl, slots)
        assert d_blk is None and blk.shape == (slots, E, 4) and blk[0, 0, 0] == rank * slots * E * 4 + w
        summ = np.stack([np.full(E, slots, dtype=np.float64), np.zeros(E), blk[-1, :, 0], np.full(E, 100.0 * (rank + 1))], axis=1)

        class _B(object):                                     # what gather_summary reads off a ClothBatch
            def run_summary(self_):
                return summ
        res = xch.gather_summary(_B())
        assert res.shape == (world * E, 4)
        total += float(res[:, 3].sum())
    dt = xch.max_over_ranks(0.001 * (rank + 1))
    n_sub = xch.sum_over_ranks(100.0 * (rank + 1) * E * args.steps)
    xch.barrier()
    t.close()
    if world > 1 and rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    if rank == 0:
        print(json.dumps({"metric": "dry run (no GPU): launcher, rendezvous and exchange plumbing", "value": n_sub / dt,
                          "unit": "cloth-substeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "config": {"workload": "dry run", "transport": "TCP sockets (dry run)", "gathered_substeps": total}}))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=512, help="cloths per GPU")
    ap.add_argument("--n-side", type=int, default=25)
    ap.add_argument("--thickness", type=float, default=None)
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--init", default="tier1", choices=["tier1", "tier2", "tier3"],
                    help="start state of every env (reset draws come from RandomState(1000+e))")
    ap.add_argument("--mode", default="fused", choices=["fused", "step"])
    ap.add_argument("--fuse", type=int, default=10, help="fused mode: nominal steps per launch; the timed region is steps // fuse launches")
    ap.add_argument("--step-ms", type=float, default=None,
                    help="fused mode: a HINT for the duration of one env step. A launch is a time slice of fuse * (step time); the step "
                         "time is measured by a calibration launch and the warm-up launches of the run itself, outside the timed region, "
                         "so that the timed region covers --steps env steps whatever kernel is measured")
    ap.add_argument("--slots", type=int, default=0, help="fused mode: action slots per env and launch (0: 4 * fuse)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-tcp-fallback", action="store_true",
                    help="multi-GPU: run the exchange over TCP sockets if RCCL cannot be initialised (default: fail)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: exercise launcher, rendezvous and exchange plumbing only")
    ap.add_argument("--no-extra", action="store_true", help="skip the companion records (f64, step mode, tier-2, 50x50, 2048 cloths)")
    args = ap.parse_args()

    if args.step_ms is None:
        args.step_ms = STEP_MS_HINT * (args.n_side / 25.0) ** 2 * (1.8 if args.precision == "f64" else 1.0) * max(1.0, args.envs / 1024.0)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    from gym_cloth_amd.dist import env_from_launcher
    rank, local_rank, world = env_from_launcher()
    if args.gpus > 1 and world != args.gpus and rank == 0:
        print("bench: --gpus %d but the launcher started %d ranks; using %d" % (args.gpus, world, world), file=sys.stderr)
    if args.dry_run:
        sys.exit(dry_run(args, rank, world))

    head = run_workload(args.n_side, args.envs, args.precision, args.init, args.mode, args.steps, args.warmup, args.fuse,
                        rank, world, local_rank, thickness=args.thickness,
                        want_cpu=(world == 1 and not args.no_cpu_baseline), step_ms=args.step_ms, slots=args.slots,
                        allow_tcp_fallback=args.allow_tcp_fallback)
    extra = []
    if world == 1 and not args.no_extra:
        def companion(key, label, **kw):
            t0 = time.perf_counter()
            try:
                r = run_workload(**kw)
            except Exception as exc:                         # a companion must never cost the headline
                r = {"error": "%s: %s" % (type(exc).__name__, exc)}
            r["key"], r["label"], r["wall_s"] = key, label, time.perf_counter() - t0
            extra.append(r)
        k5 = dict(rank=0, world=1, local_rank=local_rank, slots=args.slots)
        other = "f64" if args.precision == "f32" else "f32"
        companion(other, "same workload, %s instantiation%s" % (other, " (bit-exact vs the reference)" if other == "f64" else ""),
                  n_side=args.n_side, E=args.envs, precision=other, init=args.init, mode=args.mode, steps=10, warmup=5,
                  fuse_max=10, step_ms=args.step_ms * (1.8 if other == "f64" else 0.6), **k5)
        companion("step_mode" if args.mode == "fused" else "fused_mode", "same workload, the other execution mode", n_side=args.n_side, E=args.envs, precision=args.precision,
                  init=args.init, mode="step" if args.mode == "fused" else "fused", steps=10, warmup=5, fuse_max=10,
                  step_ms=args.step_ms, **k5)
        if args.precision == "f32" and args.n_side == 25 and args.envs <= 512 and args.init != "tier2" and args.mode == "fused":
            companion("relaxed_order", "RELAXED ORDER, no parity (SURVEY 7-H4): same workload with self-collision in Jacobi order and the strain limit in "
                      "coloured order -- what the reference's exact order costs; never `value`",
                      n_side=args.n_side, E=args.envs, precision="f32", init=args.init, mode="fused", steps=10, warmup=5, fuse_max=10,
                      step_ms=0.5 * args.step_ms, relaxed=True, **k5)
        if args.init != "tier2":
            companion("tier2_512", "BASELINE configs[3] shape: tier-2 start and tier-2 episode resets (per-env rest tables), 512 cloths per GPU",
                      n_side=25, E=512, precision=args.precision, init="tier2", mode=args.mode, steps=10, warmup=5, fuse_max=10,
                      step_ms=args.step_ms, **k5)
        if args.envs < 2048 and args.n_side == 25:
            companion("e2048", "2048 cloths per GPU (LEAN stepper variant at the residency clothhip_create picks: config.variant)",
                      n_side=25, E=2048, precision=args.precision, init="tier1", mode="fused", steps=5, warmup=0, fuse_max=5,
                      step_ms=1.3 * args.step_ms, **k5)
            companion("e1536", "1536 cloths per GPU (LEAN stepper variant at the residency clothhip_create picks: config.variant)",
                      n_side=25, E=1536, precision=args.precision, init="tier1", mode="fused", steps=5, warmup=0, fuse_max=5,
                      step_ms=1.1 * args.step_ms, **k5)
        if args.n_side == 25:
            companion("configs4_50x50", "BASELINE configs[4]: 50x50 x 1024 cloths, thickness 0.0095", n_side=50, E=1024,
                      precision=args.precision, init="tier1", mode="fused", steps=5, warmup=0, fuse_max=5, step_ms=250.0,
                      want_cpu=not args.no_cpu_baseline, **k5)
        if args.n_side == 25:
            t0 = time.perf_counter()
            try:
                r = render_bench(args.envs, local_rank)
            except Exception as exc:
                r = {"error": "%s: %s" % (type(exc).__name__, exc)}
            r["key"], r["label"], r["wall_s"] = "render", "headless rasteriser (SURVEY 8f-f4): image observations of the whole batch", time.perf_counter() - t0
            extra.append(r)
            t0 = time.perf_counter()
            try:
                r = demo_bench(args.envs, local_rank)
            except Exception as exc:
                r = {"error": "%s: %s" % (type(exc).__name__, exc)}
            r["key"], r["label"], r["wall_s"] = "demos", "demonstration writer (SURVEY 8f-f2): 4000 oracle-corner episodes, policy in the kernel", time.perf_counter() - t0
            extra.append(r)
    if rank == 0:
        out = compact_line(args, world, head, extra)
        line = json.dumps(out, separators=(",", ":"))
        assert len(line) <= MAX_LINE_BYTES, len(line)      # the driver parses ONE short line; everything else is in the side file
        write_extra(head, extra, out)
        print(line)


MAX_LINE_BYTES = 4096
EXTRA_FILE = "bench_extra.json"


def _sig(v, n=6):
    """Floats of the printed line to n significant digits (the side file keeps full precision)."""
    if isinstance(v, float):
        return float("%.*g" % (n, v)) if v == v and abs(v) != float("inf") else None
    if isinstance(v, dict):
        return {k: _sig(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, n) for x in v]
    return v


def _summary(r):
    """One companion record on the printed line: its value, roofline fraction and blended rate only."""
    if r is None:
        return None
    if "error" in r:
        return {"error": r["error"][:120]}
    s = {"value": r["value"]}
    if "roofline" in r:
        s["frac"] = r["roofline"]["frac_on_fp32_bytes"]          # on SURVEY 8d's 49 P bytes per substep whatever the instantiation's state width
        s["blended"] = r["config"]["blended_substeps_per_s"]
        s["variant"] = r["config"]["variant_short"]
    return s


def compact_line(args, world, head, extra):
    """The ONE JSON line the driver parses (<= MAX_LINE_BYTES): the contract keys, `config` (workload + how the timed region went),
    `roofline`, `cpu_baseline`, and one-line summaries of the companion measurements. The full records go to bench_extra.json."""
    c, r = head["config"], head["roofline"]
    keep_c = ("workload", "mode", "envs_per_gpu", "n_side", "init", "exact_order", "variant", "transport", "rccl_nranks", "launches", "slice_ms",
              "timed_region_s", "action_time_frac", "steps_equivalent", "env_steps_executed", "env_steps_per_s", "blended_substeps_per_s",
              "substeps_per_env_step", "action_substeps_per_env_step", "grabbed_env_frac", "episode_resets_in_timed_region")
    keep_r = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "kernel_ms_avg", "kernel_ms_in_actions_avg",
              "launches", "dispatches_per_launch", "alg_bytes_per_substep", "action_substeps_per_launch", "substeps_per_launch",
              "achieved_blended", "frac_blended")
    out = {
        "metric": "cloth substeps/sec (25x25 grid, batched envs)" if args.n_side == 25 else
                  "cloth substeps/sec (%dx%d grid, batched envs)" % (args.n_side, args.n_side),
        "value": head["value"], "unit": "cloth-substeps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        # machine-visible: WHAT `value` divides by what. Rounds 1-4 printed config.blended_substeps_per_s as `value`.
        "value_definition": head["value_definition"],
        "config": {k: c[k] for k in keep_c if k in c},
        "roofline": {k: r[k] for k in keep_r if k in r},
    }
    if "cpu_baseline" in head:
        b = head["cpu_baseline"]
        out["cpu_baseline"] = {k: b[k] for k in ("value", "unit", "cores", "kind", "sample", "single_core_value") if k in b}
    elif world > 1:
        out["cpu_baseline"] = "skipped (world > 1: the CPU port is timed by the 1-GPU run only)"
    by_key = {r_.get("key"): r_ for r_ in extra}
    for key in ("f64", "f32", "relaxed_order", "step_mode", "fused_mode", "tier2_512", "e1536", "e2048", "configs4_50x50"):
        if key in by_key:
            out[key] = _summary(by_key[key])
    if "relaxed_order" in out and "error" not in out["relaxed_order"]:
        out["relaxed_order"].update({"exact_order": False, "parity": "none"})
    for key in ("render", "demos"):
        if key in by_key:
            out[key] = {"value": by_key[key].get("value"), "unit": by_key[key].get("unit")} if "error" not in by_key[key] else {"error": by_key[key]["error"][:120]}
    if extra:
        out["extra_file"] = EXTRA_FILE
    return _sig(out)


def write_extra(head, extra, line):
    """Full-precision records of the headline and of every companion measurement, beside the script (and a note on stderr)."""
    try:
        with open(os.path.join(ROOT, EXTRA_FILE), "w") as fh:
            json.dump({"line": line, "headline": head, "extra": extra}, fh, indent=1, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
        print("bench: full records (headline + %d companions) written to %s" % (len(extra), EXTRA_FILE), file=sys.stderr)
    except OSError as exc:
        print("bench: could not write %s (%s)" % (EXTRA_FILE, exc), file=sys.stderr)


if __name__ == "__main__":
    main()
