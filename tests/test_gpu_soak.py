"""Reduced soak, in pytest: whole env steps of every start tier and of the 50x50 grid in fp64 against the CPU oracle
(bit-exact, env by env), and the fp32 mode's long-horizon agreement stated on OUTCOMES over many seeds (SURVEY 7-H2 iii).
The full-size version of the first part is tools/soak_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ocfg(cfg):
    c = cfg["cloth"]
    return {"n_side": c["num_width_points"], "width": c["width"], "height": c["height"], "density": c["density"],
            "ks": c["ks"], "damping": c["damping"], "thickness": c["thickness"], "plane_friction": c["plane_friction"],
            "tear_thresh": c["tear_thresh"], "frames_per_sec": cfg["frames_per_sec"],
            "simulation_steps": cfg["simulation_steps"], "gravity": -9.8, "minimum_z": 0.0,
            "grip_radius": cfg["env"]["grip_radius"]}


@pytest.mark.parametrize("tier,n_side,E,steps", [("tier2", 25, 64, 2), ("tier3", 25, 16, 2), ("tier1", 50, 6, 1)])
def test_env_steps_match_oracle_f64(tier, n_side, E, steps, oracle_lib):
    """bench.py's workload shape (reset drawn from RandomState(1000+e), random actions from RandomState(2000+e)) for the
    tiers and grid the headline run does not cover: after every env step every env is bit-identical to the oracle
    advanced from the same pre-step state (tier 2 carries per-env rest lengths)."""
    import bench
    from gym_cloth_amd.envs import ClothVecEnv, decode_actions
    cfg = bench.bench_cfg(n_side, 0.02 if n_side == 25 else 0.0095, tier)
    env = ClothVecEnv(cfg, n_envs=E, precision="f64", consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000 + e)
    env.reset()
    rest = env.batch.get_rest()
    ev = cfg["env"]
    acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(steps, 4)) for e in range(E)], axis=1)
    busy = 0
    for t in range(steps):
        pos0, prev0, pin0 = env.batch.get_state()
        tear0 = env.batch.tear
        env.step(acts[t])
        pos1, prev1, pin1 = env.batch.get_state()
        d = decode_actions(acts[t], [-1.] * 4, [1.] * 4, True, True, ev["reduce_factor"], ev["iters_up"], ev["iters_up_rest"],
                           ev["iters_pull_max"], ev["iters_grip_rest"], ev["iters_rest"])
        for e in range(E):
            oc = oracle_lib.OracleCloth(_ocfg(cfg))
            oc.set_state(pos0[e], prev0[e], pin0[e], rest[e])
            oc.have_tear = bool(tear0[e])
            ng = oc.grab_top(float(d["x"][e]), float(d["y"][e]))
            n = oc.run_schedule(d["bounds"][e], 0.0025, float(d["x_dir_r"][e]), float(d["y_dir_r"][e]), True) if ng > 0 else 0
            assert n == env.last_executed[e], (tier, t, e, n, env.last_executed[e])
            op, oq, _ = oc.get_state()
            assert np.array_equal(pos1[e], op) and np.array_equal(prev1[e], oq), (tier, t, e, float(np.abs(pos1[e] - op).max()))
            busy += int(n > 0)
    assert busy >= E * steps // 3
    env.close()


def test_f32_outcome_distribution_vs_f64():
    """fp32 is the throughput mode; over a whole action rounding differences grow chaotically, so parity is
    stated on outcomes, over 64 independently seeded envs stepped twice in both precisions from identical start states:
    the discrete outcomes (nothing grabbed / done / tear / out of bounds) agree for at least 90 % of the env-steps, the
    substep counts for at least 90 %, and the coverage differs by less than 0.02 in the median and 0.1 at the 90th
    percentile."""
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    E, steps = 64, 2
    cfg = bench.bench_cfg(25, 0.02)
    envs = {}
    for prec in ("f64", "f32"):
        v = ClothVecEnv(cfg, n_envs=E, precision=prec, consume_domrand_draws=False)
        for e in range(E):
            v.np_randoms[e] = np.random.RandomState(1000 + e)
        envs[prec] = v
    envs["f64"].reset()
    p, q, c = envs["f64"].batch.get_state()
    envs["f32"].reset()                                            # same RNG draws; then overwrite with the fp64 start state
    envs["f32"].batch.set_state(p, q, c)
    acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(steps, 4)) for e in range(E)], axis=1)
    agree, same_n, dcov = [], [], []
    for t in range(steps):
        res = {}
        for prec, v in envs.items():
            obs, rew, done, info = v.step(acts[t])
            res[prec] = (done.copy(), info["have_tear"].copy(), info["out_of_bounds"].copy(), v.last_executed.copy(),
                         info["actual_coverage"].copy())
        a, b = res["f64"], res["f32"]
        agree.append((a[0] == b[0]) & (a[1] == b[1]) & (a[2] == b[2]) & ((a[3] == 0) == (b[3] == 0)))
        same_n.append(a[3] == b[3])
        dcov.append(np.abs(a[4] - b[4]))
        envs["f32"].batch.set_state(*envs["f64"].batch.get_state())   # re-synchronise at the action boundary, tear flag included
        envs["f32"].batch.tear = envs["f64"].batch.tear
        envs["f32"].have_tear[:] = envs["f64"].have_tear
    agree, same_n, dcov = np.concatenate(agree), np.concatenate(same_n), np.concatenate(dcov)
    print("\nfp32 vs fp64 over %d env-steps: outcome agreement %.3f, same substep count %.3f, |dcoverage| median %.2e p90 %.2e max %.2e"
          % (len(agree), agree.mean(), same_n.mean(), np.median(dcov), np.percentile(dcov, 90), dcov.max()))
    assert agree.mean() >= 0.90 and same_n.mean() >= 0.90
    assert np.median(dcov) < 0.02 and np.percentile(dcov, 90) < 0.1
    for v in envs.values():
        v.close()


def test_f32_free_running_episode_outcomes_vs_f64():
    """Long horizon, free running (SURVEY 7-H2 iii): 128 independently seeded envs run ONE WHOLE EPISODE each (up to max_actions = 10
    random actions, never re-synchronised) in fp32 and in fp64 from the same seeds and action streams. Individual trajectories
    decorrelate after the first action or two -- what must agree is the DISTRIBUTION of outcomes: episode length, how episodes end
    (out of bounds / tear / coverage reached / action budget), final coverage. The first action, which both precisions start from
    (nearly) the same state, must also agree env by env. With CLOTH_OUTCOME_REPORT=<path> the measured figures are written there
    (profiles/r03_f32_outcomes.json is that file from the round's GPU run)."""
    import json
    import os
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    E, T = 128, 10
    cfg = bench.bench_cfg(25, 0.02)
    acts = np.ascontiguousarray(np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1))
    res = {}
    for prec in ("f64", "f32"):
        v = ClothVecEnv(cfg, n_envs=E, precision=prec, consume_domrand_draws=False)
        for e in range(E):
            v.np_randoms[e] = np.random.RandomState(1000 + e)
        v.reset()
        out = v.step_many(acts, auto_reset=False)
        ran = out["ran"]
        n = ran.sum(axis=0)                                        # episode length in actions
        last = np.maximum(n - 1, 0)
        idx = (last, np.arange(E))
        cov = out["actual_coverage"][idx]
        reason = np.where(out["have_tear"][idx], 1, np.where(out["out_of_bounds"][idx], 2, np.where(cov > 0.92, 3, 4)))
        res[prec] = dict(n=n, cov=cov, reason=reason, first_done=out["done"][0].copy(), first_exec=out["executed"][0].copy(),
                         first_cov=out["actual_coverage"][0].copy(), first_grab=(out["n_grabbed"][0] > 0).copy(),
                         rew_sum=(out["rew"] * ran).sum(axis=0))
        v.close()
    a, b = res["f64"], res["f32"]
    names = {1: "tear", 2: "out_of_bounds", 3: "coverage_reached", 4: "action_budget"}
    freq = lambda r: {names[k]: float((r["reason"] == k).mean()) for k in names}
    first_agree = float(((a["first_done"] == b["first_done"]) & (a["first_grab"] == b["first_grab"])).mean())
    first_same_n = float((a["first_exec"] == b["first_exec"]).mean())
    first_dcov = np.abs(a["first_cov"] - b["first_cov"])
    rep = {"envs": E, "max_actions": T,
           "first_action": {"outcome_agreement": first_agree, "same_substep_count": first_same_n,
                            "abs_dcoverage_median": float(np.median(first_dcov)), "abs_dcoverage_p90": float(np.percentile(first_dcov, 90))},
           "episode": {"same_length_frac": float((a["n"] == b["n"]).mean()), "same_end_reason_frac": float((a["reason"] == b["reason"]).mean()),
                       "mean_length": {"f64": float(a["n"].mean()), "f32": float(b["n"].mean())},
                       "end_reason_freq": {"f64": freq(a), "f32": freq(b)},
                       "final_coverage_mean": {"f64": float(a["cov"].mean()), "f32": float(b["cov"].mean())},
                       "final_coverage_quartiles": {"f64": np.percentile(a["cov"], [25, 50, 75]).tolist(),
                                                    "f32": np.percentile(b["cov"], [25, 50, 75]).tolist()},
                       "return_mean": {"f64": float(a["rew_sum"].mean()), "f32": float(b["rew_sum"].mean())},
                       "abs_dfinal_coverage_median_per_env": float(np.median(np.abs(a["cov"] - b["cov"])))}}
    print("\n" + json.dumps(rep, indent=1))
    path = os.environ.get("CLOTH_OUTCOME_REPORT")
    if path:
        with open(path, "w") as fh:
            json.dump(rep, fh, indent=1)
    assert first_agree >= 0.90 and first_same_n >= 0.85 and np.median(first_dcov) < 0.02
    assert abs(a["n"].mean() - b["n"].mean()) < 0.5                                   # episode length distribution
    fa, fb = freq(a), freq(b)
    assert max(abs(fa[k] - fb[k]) for k in fa) < 0.10                                 # how episodes end
    assert abs(a["cov"].mean() - b["cov"].mean()) < 0.03                              # where they end
    qa, qb = np.percentile(a["cov"], [25, 50, 75]), np.percentile(b["cov"], [25, 50, 75])
    assert np.abs(qa - qb).max() < 0.06
