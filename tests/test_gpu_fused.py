"""GPU tests of the whole-episode launch (clothhip_run_actions / ClothVecEnv.step_many): action decoding, grab_top,
the substep loop, metrics, terminal test and episode resets inside ONE kernel launch must give, env for env and bit for
bit in fp64, what the per-step path gives (which the other GPU tests pin to the reference goldens and the CPU oracle),
and the in-kernel reset must reproduce the reference's reset goldens directly."""
import numpy as np
import pytest

from test_gpu_env import base_cfg

pytestmark = pytest.mark.gpu


def _bench_env(E, prec, tier="tier1", n_side=25):
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    cfg = bench.bench_cfg(n_side, 0.02 if n_side <= 25 else 0.0095, tier)
    env = ClothVecEnv(cfg, n_envs=E, precision=prec, consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000 + e)
    return env


def _rng_equal(a, b):
    sa, sb = a.get_state(), b.get_state()
    return sa[0] == sb[0] and np.array_equal(sa[1], sb[1]) and sa[2:] == sb[2:]


@pytest.mark.parametrize("tier,T,E,device_rng", [("tier1", 4, 24, True), ("tier1", 4, 24, False), ("tier3", 3, 12, True),
                                                  ("tier3", 3, 12, False), ("tier2", 3, 12, True)])
def test_step_many_equals_sequential_steps_f64(tier, T, E, device_rng):
    """T actions per env with auto-reset: one fused launch == T calls of step(auto_reset=True). Every reward, done flag,
    info value, counter, the final particle state and the state of every env's RNG must be identical. Uniformly random
    actions end many episodes early (out of bounds, tears), so the in-kernel reset path is exercised, including
    episodes that end twice within the launch."""
    acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1)
    a = _bench_env(E, "f64", tier); a.reset()
    b = _bench_env(E, "f64", tier); b.reset()
    assert np.array_equal(a.batch.get_state()[0], b.batch.get_state()[0])
    seq = []
    for t in range(T):
        obs, rew, done, info = a.step(acts[t], auto_reset=True)
        seq.append((rew, done, info, info["executed"], info["n_grabbed"]))
    out = b.step_many(acts, reset_tail=True, device_rng=device_rng)
    n_idle = 0
    for t in range(T):
        rew, done, info, ex, ng = seq[t]
        ran = out["ran"][t]
        n_idle += int((~ran).sum())
        # an env whose reset ran tier 1's conditional third pull idles after its NEXT episode (its later scripts are void)
        assert np.array_equal(rew[ran], out["rew"][t][ran]), (t, rew, out["rew"][t])
        assert np.array_equal(done[ran], out["done"][t][ran])
        assert np.array_equal(ex[ran], out["executed"][t][ran]) and np.array_equal(ng[ran], out["n_grabbed"][t][ran])
        for k in ("num_steps", "num_sim_steps", "actual_coverage", "start_coverage", "variance_inv",
                  "start_variance_inv", "have_tear", "out_of_bounds"):
            assert np.array_equal(np.asarray(info[k])[ran], out[k][t][ran]), (t, k)
    assert n_idle <= (0 if device_rng else E * T // 8), n_idle      # device-drawn resets never leave an env without a reset
    assert out["reset_before"].sum() > 0, "the workload must exercise the in-kernel reset"
    pa, qa, ca = a.batch.get_state()
    pb, qb, cb = b.batch.get_state()
    assert np.array_equal(pa, pb) and np.array_equal(qa, qb) and np.array_equal(ca, cb)
    assert np.array_equal(a.batch.tear, b.batch.tear)
    assert np.array_equal(a.num_steps, b.num_steps) and np.array_equal(a.num_sim_steps, b.num_sim_steps)
    assert a.total_substeps == b.total_substeps
    assert all(_rng_equal(a.np_randoms[e], b.np_randoms[e]) for e in range(E))
    a.close(); b.close()


@pytest.mark.parametrize("fixture,tier,seed", [("g_env_tier1_1337.npz", "tier1", 1337),
                                               ("g_env_tier3_1337.npz", "tier3", 1337),
                                               ("g_env_tier3_1339.npz", "tier3", 1339)])
def test_in_kernel_reset_matches_reference_f64(fixture, tier, seed, oracle_lib):
    """The reset executed INSIDE the kernel (flat grid, scripted pulls with the pick point read from the particle state,
    _prevent_oob, tier-1's coverage-conditional third pull, tier-3's float iters_up and 800 settling updates) against
    the reference's reset capture: same clip-space reset actions, same update() counts, same post-reset observation and
    start coverage; then the first episode action as in the golden."""
    from gym_cloth_amd.envs import ClothVecEnv
    g = oracle_lib.load_golden(fixture)
    v = ClothVecEnv(base_cfg(tier, seed), n_envs=1, precision="f64")
    v.seed([seed])
    v._ep_done[:] = True                                   # "episode over": the launch starts with the reset
    nreset = int(g["n_reset_calls"])
    first = g["act"][nreset] if len(g["act"]) > nreset else np.array([0.1, 0.1, 0.2, 0.2])
    rec, rst, obs_t, robs = v.batch.run_actions(v._episode_params(), 1, np.zeros(1, dtype=np.int32),
                                                np.ones(1, dtype=np.uint8), actions=first[None, None, :],
                                                scripts=v._prepare_scripts(1), want_obs=True)
    q = rst[0, 0]
    assert q["consumed"] == 1 and q["pulls_run"] == nreset and rec[0, 0]["reset_before"] == 1
    for k in range(nreset):
        assert np.array_equal(q["action"][k], g["act"][k]), (k, q["action"][k], g["act"][k])
        assert q["executed"][k] == int(g["act_n_updates"][k])
    assert np.array_equal(robs[0, 0], g["reset_obs"].astype(np.float32))
    assert abs(q["start_coverage"] - float(g["start_coverage"])) <= 1e-12
    assert abs(q["start_variance_inv"] - float(g["start_variance_inv"])) <= 1e-9 * float(g["start_variance_inv"])
    if len(g["act"]) > nreset:                             # tier-1 fixture: the oracle action of the episode follows
        assert rec[0, 0]["executed"] == int(g["act_n_updates"][nreset])
        assert np.array_equal(obs_t[0, 0], g["act_pos1"][nreset].reshape(-1).astype(np.float32))
        assert abs(rec[0, 0]["coverage"] - g["info"][0]["actual_coverage"]) <= 1e-12
        assert bool(rec[0, 0]["done"]) == bool(g["done"][0])
    v.close()


def test_device_oracle_policy_episode_f64(oracle_lib):
    """The oracle-corner policy evaluated in the kernel picks the action the reference's examples/analytic.py picked
    (seed 1337, tier 1) and the episode ends with the reference's reward."""
    from gym_cloth_amd.envs import ClothVecEnv
    g = oracle_lib.load_golden("g_env_tier1_1337.npz")
    v = ClothVecEnv(base_cfg("tier1", 1337), n_envs=2, precision="f64")
    v.seed([1337, 1337])
    v.reset()
    out = v.step_many(policy="oracle_corner", n_actions=2, auto_reset=False)
    k = int(g["n_reset_calls"])
    for e in range(2):
        assert np.array_equal(out["actions"][0, e], g["act"][k])
        assert abs(out["rew"][0, e] - g["rew"][0]) <= 1e-12 and out["done"][0, e]
        assert not out["ran"][1, e]                        # episode over, no reset requested: the second slot idles
    v.close()


def test_step_many_f32_outcomes_and_obs():
    """fp32 instantiation of the fused launch: identical to the fp32 per-step path (same kernel arithmetic), and the
    per-slot observations equal the state after each step."""
    E, T = 8, 2
    acts = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1)
    a = _bench_env(E, "f32"); a.reset()
    b = _bench_env(E, "f32"); b.reset()
    obs_seq = []
    for t in range(T):
        obs, rew, done, info = a.step(acts[t], auto_reset=False)
        obs_seq.append((obs.copy(), rew, done))
    out = b.step_many(acts, auto_reset=False, want_obs=True)
    for t in range(T):
        ran = out["ran"][t]
        assert np.array_equal(obs_seq[t][1][ran], out["rew"][t][ran]) and np.array_equal(obs_seq[t][2][ran], out["done"][t][ran])
        if t == 0:
            assert ran.all() and np.array_equal(out["obs_t"][0], obs_seq[0][0].astype(np.float32))
    a.close(); b.close()


def test_fused_refuses_what_it_cannot_run():
    from gym_cloth_amd import _lib
    from gym_cloth_amd.envs import ClothVecEnv
    v = ClothVecEnv(base_cfg("tier2", 3), n_envs=2, precision="f32")
    v.seed(3); v.reset()                                   # tier 2: per-env rest tables
    out = v.step_many(np.zeros((1, 2, 4)), auto_reset=True, device_rng=False)    # allowed: resets simply stay on the host
    assert out["ran"].all()
    with pytest.raises(Exception):
        v.batch.run_actions(v._episode_params(), 1, np.zeros(2, dtype=np.int32), np.zeros(2, dtype=np.uint8),
                            actions=np.zeros((1, 2, 4)), scripts=np.zeros((2, 3), dtype=_lib.RESET_SCRIPT_DTYPE))
    v.close()


@pytest.mark.parametrize("tier", ["tier1", "tier2", "tier3"])
def test_time_sliced_launches_give_the_same_trajectories_f64(tier):
    """A launch with a time budget lets every env advance at its own pace; an env's unused slots are passed again in the
    next launch. However the action sequence of an env is cut into launches, its trajectory is the same: the concatenated
    per-env (reward, done, executed) sequences and the final states equal those of ONE launch over the whole sequence."""
    E, N = (12, 6) if tier == "tier1" else (6, 4)
    streams = np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(N, 4)) for e in range(E)])   # [E, N, 4]
    a = _bench_env(E, "f64", tier); a.reset()
    ref = a.step_many(np.ascontiguousarray(streams.transpose(1, 0, 2)), max_resets=8)
    b = _bench_env(E, "f64", tier); b.reset()
    cnt = np.zeros(E, dtype=np.int64)
    got = [[] for _ in range(E)]
    slots, launches = 4, 0
    while (cnt < N).any():
        idx = np.minimum(cnt[None, :] + np.arange(slots)[:, None], N - 1)
        tbl = streams[np.arange(E)[None, :], idx]
        # ~25 ms per slice: every launch cuts the running action or reset of every env somewhere in the middle
        out = b.step_many(tbl, max_resets=8, time_budget_ms=25.0)
        launches += 1
        for e in range(E):
            n_e = int(out["ran"][:, e].sum())
            assert out["ran"][:n_e, e].all() and not out["ran"][n_e:, e].any()        # executed slots form a prefix
            take = min(n_e, N - int(cnt[e]))
            got[e] += [(out["rew"][t, e], bool(out["done"][t, e]), int(out["executed"][t, e])) for t in range(take)]
            if take < n_e:                                                         # ran past its stream: stop comparing this env
                cnt[e] = N + 1000
            else:
                cnt[e] += n_e
        assert launches < 3000
    assert launches > 10, "the budget must actually cut the sequences"
    for e in range(E):
        want = [(ref["rew"][t, e], bool(ref["done"][t, e]), int(ref["executed"][t, e])) for t in range(N) if ref["ran"][t, e]]
        n = min(len(want), len(got[e]))
        assert n >= 2 and got[e][:n] == want[:n], (e, got[e][:n], want[:n])
    a.close(); b.close()


def test_in_flight_operations_are_dropped_by_outside_state_changes():
    """A time slice leaves operations in flight inside the handle; a host-side reset (or any other state upload) must void
    them: the next launch then starts from the uploaded state like a fresh env."""
    E = 4
    acts = np.stack([np.random.RandomState(2000 + e).uniform(-0.5, 0.5, size=(3, 4)) for e in range(E)], axis=1)
    a = _bench_env(E, "f64"); a.reset()
    a.step_many(acts, time_budget_ms=5.0)                  # cut in the middle of the first action
    a.reset()                                              # host-side reset: flat grid + scripted pulls, in-flight ops dropped
    b = _bench_env(E, "f64"); b.reset(); b.reset()         # same RNG consumption: two host resets
    oa = a.step_many(acts)
    ob = b.step_many(acts)
    assert np.array_equal(oa["rew"], ob["rew"]) and np.array_equal(oa["obs"], ob["obs"])
    a.close(); b.close()


def test_masked_host_reset_keeps_the_other_envs_in_flight_operations_f64():
    """A time slice parks the running operation of every env inside the handle. A host-side reset of SOME envs
    (ClothVecEnv.reset(mask)) voids the parked operations of those envs only: the others continue theirs in the next launch
    exactly as if nothing had happened in between (their rewards / executed counts equal an undisturbed run's), and the reset
    envs restart from their new episode like envs that were reset between two whole launches."""
    E, N = 6, 3
    acts = np.stack([np.random.RandomState(2000 + e).uniform(-0.6, 0.6, size=(N, 4)) for e in range(E)], axis=1)   # [N, E, 4]
    mask = np.array([True, False, True, False, False, True])

    def collect(out, got):
        for e in range(E):
            n_e = int(out["ran"][:, e].sum())
            got[e] += [(out["rew"][t, e], bool(out["done"][t, e]), int(out["executed"][t, e])) for t in range(n_e)]
        return out["ran"].sum(axis=0)

    # undisturbed reference for the envs that are NOT reset: one launch over the whole sequence
    ref = _bench_env(E, "f64"); ref.reset()
    r = ref.step_many(acts, auto_reset=False)
    # a: sliced launch (cuts every env's first action), masked host reset, then launches until every env is through
    a = _bench_env(E, "f64"); a.reset()
    got = [[] for _ in range(E)]
    cnt = collect(a.step_many(acts, auto_reset=False, time_budget_ms=8.0), got)
    assert (cnt == 0).all(), "the slice must end inside the first action"
    a.reset(mask=mask)
    # b: the reset envs' reference -- same RNG consumption (two host resets of those envs), nothing in flight
    b = _bench_env(E, "f64"); b.reset(); b.reset(mask=mask)
    rb = b.step_many(acts, auto_reset=False)
    for _ in range(400):
        idx = np.minimum(cnt[None, :] + np.arange(N)[:, None], N - 1)
        tbl = acts[idx, np.arange(E)[None, :]]
        live = (cnt < N) & ~a._ep_done                       # an env is through after N actions or at its episode's end
        if not live.any():
            break
        a._ep_done = a._ep_done | ~live                      # envs that are through idle
        cnt = cnt + collect(a.step_many(tbl, auto_reset=False, time_budget_ms=8.0), got)
    for e in range(E):
        src = rb if mask[e] else r
        want = [(src["rew"][t, e], bool(src["done"][t, e]), int(src["executed"][t, e])) for t in range(N) if src["ran"][t, e]]
        n = min(len(want), len(got[e]))
        assert n >= 1 and got[e][:n] == want[:n], (e, bool(mask[e]), got[e][:n], want[:n])
    ref.close(); a.close(); b.close()


@pytest.mark.parametrize("tier,seed", [("tier1", 1337), ("tier1", 21), ("tier3", 1339), ("tier2", 1337), ("tier2", 1338)])
def test_device_drawn_reset_equals_host_reset_f64(tier, seed, oracle_lib):
    """The reset the kernel draws from the env's numpy stream (MT19937 on the device, csrc/cloth_rng.hpp) equals
    ClothEnv.reset on the host: same post-reset particles, same start coverage, same init_side, and -- with the
    domain-randomisation draws of cloth_env.py:786-789 consumed on both sides -- the same RandomState afterwards."""
    from gym_cloth_amd.envs import ClothVecEnv
    h = ClothVecEnv(base_cfg(tier, seed), n_envs=2, precision="f64")
    h.seed([seed, seed + 1])
    h.reset()
    d = ClothVecEnv(base_cfg(tier, seed), n_envs=2, precision="f64")
    d.seed([seed, seed + 1])
    d._ep_done[:] = True
    # an action that grabs nothing (far corner of the action space, nothing there after a reset pull in most cases) keeps the
    # post-reset state observable; if it does grab, the comparison below uses the reset observation instead
    out = d.step_many(np.tile(np.array([0.999, 0.999, 0.0, 0.0]), (1, 2, 1)), want_obs=True, max_resets=2)
    assert (out["reset_before"][0] == 1).all()
    hp = h.batch.get_state()[0]
    assert np.array_equal(out["reset_obs"][:, 0], hp.reshape(2, -1).astype(np.float32))
    assert np.array_equal(d._start_coverage, h._start_coverage) and np.array_equal(d._start_variance_inv, h._start_variance_inv)
    assert np.array_equal(d.init_side, h.init_side)
    for e in range(2):
        assert _rng_equal(d.np_randoms[e], h.np_randoms[e]), e
    if seed in (1337, 1338) and tier in ("tier1", "tier2"):          # the reference's own capture of this reset
        g = oracle_lib.load_golden("g_env_%s_%d.npz" % (tier, seed))
        assert np.array_equal(out["reset_obs"][0, 0], g["reset_obs"].astype(np.float32))
    if tier == "tier2":                                               # the rest lengths the kernel rebuilt (cloth.pyx:417)
        assert np.array_equal(d.batch.get_rest(), h.batch.get_rest())
    h.close(); d.close()


def test_demo_writer_device_equals_host_loop_f64(tmp_path):
    """collect_demos with the oracle-corner policy evaluated in the kernel (policy, steps and episode resets in one
    launch) writes the same episodes as the reference-shaped host loop (policy object -> step -> reset), env by env."""
    import pickle
    from gym_cloth_amd.demos import collect_demos
    from gym_cloth_amd.envs import ClothVecEnv
    from gym_cloth_amd.policies import OracleCornerPolicy

    def make():
        v = ClothVecEnv(base_cfg("tier1", 1337), n_envs=3, precision="f64", consume_domrand_draws=False)
        v.seed([1337, 1338, 1339])
        return v
    a, b = make(), make()
    dev = collect_demos(a, "oracle_corner", max_episodes=6, slots_per_launch=6, path=str(tmp_path / "demos.pkl"))
    host = collect_demos(b, OracleCornerPolicy(b), max_episodes=6)
    assert pickle.load(open(str(tmp_path / "demos.pkl"), "rb"))[0]["act"] == dev[0]["act"]
    by_env = lambda eps: {e: [ep for ep in eps if ep["env"] == e] for e in range(3)}
    d, h = by_env(dev), by_env(host)
    compared = 0
    for e in range(3):
        for ed, eh in zip(d[e], h[e]):
            assert ed["act"] == eh["act"] and ed["rew"] == eh["rew"] and ed["done"] == eh["done"]
            assert len(ed["obs"]) == len(eh["obs"]) == len(ed["act"]) + 1
            for od, oh in zip(ed["obs"], eh["obs"]):
                assert np.array_equal(od, oh.astype(np.float32))
            assert ed["info"] == eh["info"]
            compared += 1
    assert compared >= 3
    a.close(); b.close()


def test_demo_writer_with_time_slices_equals_unsliced_f64():
    """collect_demos over time-sliced launches: a slice may end right after a completed reset, so that no action of that launch
    carries the reset mark; the writer must still open a new episode there. The episodes equal those of unsliced launches."""
    from gym_cloth_amd.demos import collect_demos
    from gym_cloth_amd.envs import ClothVecEnv

    def make():
        v = ClothVecEnv(base_cfg("tier1", 1337), n_envs=3, precision="f64", consume_domrand_draws=False)
        v.seed([1337, 1338, 1339])
        return v
    a, b = make(), make()
    whole = collect_demos(a, "oracle_corner", max_episodes=6, slots_per_launch=6)
    sliced = collect_demos(b, "oracle_corner", max_episodes=6, slots_per_launch=6, time_budget_ms=20.0)
    by_env = lambda eps: {e: [ep for ep in eps if ep["env"] == e] for e in range(3)}
    w, s_ = by_env(whole), by_env(sliced)
    compared = 0
    for e in range(3):
        for ew, es in zip(w[e], s_[e]):
            assert ew["act"] == es["act"] and ew["rew"] == es["rew"] and ew["done"] == es["done"], e
            assert len(es["obs"]) == len(es["act"]) + 1
            for ow, os_ in zip(ew["obs"], es["obs"]):
                assert np.array_equal(ow, os_)
            compared += 1
    assert compared >= 3
    a.close(); b.close()


@pytest.mark.parametrize("tier", ["tier1", "tier2"])
def test_highest_point_policy_on_the_device_equals_the_host_policy(tier):
    """examples/analytic.py:723-808 evaluated in the kernel (k + 1 rounds of a stable arg-max over the heights, target = the
    point's place on the flat cloth, tier 2's (orig_z, orig_y) / (1 - orig_z, orig_y) included) against the same policy run
    on the host through ClothVecEnv.step, fed the same picks: identical actions, rewards, dones and observations over
    whole episodes with their resets, fp64."""
    from gym_cloth_amd.demos import collect_demos
    from gym_cloth_amd.envs import ClothVecEnv
    from gym_cloth_amd.policies import HighestPointPolicy

    def make():
        v = ClothVecEnv(base_cfg(tier, 1337), n_envs=3, precision="f64", consume_domrand_draws=False)
        v.seed([1337, 1338, 1339])
        return v
    a, b = make(), make()
    dev = collect_demos(a, HighestPointPolicy(a, seed=11), max_episodes=5, slots_per_launch=5, on_device=True)
    host = collect_demos(b, HighestPointPolicy(b, seed=11), max_episodes=5)
    by_env = lambda eps: {e: [ep for ep in eps if ep["env"] == e] for e in range(3)}
    d, h = by_env(dev), by_env(host)
    compared = steps = 0
    for e in range(3):
        for ed, eh in zip(d[e], h[e]):
            assert ed["act"] == eh["act"] and ed["rew"] == eh["rew"] and ed["done"] == eh["done"], (tier, e)
            for od, oh in zip(ed["obs"], eh["obs"]):
                assert np.array_equal(od, oh.astype(np.float32))
            assert ed["info"] == eh["info"]
            compared += 1; steps += len(ed["act"])
    assert compared >= 3 and steps >= 6
    a.close(); b.close()


@pytest.mark.parametrize("variant", ["force_grab", "no_clip", "grid50_f32", "grid50_f64", "grid64_f32"])
def test_step_many_equals_sequential_other_configurations(variant):
    """The episode launch against sequential step() calls in configurations the other tests do not touch: force_grab (the
    in-kernel radius-growing loop, cloth_env.py:434-444), clip_act_space off (actions and reset pulls in world units, the
    out-of-bounds action penalty computed from the recorded action), and the 50x50 grid in fp32 (the 512-thread x 5-particle
    variant with float sort keys in the in-kernel metrics; same kernel arithmetic on both paths, so identical results), and in
    fp64 (BASELINE configs[4] in the reference's arithmetic: the in-kernel metrics keep their hull stack as indices there -- 71 KB of
    LDS scratch instead of 107 KB -- and must give the bits of the stand-alone metrics kernel the step path runs)."""
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    n_side, prec, E, T = {"grid50_f32": (50, "f32", 3, 2), "grid50_f64": (50, "f64", 3, 2), "grid64_f32": (64, "f32", 2, 2)}.get(variant, (25, "f64", 8, 3))
    cfg = bench.bench_cfg(n_side, 0.02 if n_side == 25 else (0.0095 if n_side == 50 else 0.007))     # (64x64: the largest grid, the 1024 x 4 variant)
    if variant == "force_grab":
        cfg["env"]["force_grab"] = True
    if variant == "no_clip":
        cfg["env"]["clip_act_space"] = False
    envs = []
    for _ in range(2):
        v = ClothVecEnv(cfg, n_envs=E, precision=prec, consume_domrand_draws=False)
        for e in range(E):
            v.np_randoms[e] = np.random.RandomState(1000 + e)
        v.reset()
        envs.append(v)
    a, b = envs
    assert b.batch.fused_supported
    lo, hi = (-1.2, 1.2) if variant != "no_clip" else (-0.3, 1.3)       # some actions outside the action space
    acts = np.stack([np.random.RandomState(2000 + e).uniform(lo, hi, size=(T, 4)) for e in range(E)], axis=1)
    seq = [a.step(acts[t], auto_reset=True) for t in range(T)]
    out = b.step_many(acts, reset_tail=True)
    for t in range(T):
        obs, rew, done, info = seq[t]
        assert out["ran"][t].all()
        assert np.array_equal(rew, out["rew"][t]) and np.array_equal(done, out["done"][t]), (variant, t)
        assert np.array_equal(info["executed"], out["executed"][t]) and np.array_equal(info["n_grabbed"], out["n_grabbed"][t])
    if variant == "force_grab":
        assert (out["n_grabbed"] > 0).all()
    assert np.array_equal(a.batch.get_state()[0], b.batch.get_state()[0])
    a.close(); b.close()
