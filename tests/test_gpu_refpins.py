"""GPU parity pins against captures of the REFERENCE's own callers (tests/golden/make_golden.py imports the reference and
records them; nothing here compares the HIP path with itself): whole random-action episodes of ClothEnv incl. the reset between
two episodes, replayed through the fused episode launch; examples/analytic.py's HighestPointPolicy; the non-delta 'coverage'
reward; the save_state -> start_state_path round trip of a tier-2 cloth. fp64: particles, observations, counters and flags bit
for bit; coverage and what derives from it (rewards) to 1e-12 -- the reference takes the hull area from Qhull (scipy), this package
from its own monotone chain, pinned to Qhull's within 2e-16 by the metrics golden (the physics never sees the value)."""
COV_TOL = 1e-12
import numpy as np
import pytest

from test_gpu_env import base_cfg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture,tier,seed", [("g_episodes_tier1_1337.npz", "tier1", 1337), ("g_episodes_tier3_1339.npz", "tier3", 1339)])
def test_fused_episode_launch_replays_the_references_episodes_f64(fixture, tier, seed, oracle_lib):
    """The reference's collection loop (analytic.py:866-882) with random actions: reset, step until done, reset, ... One fused
    launch (ClothVecEnv.step_many: decode, grab_top, substeps, metrics, terminal test AND the episode resets, drawn from the env's
    numpy RandomState on the device) over the same action sequence gives the reference's rewards, done flags, update() counts,
    info values, the first observation of every new episode and the final particles -- directly fused-versus-reference."""
    from gym_cloth_amd.envs import ClothVecEnv
    g = oracle_lib.load_golden(fixture)
    T = len(g["act"])
    v = ClothVecEnv(base_cfg(tier, seed), n_envs=1, precision="f64")
    v.seed([seed])
    obs0 = v.reset()
    assert np.array_equal(obs0[0], g["reset_obs"][0]) and abs(v._start_coverage[0] - g["reset_start_coverage"][0]) <= COV_TOL
    out = v.step_many(np.ascontiguousarray(g["act"][:, None, :]), auto_reset=True, want_obs=True, max_resets=8)
    assert out["ran"].all()
    assert np.abs(out["rew"][:, 0] - g["rew"]).max() <= COV_TOL, (out["rew"][:, 0], g["rew"])
    assert np.array_equal(out["done"][:, 0], g["done"].astype(bool))
    assert np.array_equal(out["executed"][:, 0], g["n_updates"])
    rb = out["reset_before"][:, 0]
    assert np.array_equal(rb[1:] > 0, g["reset_before"][1:].astype(bool)) and rb[0] == 0      # (the first reset was ours, above)
    info = g["info"]
    for t in range(T):
        for k in ("actual_coverage", "start_coverage"):
            assert abs(float(out[k][t, 0]) - float(info[t][k])) <= COV_TOL, (t, k, out[k][t, 0], info[t][k])
        for k in ("variance_inv", "start_variance_inv"):
            assert abs(float(out[k][t, 0]) - float(info[t][k])) <= 1e-12 * abs(float(info[t][k])), (t, k, out[k][t, 0], info[t][k])
        for k in ("num_steps", "num_sim_steps"):
            assert float(out[k][t, 0]) == float(info[t][k]), (t, k, out[k][t, 0], info[t][k])
        assert bool(out["have_tear"][t, 0]) == bool(info[t]["have_tear"]) and bool(out["out_of_bounds"][t, 0]) == bool(info[t]["out_of_bounds"])
    k = 1
    for t in range(1, T):                                            # what reset() returned in the reference, reset by reset
        if rb[t]:
            assert np.array_equal(out["reset_obs"][0, rb[t] - 1], g["reset_obs"][k].astype(np.float32)), (t, k)
            k += 1
    assert k == len(g["reset_obs"])
    pos, prev, pin = v.batch.get_state()
    assert np.array_equal(pos[0], g["final_pos"]) and np.array_equal(prev[0], g["final_prev"])
    assert np.array_equal(pin[0].astype(bool), g["final_pinned"].astype(bool))
    v.close()


@pytest.mark.parametrize("case", [0, 1, 2])
def test_highest_point_policy_on_the_device_picks_the_references_action_f64(case, oracle_lib):
    """examples/analytic.py:792-808 (HighestPointPolicy.get_action under np.random.seed(k), k = 0..7) captured from the reference
    on a post-reset and a post-action state, tier 1 and both tier-2 sides; the policy evaluated in the kernel (stable arg-max
    over the heights, the flat-cloth target incl. tier 2's (orig_z, orig_y) mapping) picks the same action for the same rank."""
    from gym_cloth_amd.envs import ClothVecEnv
    g = oracle_lib.load_golden("g_highest_point.npz")
    tier, seed = str(g["tiers"][case]), int(g["seeds"][case])
    E = 8
    v = ClothVecEnv(base_cfg(tier, seed), n_envs=E, precision="f64")
    v.seed([seed] * E)
    v.reset()
    assert bool(v.init_side[0]) == bool(g["init_side"][case])
    for si in range(2):
        pos = g["c%d_s%d_pos" % (case, si)]
        if si == 0:
            assert np.array_equal(v.batch.get_state()[0][0], pos)     # our reset == the reference's (pinned elsewhere too)
        v.batch.set_state(pos[None].repeat(E, 0), pos[None].repeat(E, 0), np.zeros((E, 625), dtype=np.uint8))
        v._ep_done[:] = False
        picks = g["c%d_s%d_pick" % (case, si)].astype(np.int32)       # np.random.seed(k); np.random.randint(5)
        assert len(set(picks.tolist())) >= 3
        out = v.step_many(policy="highest_point", n_actions=1, policy_choices=picks[None, :], auto_reset=False)
        assert np.array_equal(out["actions"][0], g["c%d_s%d_act" % (case, si)]), (si, out["actions"][0], g["c%d_s%d_act" % (case, si)])
    v.close()


def test_coverage_reward_type_matches_reference_f64(oracle_lib):
    """reward_type 'coverage' (cloth_env.py:656-662: the reward is the coverage itself, not its change) over the reference's
    captured steps: same rewards, same coverage."""
    from gym_cloth_amd.envs import ClothVecEnv
    g = oracle_lib.load_golden("g_coverage_reward.npz")
    cfg = base_cfg("tier1", 1337)
    cfg["env"]["reward_type"] = "coverage"
    for fused in (False, True):
        v = ClothVecEnv(cfg, n_envs=1, precision="f64")
        v.seed([1337]); v.reset()
        if fused:
            out = v.step_many(np.ascontiguousarray(g["act"][:, None, :]), auto_reset=False)
            rew, cov, done = out["rew"][:, 0], out["actual_coverage"][:, 0], out["done"][:, 0]
        else:
            rew, cov, done = [], [], []
            for a in g["act"]:
                _, r, d, info = v.step(a[None])
                rew.append(r[0]); cov.append(info["actual_coverage"][0]); done.append(d[0])
        assert np.abs(np.asarray(rew) - g["rew"]).max() <= COV_TOL, (fused, rew, g["rew"])
        assert np.abs(np.asarray(cov) - g["coverage"]).max() <= COV_TOL and np.array_equal(np.asarray(done), g["done"].astype(bool))
        v.close()


def test_save_state_start_state_round_trip_tier2_f64(tmp_path):
    """cloth_env.py:343-350 / :120-124 / :736-741: save_state of a tier-2 cloth (its rest lengths were measured on the noisy sheet,
    cloth.pyx:417) and a new env constructed with start_state_path: reset() restores particles, previous positions, pins AND the
    rest lengths, skips the reset actions, and the next step is bit-identical to the saved env's."""
    from gym_cloth_amd.envs import ClothEnv
    cfg = base_cfg("tier2", 1338)
    a = ClothEnv(cfg, precision="f64")
    a.seed(1338); a.reset()
    a.step(np.array([0.3, -0.2, 0.4, 0.3]))
    path = str(tmp_path / "cloth_state.npz")
    a.save_state(path)
    pos_a, prev_a, pin_a = a._vec.batch.get_state()
    rest_a = a._vec.batch.get_rest()
    flat = a._vec.batch.init_grid(1)[1]
    assert not np.array_equal(rest_a[0], flat), "tier-2 rest lengths carry the sheet's noise"
    b = ClothEnv(cfg, precision="f64", start_state_path=path)
    b.seed(7)
    obs = b.reset()
    pos_b, prev_b, pin_b = b._vec.batch.get_state()
    assert np.array_equal(pos_a, pos_b) and np.array_equal(prev_a, prev_b) and np.array_equal(pin_a, pin_b)
    assert np.array_equal(b._vec.batch.get_rest(), rest_a)
    assert np.array_equal(obs, pos_b[0].reshape(-1)) and b.num_steps == 0 and b.num_sim_steps == 0
    act = np.array([-0.1, 0.25, -0.5, 0.2])
    oa, ra, da, ia = a.step(act)
    ob, rb, db, ib = b.step(act)
    assert np.array_equal(oa, ob) and da == db
    assert ia["actual_coverage"] == ib["actual_coverage"] and ia["variance_inv"] == ib["variance_inv"]
    assert np.array_equal(a._vec.batch.get_state()[1], b._vec.batch.get_state()[1])
    a.close(); b.close()


def test_physics_facade_springs_and_colour_points(oracle_lib):
    """Cloth.springs (cloth.pyx:134-146, :411-417) and the colour-point arrays (cloth.pyx:147-164, :398-404) of the façade."""
    from gym_cloth_amd.envs import ClothEnv
    g = oracle_lib.load_golden("g_traj_lift_pull_25.npz")
    e = ClothEnv(base_cfg("tier1", 3), precision="f64")
    e.seed(3); e.reset()
    sp = e.cloth.springs
    assert len(sp) == 3502
    assert [s.ptA._i for s in sp] == list(g["spring_a"]) and [s.ptB._i for s in sp] == list(g["spring_b"])
    assert [s.type for s in sp[:6]] == ["STRUCTURAL"] + ["STRUCTURAL"] * 0 + [("STRUCTURAL", "SHEARING", "BENDING")[t] for t in g["spring_type"][1:6]]
    assert np.array_equal(np.array([s.rest_length for s in sp]), g["rest"])
    n_col = len(e.cloth.color_pts)                      # base_cfg: color_pts 'diag1' -> |(1 - x) - y| < 0.05 on the flat grid
    assert n_col == len(e.cloth.colorpts_arr) and n_col + len(e.cloth.noncolorpts_arr) == 625 and 25 <= n_col <= 75
    e.close()


def test_physics_facade_follows_the_rebuilt_cloth_tier2_f64():
    """The reference builds a new Cloth and new Springs in every reset() (cloth_env.py:737-746); tier 2 measures new rest lengths on
    each new noisy sheet (cloth.pyx:94-116, :417) and the points' orig_* are that sheet's positions. The façade's cached views must
    follow: Spring.rest_length == the device's rest table after every reset, orig_* == the positions the cloth was built with."""
    from gym_cloth_amd.envs import ClothEnv
    e = ClothEnv(base_cfg("tier2", 1338), precision="f64")
    e.seed(1338); e.reset()
    r1 = np.array([s.rest_length for s in e.cloth.springs])
    o1 = np.array([[p.orig_x, p.orig_y, p.orig_z] for p in e.cloth.pts])
    assert np.array_equal(r1, e._vec.batch.get_rest(0, 1)[0])
    side1 = e.cloth.init_side
    assert np.all((o1[:, 0] <= 0.005) if side1 else (o1[:, 0] >= 0.995)) and np.ptp(o1[:, 2]) > 0.9     # the vertical sheet, not where the reset left it
    e.step(np.array([0.3, -0.2, 0.4, 0.3]))
    e.reset()
    r2 = np.array([s.rest_length for s in e.cloth.springs])
    o2 = np.array([[p.orig_x, p.orig_y, p.orig_z] for p in e.cloth.pts])
    assert np.array_equal(r2, e._vec.batch.get_rest(0, 1)[0]) and not np.array_equal(r1, r2)
    assert not np.array_equal(o1, o2) and np.ptp(o2[:, 2]) > 0.9
    e.close()
