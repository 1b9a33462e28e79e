"""Full-size parity, inside the driver-run suite: the very handles bench.py times -- 512 x 25x25 tier 1 (BASELINE configs[2]), 512 x
25x25 tier 2 (configs[3]'s per-GPU shape), 1 024 x 50x50 (configs[4]) and the 1 536 / 2 048-cloth companions -- are created WITHOUT
any CLOTHHIP_DEBUG_* override, so that clothhip_create's own batch-size pick decides the stepper variant, and

  * fp32 (what the bench's `value` runs): clothhip_last_variant must name the variant the bench record names (config.variant), and an
    episode launch with in-kernel resets over the whole batch -- several generations of workgroups where the batch exceeds what is
    resident -- must equal the STANDARD fp32 variant's launch bit for bit (records, observations, particles); the standard variant
    is pinned to the reference by tests/test_gpu_parity.py;
  * fp64 (the reference's arithmetic): one whole bench step of the full batch through the episode launch, then a seeded random sample
    of 64 envs replayed by the CPU oracle from their pre-step states: positions, previous positions and update() counts bit for bit.

Reference loop matched: ClothEnv.step (cloth_env.py:472-515) over Cloth.update (cloth.pyx:169-214)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DEBUG_VARS = ("CLOTHHIP_DEBUG_LEAN", "CLOTHHIP_DEBUG_W8", "CLOTHHIP_DEBUG_LARGE2", "CLOTHHIP_DEBUG_TAB_LDS", "CLOTHHIP_DEBUG_CELL_COPY",
              "CLOTHHIP_DEBUG_REST_REG", "CLOTHHIP_DEBUG_NT1024", "CLOTHHIP_DEBUG_PHASES")


def _bench_env(E, n_side, tier, prec):
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    cfg = bench.bench_cfg(n_side, 0.02 if n_side <= 25 else 0.0095, tier)
    env = ClothVecEnv(cfg, n_envs=E, precision=prec, consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000 + e)          # bench.py's reset streams
    env.reset()
    return cfg, env


def _bench_actions(E, T):
    return np.ascontiguousarray(np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1))


# (E, grid, tier) -> what clothhip_create must pick by itself for the fp32 handle on a 256-CU device: threads per cloth, particles per
# thread, table mode, LEAN arithmetic, cloths per CU at least -- the variants BENCH_r04.json / profiles/r0x_bench.json name
PICKS = [
    (512, 25, "tier1", dict(threads=512, particles_per_thread=2, table_mode=2, lean=True, cloths_per_cu=2)),     # the headline
    (512, 25, "tier2", dict(threads=512, particles_per_thread=2, table_mode=1, lean=False, cloths_per_cu=2)),    # per-env rest tables
    (1024, 50, "tier1", dict(threads=512, particles_per_thread=5, table_mode=4, lean=True, cloths_per_cu=2)),    # configs[4], two per CU
    (1536, 25, "tier1", dict(threads=256, particles_per_thread=3, table_mode=-3, lean=True, cloths_per_cu=6)),   # six per CU, one generation
    (2048, 25, "tier1", dict(threads=256, particles_per_thread=3, table_mode=-1, lean=True, cloths_per_cu=4)),   # four per CU, two generations
]


@pytest.mark.parametrize("E,n_side,tier,want", PICKS, ids=["512x25-tier1", "512x25-tier2", "1024x50", "1536x25", "2048x25"])
def test_picked_f32_variant_is_the_benched_one_and_equals_the_standard_variant(E, n_side, tier, want, monkeypatch):
    for v in DEBUG_VARS:
        monkeypatch.delenv(v, raising=False)
    T = 1 if n_side == 50 else 2
    acts = _bench_actions(E, T)
    runs = []
    for standard in (False, True):
        if standard:                                                 # the standard arithmetic at its own layout (one large-grid cloth per CU)
            monkeypatch.setenv("CLOTHHIP_DEBUG_LEAN", "0")
            monkeypatch.setenv("CLOTHHIP_DEBUG_LARGE2", "0")
        cfg, env = _bench_env(E, n_side, tier, "f32")
        out = env.step_many(acts, auto_reset=True)                   # actions, terminal tests and episode resets in the kernel
        var = env.batch.last_variant()
        runs.append((var, out["rew"].copy(), out["executed"].copy(), out["done"].copy(), out["actual_coverage"].copy(), out["obs"].copy(),
                     out["reset_before"].copy(), [x.copy() for x in env.batch.get_state()]))
        env.close()
    (vp, *a), (vs, *b) = runs
    if vp["n_cus"] == 256:                                           # (the pick depends on the device's CU count: MI355X)
        for k, v in want.items():
            assert (vp[k] >= v) if k == "cloths_per_cu" else (vp[k] == v), (k, vp)
        assert vp["fused"] == (2 if tier == "tier2" else 1) and vp["precision"] == "f32", vp
        # (round 6) every benched fp32 handle runs the grid-specialised build of its variant; the standard-arithmetic run it is compared with
        # below is the specialised tier-2 build at 25x25 (same variant as tier 2's pick) and a generic build at 50x50
        assert vp["spec_n_side"] == n_side, vp
        generations = -(-E // (vp["cloths_per_cu"] * vp["n_cus"]))
        assert generations == (2 if E in (2048, 1024) else 1), (generations, vp)
    assert not vs["lean"], vs
    assert a[1].sum() > 200 * E                                      # the launch did real work: > 200 update() calls per env on average
    for x, y in zip(a[:6], b[:6]):
        assert np.array_equal(x, y, equal_nan=True)
    for x, y in zip(a[6], b[6]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("E,n_side,tier", [(512, 25, "tier1"), (512, 25, "tier2"), (1024, 50, "tier1")],
                         ids=["512x25-tier1", "512x25-tier2", "1024x50"])
def test_full_batch_bench_step_f64_sample_matches_oracle(E, n_side, tier, oracle_lib, monkeypatch):
    for v in DEBUG_VARS:
        monkeypatch.delenv(v, raising=False)
    from gym_cloth_amd.envs import decode_actions
    cfg, env = _bench_env(E, n_side, tier, "f64")
    ev, c = cfg["env"], cfg["cloth"]
    ocfg = {"n_side": n_side, "width": c["width"], "height": c["height"], "density": c["density"], "ks": c["ks"],
            "damping": c["damping"], "thickness": c["thickness"], "plane_friction": c["plane_friction"],
            "tear_thresh": c["tear_thresh"], "frames_per_sec": cfg["frames_per_sec"], "simulation_steps": cfg["simulation_steps"],
            "gravity": -9.8, "minimum_z": 0.0, "grip_radius": ev["grip_radius"]}
    acts = _bench_actions(E, 1)
    pos0, prev0, pin0 = env.batch.get_state()
    tear0 = np.array(env.batch.tear).copy()
    rest = env.batch.get_rest() if tier == "tier2" else None
    if env.batch.fused_supported:
        out = env.step_many(acts, auto_reset=False)                  # ONE bench step of the whole batch, episode-launch kernel
        ex = out["executed"][0]
        assert out["ran"][0].all()
    else:                                                            # (a grid whose in-kernel metrics do not fit the CU's LDS: none of the
        env.step(acts[0], auto_reset=False)                          #  benched ones since the fp64 large-grid variants keep the hull stack as indices)
        ex = env.last_executed.copy()
    assert env.batch.fused_supported                                 # 512 x 25x25 and 1 024 x 50x50 alike
    var = env.batch.last_variant()
    # (round 6: the flat tiers' fp64 handle of the 25x25 class runs the fp64 LEAN build -- stencil recomputed, rest = palette bits + ulp offset --;
    #  per-env rest tables (tier 2) and the large grids keep the standard arithmetic)
    assert var["precision"] == "f64" and var["lean"] == (tier != "tier2" and n_side == 25) and (var["fused"] >= 1) == bool(env.batch.fused_supported), var
    if var["n_cus"] == 256:
        assert var["spec_n_side"] == (25 if var["lean"] else 0), var          # the fp64 LEAN build of 25x25 is grid-specialised, the others generic
    if var["n_cus"] == 256:
        assert var["threads"] == 512 and var["particles_per_thread"] == (2 if n_side == 25 else 5), var
    pos1, prev1, _ = env.batch.get_state()
    assert ex.sum() > 300 * E
    d = decode_actions(acts[0], [-1.] * 4, [1.] * 4, True, True, ev["reduce_factor"], ev["iters_up"], ev["iters_up_rest"],
                       ev["iters_pull_max"], ev["iters_grip_rest"], ev["iters_rest"])
    sample = np.sort(np.random.RandomState(20251004 + E + n_side).choice(E, 64, replace=False))
    cloths, sched, delta = [], np.zeros((64, 5), dtype=np.int32), np.zeros((64, 3))
    for k, e in enumerate(sample):
        oc = oracle_lib.OracleCloth(ocfg)
        oc.set_state(pos0[e], prev0[e], pin0[e], None if rest is None else rest[e])
        oc.have_tear = bool(tear0[e])
        ng = oc.grab_top(float(d["x"][e]), float(d["y"][e]))
        sched[k] = d["bounds"][e] if ng > 0 else 0
        delta[k] = (0.0025, d["x_dir_r"][e], d["y_dir_r"][e])
        cloths.append(oc)
    threads = max(1, min(len(os.sched_getaffinity(0)), oracle_lib.lib().oracle_max_threads(), 64))
    exo = oracle_lib.batch_run_schedule(cloths, sched, delta, True, threads)
    busy = 0
    for k, e in enumerate(sample):
        op, oq, _ = cloths[k].get_state()
        assert exo[k] == ex[e], (e, int(exo[k]), int(ex[e]))
        assert np.array_equal(pos1[e], op) and np.array_equal(prev1[e], oq), (e, float(np.abs(pos1[e] - op).max()))
        busy += int(ex[e] > 0)
    assert busy >= 20                                                # a third of the random pick points hit the cloth
    env.close()


def test_relaxed_order_companion_runs_and_is_labelled(monkeypatch):
    """bench.py's relaxed-order companion (ClothBatch.set_relaxed_order / clothhip_set_relaxed_order, per handle: self-collision in Jacobi order, strain limit in coloured order) is a
    MEASUREMENT, not a mode with a parity claim: this only checks that the companion kernel is the one that runs (clothhip_last_variant:
    episode flavour 3), that it is refused where it does not exist (fp64), that it leaves sane cloths behind (finite positions, coverage
    comparable with the exact order's) and that its results DO differ from the exact order's -- so that nobody mistakes one for the
    other."""
    import bench
    from gym_cloth_amd._lib import ClothHipError
    from gym_cloth_amd.envs import ClothVecEnv
    for v in DEBUG_VARS:
        monkeypatch.delenv(v, raising=False)
    E, T = 64, 2
    acts = _bench_actions(E, T)
    outs = {}
    for relaxed in (False, True):
        cfg, env = _bench_env(E, 25, "tier1", "f32")
        if relaxed:
            env.batch.set_relaxed_order(True)
        out = env.step_many(acts, auto_reset=False)
        var = env.batch.last_variant()
        assert var["fused"] == (3 if relaxed else 1) and var["lean"] and var["threads"] == 512, var
        pos = env.batch.get_state()[0]
        assert np.isfinite(pos).all()
        outs[relaxed] = (out["executed"].copy(), out["actual_coverage"].copy(), pos.copy())
        env.close()
    # (the first action only: both start from the same post-reset states; from the second action on the trajectories have parted --
    #  and the relaxed kernel sums a particle's hits in the order the hash build happened to rank them, so its own low bits vary from
    #  run to run: one more reason why it carries no parity claim)
    assert np.array_equal(outs[False][0][0] > 0, outs[True][0][0] > 0)               # the same envs grabbed something
    assert not np.array_equal(outs[False][2], outs[True][2])                          # ... but these are different trajectories
    assert np.median(np.abs(outs[False][1][0] - outs[True][1][0])) < 0.1              # of the same physics (coverage stays comparable)
    cfg64 = bench.bench_cfg(25, 0.02, "tier1")
    env = ClothVecEnv(cfg64, n_envs=4, precision="f64", consume_domrand_draws=False)
    env.batch.set_relaxed_order(True)
    env.reset()                                                                       # (the per-step path is not affected)
    with pytest.raises(ClothHipError):
        env.step_many(_bench_actions(4, 1), auto_reset=False)
    env.close()


def test_time_sliced_launch_goes_out_per_generation_and_equals_the_single_launch(monkeypatch):
    """A time-sliced episode launch over more cloths than are resident is issued as one launch per generation (launch_run: workgroup
    0 of a launch is env `e0`): 1 100 cloths at two per CU = 512 + 512 + 76. With a slice long enough for every action the result
    must equal the single launch over all workgroups (CLOTHHIP_DEBUG_ONE_LAUNCH) bit for bit -- records, observations, particles."""
    for v in DEBUG_VARS:
        monkeypatch.delenv(v, raising=False)
    monkeypatch.setenv("CLOTHHIP_DEBUG_LEAN", "8")                   # the eight-wave build: two cloths per CU whatever the batch size
    E, T = 1100, 2
    acts = _bench_actions(E, T)
    runs = []
    for one_launch in (False, True):
        if one_launch:
            monkeypatch.setenv("CLOTHHIP_DEBUG_ONE_LAUNCH", "1")
        cfg, env = _bench_env(E, 25, "tier1", "f32")
        out = env.step_many(acts, auto_reset=True, time_budget_ms=120000.0)
        var = env.batch.last_variant()
        assert out["ran"].all()
        runs.append((var, out["rew"].copy(), out["executed"].copy(), out["done"].copy(), out["actual_coverage"].copy(), out["obs"].copy(),
                     out["reset_before"].copy(), [x.copy() for x in env.batch.get_state()]))
        env.close()
    (va, *a), (vb, *b) = runs
    # (the library says how many kernel dispatches a launch went out as: 512 + 512 + 76 -> three, the debug override -> one)
    assert va.pop("dispatches") == 3 and vb.pop("dispatches") == 1
    assert va["threads"] == 512 and va["cloths_per_cu"] == 2 and va == vb, (va, vb)
    assert a[1].sum() > 200 * E
    for x, y in zip(a[:6], b[:6]):
        assert np.array_equal(x, y, equal_nan=True)
    for x, y in zip(a[6], b[6]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("E,n_side,want_tab", [(512, 25, 2), (1536, 25, -3), (1024, 50, 4)], ids=["512x25", "1536x25", "1024x50"])
def test_benched_f32_handles_meet_the_oracle_directly_over_one_substep(E, n_side, want_tab, oracle_lib, monkeypatch):
    """The fp32 variants the bench runs -- picked by clothhip_create from the batch size, no override -- against the fp64 ORACLE itself (not
    against another fp32 build): ONE update() of the whole batch from its tier-1 post-reset states (crumpled by the reset pulls: every phase
    acts), a seeded sample of 64 cloths replayed by the oracle from the same fp32 states and the same fp32 rest lengths. After a single substep
    nothing has been amplified: the difference is the stepper's own rounding, <= 6e-7 on positions of O(1) (measured 2e-7; the band of
    tests/test_gpu_parity.py::test_f32_single_substep_on_post_reset_states). Reference: Cloth.update, cloth.pyx:169-214."""
    for v in DEBUG_VARS:
        monkeypatch.delenv(v, raising=False)
    cfg, env = _bench_env(E, n_side, "tier1", "f32")
    pos0, prev0, pin0 = env.batch.get_state()
    rest = env.batch.get_rest()
    env.batch.update(1)
    var = env.batch.last_variant()
    assert var["precision"] == "f32" and var["lean"], var
    if var["n_cus"] == 256:
        assert var["table_mode"] == want_tab, var                 # the variant the bench record names for this batch size
    pos1, prev1, _ = env.batch.get_state()
    c, ev = cfg["cloth"], cfg["env"]
    ocfg = {"n_side": n_side, "width": c["width"], "height": c["height"], "density": c["density"], "ks": c["ks"], "damping": c["damping"],
            "thickness": c["thickness"], "plane_friction": c["plane_friction"], "tear_thresh": c["tear_thresh"],
            "frames_per_sec": cfg["frames_per_sec"], "simulation_steps": cfg["simulation_steps"], "gravity": -9.8, "minimum_z": 0.0,
            "grip_radius": ev["grip_radius"]}
    sample = np.sort(np.random.RandomState(6000 + E + n_side).choice(E, 64, replace=False))
    errs, moved = [], []
    for e in sample:
        oc = oracle_lib.OracleCloth(ocfg)
        oc.set_state(pos0[e], prev0[e], pin0[e], rest[e if rest.shape[0] > 1 else 0])
        oc.update(1)
        op, oq, _ = oc.get_state()
        errs.append(float(np.abs(pos1[e] - op).max()))
        moved.append(float(np.abs(op - pos0[e]).max()))
    print("\nfp32 %s over one substep vs the oracle, 64 of %d cloths: max err %.3e, largest move %.2e" % (var["name"], E, max(errs), max(moved)))
    assert max(moved) > 1e-5                                      # the states are live
    assert max(errs) <= 6e-7, sorted(errs)[-5:]
    env.close()
