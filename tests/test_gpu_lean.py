"""The LEAN stepper variant (fp32, flat tiers; compiled for three and for four cloths per CU and -- eight waves per cloth, window table in
LDS -- for two: chosen by clothhip_create from the batch size, forced here with CLOTHHIP_DEBUG_LEAN): the gather stencil recomputed from the grid position and rest lengths from a three-value palette are a
different HOME for the same numbers, not different arithmetic -- its records and particles equal the standard fp32 variant's bit
for bit, over whole episode launches with resets and over the per-step path; and it steps aside (standard variant, same results)
when the rest table is not a palette (tier 2: per-env rest lengths)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(tier, lean, monkeypatch, E=48, T=5):
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    monkeypatch.setenv("CLOTHHIP_DEBUG_LEAN", str(lean) if lean else "0")           # 3 / 4: the build for three / four cloths per CU
    cfg = bench.bench_cfg(25, 0.02, tier)
    env = ClothVecEnv(cfg, n_envs=E, precision="f32", consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000 + e)
    env.reset()
    acts = np.ascontiguousarray(np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1))
    out = env.step_many(acts, auto_reset=True)                       # episode launch: actions, terminal tests, in-kernel resets
    var_fused = env.batch.last_variant()
    a = np.stack([np.random.RandomState(3000 + e).uniform(-1, 1, size=4) for e in range(E)])
    obs, rew, done, info = env.step(a, auto_reset=False)              # the per-step path (plain stepper variant)
    var_step = env.batch.last_variant()
    res = dict(rew=out["rew"].copy(), executed=out["executed"].copy(), done=out["done"].copy(), cov=out["actual_coverage"].copy(),
               obs=out["obs"].copy(), obs2=obs.copy(), rew2=rew.copy(), exec2=env.last_executed.copy(),
               state=[x.copy() for x in env.batch.get_state()])
    env.close()
    return res, var_fused, var_step


@pytest.mark.parametrize("tier,build", [("tier1", 3), ("tier3", 3), ("tier1", 4), ("tier1", 5), ("tier1", 6), ("tier1", 8), ("tier3", 8)])
def test_lean_variant_is_bit_identical_to_the_standard_f32_variant(tier, build, monkeypatch):
    a, va_f, va_s = _run(tier, 0, monkeypatch)
    b, vb_f, vb_s = _run(tier, build, monkeypatch)
    # the library says which kernel ran: without this the comparison below could not tell "bit-identical" from "never ran"
    assert not va_f["lean"] and not va_s["lean"] and va_f["fused"] >= 1 and va_s["fused"] == 0, (va_f, va_s)
    for v in (vb_f, vb_s):
        if build == 8:                                               # eight waves per cloth, window table in LDS, two cloths per CU
            assert v["lean"] and v["threads"] == 512 and v["particles_per_thread"] == 2 and v["table_mode"] == 2 and v["cloths_per_cu"] >= 2, v
            continue
        assert v["lean"] and v["threads"] == 256 and v["table_mode"] == 3 - build, v
        assert v["cloths_per_cu"] >= build, v                        # three / four cloths resident per CU (a build that needs fewer
                                                                     # registers than its cap may fit one more)
    assert va_f["cloths_per_cu"] == 2, va_f
    # (round 6) a 25x25 fp32 handle runs the GRID-SPECIALISED build of its variant (the grid's sizes as compile-time constants), LEAN and
    # standard arithmetic alike; the specialised builds are held to the generic ones by the test below
    assert vb_f["spec_n_side"] == 25 and vb_s["spec_n_side"] == 25 and va_f["spec_n_side"] == 25 and ",N25>" in vb_f["name"], (vb_f, vb_s, va_f)
    assert a["executed"].sum() > 100000 and a["exec2"].sum() > 10000
    for k in a:
        if k == "state":
            for x, y in zip(a[k], b[k]):
                assert np.array_equal(x, y), k
        else:
            assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("build", [8, 6, 0])
def test_grid_specialised_build_equals_the_generic_build_of_the_same_variant(build, monkeypatch):
    """CLOTHHIP_DEBUG_NOSPEC=1 makes the handle run the generic build (grid sizes from the kernel arguments) of the very same LEAN variant:
    same records, observations and particles as the specialised one, over an episode launch with resets and over the per-step path; and a
    debug phase mask -- something the specialised build has compiled out -- sends the handle to the generic build by itself."""
    monkeypatch.delenv("CLOTHHIP_DEBUG_NOSPEC", raising=False)
    a, va_f, va_s = _run("tier1", build, monkeypatch)
    monkeypatch.setenv("CLOTHHIP_DEBUG_NOSPEC", "1")
    b, vb_f, vb_s = _run("tier1", build, monkeypatch)
    assert va_f["spec_n_side"] == 25 and va_s["spec_n_side"] == 25 and vb_f["spec_n_side"] == 0 and vb_s["spec_n_side"] == 0, (va_f, vb_f)
    assert {k: v for k, v in va_f.items() if k not in ("spec_n_side", "name")} == {k: v for k, v in vb_f.items() if k not in ("spec_n_side", "name")}
    for k in a:
        if k == "state":
            for x, y in zip(a[k], b[k]):
                assert np.array_equal(x, y), k
        else:
            assert np.array_equal(a[k], b[k]), k
    monkeypatch.delenv("CLOTHHIP_DEBUG_NOSPEC", raising=False)
    monkeypatch.setenv("CLOTHHIP_DEBUG_PHASES", "31")                # all phases + "walk every window": not what the specialised build compiled in
    c, vc_f, _ = _run("tier1", build, monkeypatch, E=8, T=1)
    assert vc_f["spec_n_side"] == 0 and vc_f["lean"] == (build != 0), vc_f


def test_lean_handle_steps_aside_for_per_env_rest_tables(monkeypatch):
    """Tier 2 gives every env its own rest lengths: no palette. A handle that wants the lean variant runs the standard one then."""
    a, _, _ = _run("tier2", 0, monkeypatch, E=24, T=3)
    b, vf, vs = _run("tier2", 3, monkeypatch, E=24, T=3)
    assert not vf["lean"] and not vs["lean"], (vf, vs)              # it did step aside (clothhip_last_variant)
    for k in ("rew", "executed", "obs", "obs2", "exec2"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("prec", ["f32", "f64"])
def test_four_wave_standard_variants_equal_the_eight_wave_ones(prec, monkeypatch):
    """The standard arithmetic runs eight waves per cloth (512 threads x 2 particles) for the 25x25 class; the four-wave builds
    (256 x 3, CLOTHHIP_DEBUG_W8=0) are the same arithmetic in another thread layout: identical records and particles, fp32 and fp64."""
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    res = []
    for w8 in ("1", "0"):
        monkeypatch.setenv("CLOTHHIP_DEBUG_LEAN", "0")
        monkeypatch.setenv("CLOTHHIP_DEBUG_W8", w8)
        E, T = 16, 3
        env = ClothVecEnv(bench.bench_cfg(25, 0.02, "tier1"), n_envs=E, precision=prec, consume_domrand_draws=False)
        for e in range(E):
            env.np_randoms[e] = np.random.RandomState(1000 + e)
        env.reset()
        acts = np.ascontiguousarray(np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1))
        out = env.step_many(acts, auto_reset=True)
        var = env.batch.last_variant()
        assert not var["lean"] and var["threads"] == (512 if w8 == "1" else 256) and var["particles_per_thread"] == (2 if w8 == "1" else 3), var
        assert var["cloths_per_cu"] >= 2, var
        res.append((out["rew"].copy(), out["executed"].copy(), out["obs"].copy(), [x.copy() for x in env.batch.get_state()]))
        env.close()
    a, b = res
    assert a[1].sum() > 20000
    assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for x, y in zip(a[3], b[3]):
        assert np.array_equal(x, y)
