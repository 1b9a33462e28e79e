"""CPU tests of everything on the product side that does not need a GPU: the C-ABI library loads and exports
every symbol include/clothhip.h declares, its pure-host entry points agree with the reference goldens, it fails
loudly without a device, and the host-side env arithmetic matches the reference captures."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    if not os.path.exists(os.path.join(ROOT, "gym_cloth_amd", "libclothhip.so")):
        ge.build()
    from gym_cloth_amd import _lib
    return _lib


def golden_cfg(g):
    c = g["cfg"]
    return {"cloth": {"num_width_points": c["n_side"], "num_height_points": c["n_side"], "width": c["width"],
                      "height": c["height"], "density": c["density"], "ks": c["ks"], "damping": c["damping"],
                      "thickness": c["thickness"], "plane_friction": c["plane_friction"],
                      "tear_thresh": c["tear_thresh"]},
            "frames_per_sec": c["frames_per_sec"], "simulation_steps": c["simulation_steps"],
            "env": {"grip_radius": c["grip_radius"]}}


def test_abi_exports_every_declared_symbol(lib):
    """Each function declared in include/clothhip.h is exported by libclothhip.so and bound in _lib.SYMBOLS."""
    hdr = open(os.path.join(ROOT, "include", "clothhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(clothhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    L = lib.load()
    bound = {n for n, _, _ in lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert getattr(L, name) is not None
    assert L.clothhip_abi_version() == lib.ABI_VERSION == 7
    assert C.sizeof(lib.ClothSchedule) == 64 and C.sizeof(lib.ClothParams) == 104


def _layout(lib, n_side, precision, n_envs=512, n_cus=256, thickness=0.02):
    L = lib.load()
    p = lib.params_from_cfg({"cloth": {"num_width_points": n_side, "num_height_points": n_side, "width": 1, "height": 1,
                                       "density": 200.0, "ks": 1e4, "damping": 2.0, "thickness": thickness, "plane_friction": 1.0,
                                       "tear_thresh": 2.0},
                             "frames_per_sec": 30, "simulation_steps": 30, "env": {"grip_radius": 0.003}})
    out = np.zeros(24, dtype=np.int32)
    lib.check(L.clothhip_selftest_layout(C.byref(p), 0 if precision == "f64" else 1, n_envs, n_cus, lib.i32p(out), 24))
    keys = ("threads", "ppt", "tab", "rest_reg", "cell_copy", "lds_bytes", "HT", "scratch_have", "scratch_need", "fused_ok")
    return {"std": dict(zip(keys, out[0:10].tolist())), "lean": bool(out[10]), "lean_r": int(out[11]),
            "lean_lay": dict(zip(keys, out[12:22].tolist())), "fused_supported": bool(out[22]), "fits": bool(out[23])}


def test_layout_sweep_every_grid_keeps_the_episode_launches(lib):
    """The variant / LDS-layout plan of clothhip_create (pure host logic, clothhip_selftest_layout) over every grid size and both
    precisions: the layout fits the CU, the LDS the in-kernel metrics borrow is large enough wherever the budget allows it at all --
    round 4 lost the episode launches for fp64 21, 22, 30-32 and fp32 41-43 when the scratch moved behind the window table (ADVICE
    r4) -- , two cloths of the 25x25 class stay co-resident, and no residency is picked whose layout does not fit that often."""
    lost = []
    for prec in ("f32", "f64"):
        for n in range(3, 65):
            lay = _layout(lib, n, prec, thickness=0.02 if n <= 27 else 0.0095)
            if prec == "f64" and n >= 53:                     # fp64 state + tables beyond the CU's 160 KiB: clothhip_create refuses these
                assert not lay["fits"]
                continue
            assert lay["fits"], (prec, n, lay)
            s_ = lay["std"]
            budget = 80 * 1024 if n * n <= 768 else 160 * 1024
            assert s_["lds_bytes"] <= budget, (prec, n, s_)
            assert s_["fused_ok"] == (s_["scratch_have"] >= s_["scratch_need"])
            if lay["lean"]:
                assert lay["lean_lay"]["fused_ok"], (prec, n, lay)
                r = lay["lean_r"]
                # r cloths per CU: LDS is allocated in 1 280-byte granules (128 per CU), each cloth gets floor(128 / r) of them at most
                assert -(-lay["lean_lay"]["lds_bytes"] // 1280) <= 128 // max(r, 1), (prec, n, lay)
            if not lay["fused_supported"]:
                lost.append((prec, n))
    # the episode launches run on everything that fits the CU at all: fp64 46 .. 52 -- BASELINE configs[4]'s 50x50 among them -- and
    # fp32 64x64 since the fp64 large-grid variants and the 1024 x 4 variants keep the in-kernel metrics' hull stack as indices
    assert lost == [], lost
    for prec, n in (("f64", 21), ("f64", 22), ("f64", 30), ("f64", 31), ("f64", 32), ("f32", 41), ("f32", 42), ("f32", 43)):
        assert _layout(lib, n, prec, thickness=0.0095 if n > 27 else 0.02)["fused_supported"], (prec, n)
    # the headline layouts themselves: eight-wave LEAN at two per CU for 512 cloths, four-wave LEAN builds for the larger batches
    h = _layout(lib, 25, "f32", 512)
    assert h["lean"] and h["lean_r"] == 2 and h["lean_lay"]["threads"] == 512 and h["lean_lay"]["tab"] == 2 and h["lean_lay"]["lds_bytes"] <= 80 * 1024
    assert _layout(lib, 25, "f32", 1536)["lean_r"] == 6 and _layout(lib, 25, "f32", 2048)["lean_r"] == 4
    big = _layout(lib, 50, "f32", 1024, thickness=0.0095)
    assert big["lean"] and big["lean_lay"]["threads"] == 512 and big["lean_lay"]["tab"] == 4 and big["lean_lay"]["lds_bytes"] <= 80 * 1024
    # 27x27: the five- and six-per-CU layouts do not fit five / six times: the pick must not count on them
    for E in (1280, 1536, 3072):
        l27 = _layout(lib, 27, "f32", E)
        assert -(-l27["lean_lay"]["lds_bytes"] // 1280) <= 128 // l27["lean_r"], (E, l27)


def test_no_device_fails_loudly(lib):
    """There is no CPU fallback: without a HIP device, creating a batch raises (it never degrades silently)."""
    L = lib.load()
    if L.clothhip_device_count() > 0:
        pytest.skip("a HIP device is visible here")
    from gym_cloth_amd import ClothBatch, ClothHipError
    g_cfg = {"cloth": {"num_width_points": 25, "num_height_points": 25, "width": 1, "height": 1, "density": 200.0,
                       "ks": 1e4, "damping": 2.0, "thickness": 0.02, "plane_friction": 1.0, "tear_thresh": 2.0},
             "frames_per_sec": 30, "simulation_steps": 30, "env": {"grip_radius": 0.003}}
    with pytest.raises(ClothHipError):
        ClothBatch(g_cfg, n_envs=2)


def test_bad_params_raise_value_error(lib):
    L = lib.load()
    p = lib.params_from_cfg({"cloth": {"num_width_points": 2, "num_height_points": 2, "width": 1, "height": 1,
                                       "density": 200.0, "ks": 1e4, "damping": 2.0, "thickness": 0.02,
                                       "plane_friction": 1.0, "tear_thresh": 2.0},
                             "frames_per_sec": 30, "simulation_steps": 30})
    pos = np.zeros((4, 3))
    with pytest.raises(ValueError):
        lib.check(L.clothhip_init_grid(C.byref(p), 1, 0, None, lib.dp(pos), None))
    p.n_side = 25
    with pytest.raises(ValueError):                       # init.type outside tier1/2/3: ValueError, cloth.pyx:131-132
        lib.check(L.clothhip_init_grid(C.byref(p), 7, 0, None, lib.dp(np.zeros((625, 3))), None))
    with pytest.raises(AssertionError):                   # height == width, cloth.pyx:91
        lib.params_from_cfg({"cloth": {"num_width_points": 25, "num_height_points": 24}})


@pytest.mark.parametrize("name", ["g_env_tier1_1337.npz", "g_env_tier2_1337.npz", "g_env_tier2_1338.npz",
                                  "g_traj_fold_50.npz"])
def test_host_init_grid_and_topology_match_reference(name, lib, oracle_lib):
    """clothhip_init_grid / clothhip_spring_topology (pure host, double) against the reference's Cloth.__init__."""
    from gym_cloth_amd import seeding
    L = lib.load()
    g = oracle_lib.load_golden(name)
    p = lib.params_from_cfg(golden_cfg(g))
    P = p.n_side ** 2
    S = len(g["rest"])
    pos, rest = np.empty((P, 3)), np.empty(S)
    if "tier" in g:
        rng, _ = seeding.np_random(int(g["seed"]))
        side = rng.rand() > 0.5
        tier = {"tier1": 1, "tier2": 2, "tier3": 3}[str(g["tier"])]
        draws = rng.rand(P) if tier == 2 else None
        lib.check(L.clothhip_init_grid(C.byref(p), tier, int(side), lib.dp(draws), lib.dp(pos), lib.dp(rest)))
        assert np.array_equal(pos, g["init_pos"])
    else:
        lib.check(L.clothhip_init_grid(C.byref(p), 1, 0, None, lib.dp(pos), lib.dp(rest)))
        assert np.array_equal(pos, g["cp_pos"][0])
        a, b, t = np.empty(S, np.int32), np.empty(S, np.int32), np.empty(S, np.uint8)
        lib.check(L.clothhip_spring_topology(C.byref(p), lib.i32p(a), lib.i32p(b), lib.u8p(t)))
        assert np.array_equal(a, g["spring_a"]) and np.array_equal(b, g["spring_b"]) and np.array_equal(t, g["spring_type"])
    assert np.array_equal(rest, g["rest"])


def test_hull_area_matches_qhull_goldens(lib, oracle_lib):
    """clothhip_hull_area (monotone chain) vs scipy.spatial.ConvexHull(...).volume as the reference computes
    coverage (cloth_env.py:628-638), on flat / folded / clipped / out-of-bounds / blob states: <= 2e-16 abs."""
    L = lib.load()
    g = oracle_lib.load_golden("g_metrics.npz")
    for i, st in enumerate(g["pos"]):
        xy = np.ascontiguousarray(np.clip(st[:, :2], 0, 1))
        assert abs(L.clothhip_hull_area(lib.dp(xy), len(xy)) - g["coverage"][i]) <= 2e-16 + 1e-15 * g["coverage"][i], i
    assert L.clothhip_hull_area(lib.dp(np.zeros((5, 2))), 5) == 0.0           # degenerate hull -> coverage 0
    line = np.ascontiguousarray(np.stack([np.linspace(0, 1, 9), np.linspace(0, 1, 9)], 1))
    assert L.clothhip_hull_area(lib.dp(line), 9) == 0.0


def test_metrics_formulas_match_reference(oracle_lib):
    """variance_inv and out-of-bounds as cloth_env.py:1020-1084 computes them (numpy restatement used by the env)."""
    g = oracle_lib.load_golden("g_metrics.npz")
    for i, st in enumerate(g["pos"]):
        var = np.var(st[:, 2])
        vinv = 1000.0 if var < 0.000001 else 0.001 / var
        assert abs(vinv - g["variance_inv"][i]) <= 1e-12 * max(1.0, g["variance_inv"][i])
        oob = (st[:, 0].max() >= 1.25 or st[:, 0].min() < -0.25 or st[:, 1].max() >= 1.25 or st[:, 1].min() < -0.25 or
               st[:, 2].max() >= 1 or st[:, 2].min() < 0)
        assert bool(oob) == bool(g["oob"][i])


def test_decode_actions_matches_reference_captures(oracle_lib):
    """iters_pull and the phase boundaries for every action the reference executed (incl. tier 3's fractional
    iters_up), + clipping of out-of-range actions (cloth_env.py:402-415)."""
    from gym_cloth_amd.envs import decode_actions
    for name in ["g_env_tier1_1337.npz", "g_env_tier2_1337.npz", "g_env_tier3_1337.npz", "g_env_tier3_1339.npz"]:
        g = oracle_lib.load_golden(name)
        e = g["cfg"]["env"]
        for k in range(len(g["act"])):
            iu = float(g["act_iters_up"][k])
            d = decode_actions(g["act"][k][None], [-1.] * 4, [1.] * 4, True, True, e["reduce_factor"], iu,
                               e["iters_up_rest"], e["iters_pull_max"], e["iters_grip_rest"], e["iters_rest"])
            if int(g["act_n_updates"][k]) > 1 and not bool(g["act_tear"][k]):
                assert d["bounds"][0, 4] == int(g["act_n_updates"][k]), (name, k)
            assert d["bounds"][0, 0] == int(np.ceil(iu))
    d = decode_actions(np.array([[3.0, -7.0, 0.5, 0.0]]), [-1.] * 4, [1.] * 4, True, True, 0.002, 50, 80, 400, 300, 1000)
    assert d["x"][0] == 1.0 and d["y"][0] == 0.0
    d0 = decode_actions(np.array([[0.0, 0.0, 0.0, 0.0]]), [-1.] * 4, [1.] * 4, True, True, 0.002, 50, 80, 400, 300, 1000)
    assert d0["iters_pull"][0] == 0 and d0["bounds"][0].tolist() == [50, 130, 130, 430, 1430]


def test_seeding_is_deterministic_and_matches_golden_rng(oracle_lib):
    """gym 0.12.1's np_random restated (sha512 of str(seed) -> uint32 words -> RandomState). The golden env
    fixtures were produced with the same restatement; real-gym parity is documented as unpinned."""
    from gym_cloth_amd import seeding
    r1, s1 = seeding.np_random(1337)
    r2, _ = seeding.np_random(1337)
    assert s1 == 1337 and r1.rand() == r2.rand()
    g = oracle_lib.load_golden("g_env_tier1_1337.npz")
    r3, _ = seeding.np_random(1337)
    assert (r3.rand() > 0.5) == bool(g["init_side"])
    with pytest.raises(ValueError):
        seeding.np_random(-1)


def test_schedule_helpers():
    from gym_cloth_amd import make_schedules, schedule_bounds
    assert schedule_bounds(50, 80, 196, 300, 1000) == (50, 130, 326, 626, 1626)
    assert schedule_bounds(237.4, 80, 10, 300, 1000) == (238, 318, 328, 628, 1628)   # tier 3: i < 237.4 <=> i < 238
    s = make_schedules(3, n_total=7, active=1)
    assert s.shape == (3,) and s.dtype.itemsize == 64 and (s["n_total"] == 7).all()


def test_gym_registration_shim(monkeypatch):
    """gym_cloth/__init__.py:1-5 registers 'cloth-v0'; the same happens here whenever gym is importable (it is not a
    dependency of the stepper, so the registry is stubbed)."""
    import sys
    import types
    calls = []
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda id, entry_point: calls.append((id, entry_point))
    gym = types.ModuleType("gym"); envs = types.ModuleType("gym.envs")
    gym.envs = envs; envs.registration = reg
    for k, v in (("gym", gym), ("gym.envs", envs), ("gym.envs.registration", reg)):
        monkeypatch.setitem(sys.modules, k, v)
    import gym_cloth_amd
    assert gym_cloth_amd.register_gym_env() is True
    assert calls == [("cloth-v0", "gym_cloth_amd.envs:ClothEnv")]
    from gym_cloth_amd import envs as our_envs
    assert hasattr(our_envs, "ClothEnv")


def test_shipped_yaml_schema_matches_reference_keys():
    """cfg/t{1,2,3}.yaml: same sections and keys as the reference's cfg/t1_rgbd.yaml (listed here: the reference tree is
    not available where the tests run), values of the physics keys as shipped there."""
    import yaml
    want = {"cloth": {"damping", "density", "ks", "enable_structural", "enable_shearing", "enable_bending", "orientation",
                      "width", "height", "num_width_points", "num_height_points", "thickness", "pin_cond", "color_pts",
                      "plane_friction", "tear_thresh"},
            "env": {"max_actions", "max_z_threshold", "iters_up", "iters_up_rest", "iters_pull_max", "iters_grip_rest",
                    "iters_rest", "updates_per_move", "reduce_factor", "grip_radius", "reward_type", "force_grab",
                    "clip_act_space", "delta_actions", "obs_type", "oracle_reveal", "use_depth", "use_dom_rand", "use_rgbd"},
            "init": {"type", "debug_matplotlib", "render_opengl"}, "log": {"level", "file"}}
    for n in (1, 2, 3):
        cfg = yaml.safe_load(open(os.path.join(ROOT, "cfg", "t%d.yaml" % n)))
        assert set(cfg) == set(want) | {"frames_per_sec", "simulation_steps", "seed"}
        for sec, keys in want.items():
            assert set(cfg[sec]) == keys, (n, sec)
        assert cfg["init"]["type"] == "tier%d" % n and cfg["env"]["obs_type"] == "1d"
        c = cfg["cloth"]
        assert (c["ks"], c["damping"], c["density"], c["thickness"], c["plane_friction"], c["tear_thresh"]) == \
               (10000.0, 2.0, 200.0, 0.02, 1.0, 2.0)


def test_step_many_script_drawing_is_rng_neutral():
    """The reset scripts step_many pre-draws leave every env RNG where it was; script 0 holds exactly the draws the host
    reset makes (cloth.pyx:75; cloth_env.py:851-877), script 1 continues from the state after TWO pulls, and cached scripts
    are reused by the next call."""
    from gym_cloth_amd.envs import ClothVecEnv
    v = ClothVecEnv.__new__(ClothVecEnv)                     # host-only pieces: no device needed
    v.E, v.P, v._init_type, v._consume_domrand, v.iters_up = 3, 625, "tier1", False, 50
    v.np_randoms = [np.random.RandomState(40 + e) for e in range(3)]
    v._pending = [None] * 3
    before = [r.get_state()[1].copy() for r in v.np_randoms]
    sc = v._prepare_scripts(3)
    assert sc.shape == (3, 3) and sc["valid"].all()
    assert all(np.array_equal(b, r.get_state()[1]) for b, r in zip(before, v.np_randoms))
    ref = np.random.RandomState(40)
    for s in range(2):
        ref.rand()                                           # init_side
        st2 = None
        for k in range(3):
            if k == 2:
                st2 = ref.get_state()
            assert sc[0, s]["pull"][k]["point"] == ref.randint(625)
            assert sc[0, s]["pull"][k]["dx"] == ClothVecEnv._randval_minabs(ref, -0.2, 0.2, 0.08)
            assert sc[0, s]["pull"][k]["dy"] == ClothVecEnv._randval_minabs(ref, -0.2, 0.2, 0.08)
        assert sc[0, s]["n_pulls"] == 3 and sc[0, s]["pull"][2]["need_coverage"] == 1
        ref.set_state(st2)                                   # the next script assumes the third pull did not happen
    sc2 = v._prepare_scripts(4)                              # longer chain: the first three are the cached ones
    assert np.array_equal(sc2[:, :3], sc) and sc2["valid"].all()
    assert all(np.array_equal(b, r.get_state()[1]) for b, r in zip(before, v.np_randoms))


def _mt_image(rng):
    st = rng.get_state()
    img = np.zeros(625, dtype=np.uint32)
    img[:624], img[624] = st[1], st[2]
    return img


def test_device_rng_functions_match_numpy(lib):
    """csrc/cloth_rng.hpp (the code the kernel runs to draw episode resets) against numpy's legacy RandomState, on the
    host: raw MT19937 words across several twists, rand(), uniform(), randint(625) (masked rejection), the rejection loop
    of ClothEnv._randval_minabs, a skip of the 301 062 words the domain-randomisation draws consume, and the exact draw
    sequence of a tier-1 and a tier-3 reset (cloth.pyx:75; cloth_env.py:851-877, :959-972)."""
    from gym_cloth_amd.envs import ClothVecEnv
    L = lib.load()
    vp = lambda a: a.ctypes.data_as(C.c_void_p)

    def draw(img, kind, n, a=0.0, b=0.0, c=0.0):
        out = np.zeros(max(n, 1))
        lib.check(L.clothhip_selftest_rng(vp(img), kind, n, a, b, c, lib.dp(out)))
        return out[:n]
    for seed in (0, 1337, 2 ** 31 - 5):
        ref = np.random.RandomState(seed)
        ref.rand(100)                                          # start from a mid-buffer position
        img = _mt_image(ref)
        want = ref.randint(0, 2 ** 32, size=2000, dtype=np.uint64).astype(np.float64)      # one 32-bit word each
        assert np.array_equal(draw(img, 0, 2000), want)
        assert np.array_equal(draw(img, 1, 700), ref.rand(700))
        assert np.array_equal(draw(img, 2, 300, -0.2, 0.2), np.array([ref.uniform(-0.2, 0.2) for _ in range(300)]))
        assert np.array_equal(draw(img, 3, 500, 625), np.array([ref.randint(625) for _ in range(500)], dtype=np.float64))
        assert np.array_equal(draw(img, 3, 50, 1024), np.array([ref.randint(1024) for _ in range(50)], dtype=np.float64))
        assert np.array_equal(draw(img, 4, 200, -0.2, 0.2, 0.08),
                              np.array([ClothVecEnv._randval_minabs(ref, -0.2, 0.2, 0.08) for _ in range(200)]))
        ref.uniform(40, 50); ref.uniform(0.7, 1.3)
        lim = ref.uniform(-15.0, 15.0)
        ref.uniform(-lim, lim, size=(224, 224, 3))
        draw(img, 5, 0, float(2 * (3 + 224 * 224 * 3)))
        assert np.array_equal(img, _mt_image(ref)), "state after the domain-randomisation skip"
        # a tier-1 reset with its third pull, then a tier-3 reset, drawn word for word like the host path draws them
        side = ref.rand() > 0.5
        assert (draw(img, 1, 1)[0] > 0.5) == side
        for _ in range(3):
            assert draw(img, 3, 1, 625)[0] == ref.randint(625)
            assert draw(img, 4, 1, -0.2, 0.2, 0.08)[0] == ClothVecEnv._randval_minabs(ref, -0.2, 0.2, 0.08)
            assert draw(img, 4, 1, -0.2, 0.2, 0.08)[0] == ClothVecEnv._randval_minabs(ref, -0.2, 0.2, 0.08)
        assert draw(img, 2, 1, 200.0, 280.0)[0] == ref.uniform(low=200, high=280)
        for lo, hi, m in ((0.30, 0.70, 0.0), (0.30, 0.70, 0.0), (-0.25, 0.25, 0.10), (-0.25, 0.25, 0.10)):
            assert draw(img, 4, 1, lo, hi, m)[0] == ClothVecEnv._randval_minabs(ref, lo, hi, m if m > 0 else None)
        assert np.array_equal(img, _mt_image(ref))
