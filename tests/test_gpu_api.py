"""GPU tests of the remaining C-ABI surface: async run + sync, device-resident schedules / observations,
determinism across envs and runs (race check), force_grab, partial reset of a vector env, error paths."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_env import base_cfg

pytestmark = pytest.mark.gpu


def _pull_state(oracle_lib):
    g = oracle_lib.load_golden("g_traj_lift_pull_25.npz")
    return g, (g["cp_pos"][9], g["cp_prev"][9], g["cp_pinned"][9], g["rest"])


@pytest.mark.parametrize("prec", ["f64", "f32"])
def test_deterministic_across_envs_and_runs(prec, oracle_lib):
    """Same state + same schedule -> bit-identical results in every env and in a second run (no data race, no
    dependence on workgroup placement or LDS-atomic ordering)."""
    from gym_cloth_amd import ClothBatch, make_schedules
    g, st = _pull_state(oracle_lib)
    outs = []
    for rep in range(2):
        b = ClothBatch(base_cfg("tier1", 1), n_envs=96, precision=prec)
        b.set_state(*st)
        s = make_schedules(96, active=1, break_on_tear=1, n_pull_end=120, n_griprest_end=180, n_total=400,
                           dx_pull=0.0012, dy_pull=0.0016)
        ex = b.run(s)
        assert (ex == 400).all()
        pos, prev, pin = b.get_state()
        assert all(np.array_equal(pos[0], pos[e]) and np.array_equal(prev[0], prev[e]) for e in range(96))
        outs.append((pos[0].copy(), prev[0].copy(), pin[0].copy()))
        b.close()
    assert all(np.array_equal(a, c) for a, c in zip(outs[0], outs[1]))


def test_async_run_and_device_schedule_and_obs(oracle_lib):
    """clothhip_run_async + clothhip_sync, clothhip_run_device_sched_async (schedule table already on the device,
    as after an RCCL broadcast) and clothhip_write_obs_f32_device give the same result as the synchronous path."""
    from gym_cloth_amd import ClothBatch, make_schedules
    from gym_cloth_amd._lib import SCHED_DTYPE
    from helpers import DeviceBuffer
    g, st = _pull_state(oracle_lib)
    E = 8
    s = make_schedules(E, active=1, break_on_tear=1, n_pull_end=40, n_griprest_end=60, n_total=100,
                       dx_pull=0.0012, dy_pull=0.0016)
    s["active"][3] = 0
    ref = ClothBatch(base_cfg("tier1", 1), n_envs=E, precision="f32"); ref.set_state(*st)
    ex_ref = ref.run(s); pos_ref = ref.positions()
    a = ClothBatch(base_cfg("tier1", 1), n_envs=E, precision="f32"); a.set_state(*st)
    a.run_async(s); ex_a = a.sync()
    assert np.array_equal(ex_a, ex_ref) and ex_ref[3] == 0 and np.array_equal(a.positions(), pos_ref)
    d = ClothBatch(base_cfg("tier1", 1), n_envs=E, precision="f32"); d.set_state(*st)
    assert s.dtype == SCHED_DTYPE and s.nbytes == 64 * E
    dev_s = DeviceBuffer(s.nbytes)                      # the schedule table as a collective would leave it: on the device
    dev_s.upload(np.frombuffer(s.tobytes(), dtype=np.uint8))
    d.run_device_sched_async(dev_s.ptr); ex_d = d.sync()
    assert np.array_equal(ex_d, ex_ref) and np.array_equal(d.positions(), pos_ref)
    obs = DeviceBuffer(E * 3 * 625 * 4)
    d.write_obs_f32_device(obs.ptr); d.sync(False)
    assert np.array_equal(obs.download(np.float32, (E, 625, 3)), pos_ref.astype(np.float32))
    dev_s.free(); obs.free()
    assert d.last_kernel_ms > 0
    for x in (ref, a, d):
        x.close()


def test_force_grab_and_no_grab_penalty(oracle_lib):
    """force_grab grows the radius by 0.02 until something is grabbed (cloth_env.py:434-444); without it an empty
    grab is a no-op action with the -0.01 penalty (cloth_env.py:490-493, :564-565)."""
    from gym_cloth_amd.envs import ClothVecEnv
    cfg = base_cfg("tier1", 5)
    act = np.array([[0.99, 0.99, -0.2, -0.2]])          # after the reset pulls there is usually no cloth under (0.995, 0.995)
    v = ClothVecEnv(cfg, n_envs=1, precision="f64"); v.seed(5); v.reset()
    pos = v.batch.positions()[0]
    d2 = (pos[:, 0] - 0.995) ** 2 + (pos[:, 1] - 0.995) ** 2
    if d2.min() < cfg["env"]["grip_radius"]:
        pytest.skip("cloth happens to lie under the corner for this seed")
    obs, rew, done, info = v.step(act)
    assert v.last_grabbed[0] == 0 and v.last_executed[0] == 0 and info["num_sim_steps"][0] == 0
    assert abs(rew[0] - (-0.01)) < 1e-12
    cfg["env"]["force_grab"] = True
    f = ClothVecEnv(cfg, n_envs=1, precision="f64"); f.seed(5); f.reset()
    assert np.array_equal(f.batch.positions()[0], pos)
    oc = oracle_lib.OracleCloth(oracle_lib.load_golden("g_traj_lift_pull_25.npz")["cfg"])
    p, q, pin = f.batch.get_state()
    oc.set_state(p[0], q[0], pin[0])
    r, n = cfg["env"]["grip_radius"], 0
    while n == 0:
        r += 0.02
        n = oc.grab_top(0.995, 0.995, r)
    f.step(act)
    assert f.last_grabbed[0] == n and f.last_executed[0] > 1000
    v.close(); f.close()


def test_partial_reset_leaves_other_envs_untouched():
    from gym_cloth_amd.envs import ClothVecEnv
    v = ClothVecEnv(base_cfg("tier1", 11), n_envs=4, precision="f32"); v.seed(11); v.reset()
    before = v.batch.get_state()
    mask = np.array([False, True, False, True])
    v.reset(mask)
    after = v.batch.get_state()
    for e in (0, 2):
        assert all(np.array_equal(before[k][e], after[k][e]) for k in range(3))
    for e in (1, 3):
        assert not np.array_equal(before[0][e], after[0][e]) and v.num_steps[e] == 0
    v.close()


def test_error_paths():
    from gym_cloth_amd import ClothBatch, make_schedules
    b = ClothBatch(base_cfg("tier1", 1), n_envs=2, precision="f32")
    with pytest.raises(ValueError):
        b.run(make_schedules(2, active=1, n_up_end=10, n_uprest_end=5, n_total=20))       # decreasing boundaries
    with pytest.raises(ValueError):
        b.run(make_schedules(3, active=1, n_total=1))                                       # wrong batch size
    with pytest.raises(ValueError):
        b.pin_points(0, [9999])
    with pytest.raises(ValueError):
        b.get_state(env0=1, n=5)
    b.close()
    with pytest.raises(ValueError):
        ClothBatch(base_cfg("tier1", 1), n_envs=0)
