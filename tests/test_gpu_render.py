"""GPU tests of the headless rasteriser (clothhip_render) against its numpy restatement
(oracle/render_oracle.py): bit-identical RGB and depth images on flat, lifted and folded (two-layer, self-occluding)
cloth states from the reference's trajectories, with a straight-down and a tilted camera; plus scene sanity."""
import numpy as np
import pytest

from test_gpu_parity import cfg_from_golden

pytestmark = pytest.mark.gpu


def _states(oracle_lib):
    g1 = oracle_lib.load_golden("g_traj_lift_pull_25.npz")
    g2 = oracle_lib.load_golden("g_traj_fold_25.npz")
    return g1, np.stack([g1["cp_pos"][0], g1["cp_pos"][6], g1["cp_pos"][10], g2["cp_pos"][3], g2["cp_pos"][5]])


@pytest.mark.parametrize("prec", ["f32", "f64"])
@pytest.mark.parametrize("cam_deg,size", [((0.0, 0.0, 0.0), (224, 224)), ((4.0, -3.0, 10.0), (96, 128))])
def test_render_matches_numpy_oracle(prec, cam_deg, size, oracle_lib):
    from gym_cloth_amd import ClothBatch
    from oracle import render_oracle
    g, states = _states(oracle_lib)
    n = len(states)
    b = ClothBatch(cfg_from_golden(g), n_envs=n, precision=prec)
    b.set_state(states, states, np.zeros((n, 625), dtype=np.uint8))
    W, H = size
    swap = np.array([0, 1, 0, 1, 0], dtype=np.uint8)
    rgb, dep = b.render(width=W, height=H, cam_deg=cam_deg, swap_sides=swap)
    pos = b.positions()                                  # what the device holds (fp32-rounded in f32 mode)
    d = dict(ClothBatch.RENDER_DEFAULTS)
    for e in range(n):
        orgb, odep = render_oracle.render(pos[e], 25, W, H, d["cam_pos"], ClothBatch.camera_matrix(cam_deg), d["lens_mm"],
                                          d["sensor_mm"], d["front"], d["back"], d["background"], d["light_dir"], d["ambient"],
                                          d["energy"], swap=bool(swap[e]))
        assert np.array_equal(dep[e], odep), (e, int((dep[e] != odep).sum()))
        assert np.array_equal(rgb[e], orgb), (e, int((rgb[e] != orgb).any(axis=-1).sum()))
    # scene sanity on the flat cloth seen straight down: the cloth covers the central (1 / 1.3)^2 of the frame in the front
    # colour, everything else is the white bed at the camera height
    if cam_deg == (0.0, 0.0, 0.0):
        cloth = (rgb[0] != 255).any(axis=-1)
        assert abs(cloth.mean() - (1.0 / 1.305) ** 2) < 0.01
        assert np.allclose(dep[0][~cloth], 1.45) and np.allclose(dep[0][cloth], 1.45, atol=1e-5)
        ys, xs = np.nonzero(cloth)
        assert abs(xs.mean() - (W - 1) / 2) < 1.0 and abs(ys.mean() - (H - 1) / 2) < 1.0
        px = rgb[0][H // 2, W // 2].astype(int)
        assert px[2] > 4 * px[0] and px[2] > 4 * px[1]          # the front colour (0.07, 0.05, 0.6) of get_image_rep_279.py:253, lit
        lifted = dep[2][cloth].min()
        assert lifted < 1.45 - 0.05                       # the lifted cloth is nearer to the camera
    b.close()


def test_image_obs_shapes_and_depth_normalisation():
    from gym_cloth_amd.envs import ClothVecEnv
    from test_gpu_env import base_cfg
    v = ClothVecEnv(base_cfg("tier1", 5), n_envs=3, precision="f32")
    v.seed(5); v.reset()
    rgb = v.image_obs()
    dep = v.image_obs(use_depth=True)
    rgbd = v.image_obs(rgbd=True, width=100, height=100)
    assert rgb.shape == (3, 224, 224, 3) and rgb.dtype == np.uint8 and dep.shape == (3, 224, 224, 3)
    assert rgbd.shape == (3, 100, 100, 4)
    assert dep.max() == 205 and dep.min() == 0            # normalised to 0..255, minus 50 (cloth_env.py:301-302)
    assert (dep[..., 0] == dep[..., 1]).all()
    v.close()
