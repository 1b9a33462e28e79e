"""GPU tests of the gripper kernel (k_grab: Gripper.grab_top / Gripper.grab, gripper.pyx:23-53) against the reference's
index sets on flat, lifted and FOLDED states (two layers under the gripper), in both precisions."""
import numpy as np
import pytest

from test_gpu_parity import cfg_from_golden

pytestmark = pytest.mark.gpu


def _margins(pos, xy, radius, levels, tt):
    """distance of every point from the two decision boundaries of the grab tests (for the fp32 comparison)."""
    d2 = (pos[:, 0] - xy[0]) ** 2 + (pos[:, 1] - xy[1]) ** 2
    m_r = np.abs(d2 - radius)
    m_z = np.min(np.abs(np.abs(pos[:, 2][:, None] - levels[None, :]) - tt), axis=1)
    return m_r, m_z


@pytest.mark.parametrize("prec", ["f64", "f32"])
@pytest.mark.parametrize("top", [True, False])
def test_grab_sets_match_reference(prec, top, oracle_lib):
    from gym_cloth_amd import ClothBatch
    g = oracle_lib.load_golden("g_gripper_25.npz")
    n = len(g["xy"])
    b = ClothBatch(cfg_from_golden(g), n_envs=n, precision=prec)
    b.set_state(g["pos"], g["pos"], np.zeros((n, 625), dtype=np.uint8))
    cnt = (b.grab_top if top else b.grab)(g["xy"])
    pin = b.get_state()[2].astype(bool)
    want = g["grab_top"] if top else g["grab"]
    assert any(len(w) > 5 for w in want), "the fixture must contain a multi-layer grab"
    tt, radius = 2 * g["cfg"]["thickness"], g["cfg"]["grip_radius"]
    for q in range(n):
        got = set(np.nonzero(pin[q])[0].tolist())
        ref = set(want[q])
        if prec == "f64":
            assert got == ref and cnt[q] == len(ref), (q, sorted(got ^ ref))
        else:
            # fp32 positions: a point may flip only if it sits within fp32 rounding of a decision boundary
            m_r, m_z = _margins(g["pos"][q], g["xy"][q], radius, g["levels"], tt)
            for i in got ^ ref:
                assert m_r[i] < 1e-6 or m_z[i] < 1e-6, (q, i, m_r[i], m_z[i])
    # grabbing again appends the same points a second time (multiplicity): release clears both
    cnt2 = (b.grab_top if top else b.grab)(g["xy"])
    assert np.array_equal(cnt2 > 0, cnt > 0)
    b.release()
    assert not b.get_state()[2].any()
    b.close()
