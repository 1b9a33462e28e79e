"""Two large-grid cloths per CU (50x50: 512 threads x 5 particles each, 79.7 KB of LDS, hash table of 2880 slots -- not a power of two --,
in-kernel metrics with the hull stack as u16 indices): the same episodes, bit for bit, as the one-per-CU build."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(large2, monkeypatch, E=12, T=2):
    import bench
    from gym_cloth_amd.envs import ClothVecEnv
    monkeypatch.delenv("CLOTHHIP_DEBUG_LEAN", raising=False)
    monkeypatch.setenv("CLOTHHIP_DEBUG_LARGE2", large2)
    env = ClothVecEnv(bench.bench_cfg(50, 0.0095), n_envs=E, precision="f32", consume_domrand_draws=False)
    for e in range(E):
        env.np_randoms[e] = np.random.RandomState(1000 + e)
    env.reset()
    acts = np.ascontiguousarray(np.stack([np.random.RandomState(2000 + e).uniform(-1, 1, size=(T, 4)) for e in range(E)], axis=1))
    out = env.step_many(acts, auto_reset=True)
    var = env.batch.last_variant()
    res = (out["rew"].copy(), out["executed"].copy(), out["actual_coverage"].copy(), out["obs"].copy(), [x.copy() for x in env.batch.get_state()])
    env.close()
    return res, var


def test_two_large_cloths_per_cu_equal_the_one_per_cu_build(monkeypatch):
    a, va = _run("0", monkeypatch)
    b, vb = _run("1", monkeypatch)
    assert va["lean"] and va["threads"] == 1024 and va["cloths_per_cu"] == 1, va
    assert vb["lean"] and vb["threads"] == 512 and vb["table_mode"] == 4 and vb["cloths_per_cu"] == 2, vb
    assert a[1].sum() > 10000
    for x, y in zip(a[:4], b[:4]):
        assert np.array_equal(x, y, equal_nan=True)
    for x, y in zip(a[4], b[4]):
        assert np.array_equal(x, y)
