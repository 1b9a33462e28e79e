"""The driver's contract with bench.py, checked on a small batch: ONE JSON line with the keys the task names, `value` on SURVEY 8d's
definition (the actions' update() calls over the actions' share of the timed wall time), the roofline and CPU-baseline objects, and the
relaxed-order companion labelled as what it is."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK") and not k.startswith("CLOTHHIP_DEBUG")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_contract_small_batch():
    d = _bench("--envs", "64", "--steps", "4", "--warmup", "2", "--fuse", "2", "--no-extra")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "cloth-substeps/s"
    c, r, b = d["config"], d["roofline"], d["cpu_baseline"]
    assert "workload" in c and "model" not in c and c["exact_order"] is True
    # value: the actions' substeps over the actions' share of the clock; blended: everything over the whole clock
    n_act = c["action_substeps_per_env_step"] * c["env_steps_executed"]
    assert abs(d["value"] * c["timed_region_s"] * c["action_time_frac"] - n_act) <= 1e-6 * n_act
    assert abs(c["blended_substeps_per_s"] * c["timed_region_s"] - c["substeps_per_env_step"] * c["env_steps_executed"]) <= 1e-6 * n_act
    assert 0.2 < c["action_time_frac"] < 1.0 and c["blended_substeps_per_s"] > d["value"] * c["action_time_frac"]
    assert abs(d["ms_per_step"] * c["steps_equivalent"] - 1e3 * c["timed_region_s"] * c["action_time_frac"]) <= 1e-3
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - 1e-9 * r["alg_bytes_per_substep"] * r["action_substeps_per_launch"] / (1e-3 * r["kernel_ms_in_actions_avg"])) <= 1e-6 * r["achieved"]
    assert r["alg_bytes_per_substep"] == 49 * 625 and r["launches"] == 2 and r["kernel_ms_avg"] > 0
    assert r["dispatches_per_launch"] == 1                                   # (64 cloths: one generation of workgroups per launch)
    assert (r["traffic"] is None) == (r["traffic_source"] is None)        # (64 cloths: no committed PMC record -> null, and said so)
    assert b["kind"] == "port" and b["cores"] >= 1 and b["value"] > 0 and "sample" in b and b["unit"] == "cloth-substeps/s"
    assert "k_run_schedule<float,512,2,2,true,1>" in c["variant"]


def test_relaxed_companion_is_labelled(monkeypatch):
    import bench
    rec = bench.run_workload(25, 64, "f32", "tier1", "fused", 4, 2, 2, 0, 1, 0, step_ms=40.0, relaxed=True)
    assert rec["config"]["exact_order"] is False and rec["config"]["parity"].startswith("none")
    assert "true,3>" in rec["config"]["variant"] and rec["value"] > 0
    assert "CLOTHHIP_RELAXED_ORDER" not in os.environ
