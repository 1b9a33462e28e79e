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
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr.decode()[-3000:]
    lines = [l for l in out.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    assert len(lines[0]) <= 4096, len(lines[0])               # round 5's 22.5 KB line was not parsed by the driver
    return json.loads(lines[0])


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config", "roofline", "cpu_baseline", "value_definition")


def test_default_command_prints_one_short_parsable_line():
    """The command the driver runs -- `python bench.py --gpus 1 --steps 20 --warmup 5`, companions ON -- prints exactly one stdout line of at
    most 4 KB that json.loads accepts, with `roofline` and `cpu_baseline` on it and one-line summaries of the companions; the full records go
    to bench_extra.json."""
    side = os.path.join(ROOT, "bench_extra.json")
    if os.path.exists(side):
        os.remove(side)
    d = _bench("--gpus", "1", "--steps", "20", "--warmup", "5")
    for k in CONTRACT_KEYS:
        assert k in d, k
    assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1
    c, r, b = d["config"], d["roofline"], d["cpu_baseline"]
    for k in ("workload", "envs_per_gpu", "n_side", "init", "exact_order", "variant", "timed_region_s", "action_time_frac", "steps_equivalent",
              "blended_substeps_per_s", "env_steps_per_s", "rccl_nranks"):
        assert k in c, k
    assert c["envs_per_gpu"] == 512 and c["n_side"] == 25 and c["exact_order"] is True and "extra" not in c
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert b["kind"] == "port" and b["cores"] >= 1 and b["value"] > 0
    # value x the actions' share of the clock = the actions' substeps (6 significant digits on the line)
    n_act = c["action_substeps_per_env_step"] * c["env_steps_executed"]
    assert abs(d["value"] * c["timed_region_s"] * c["action_time_frac"] - n_act) <= 1e-4 * n_act
    assert 0 < c["timed_region_s"] < 10 and 15 <= c["steps_equivalent"] <= 25
    for k in ("f64", "relaxed_order", "step_mode", "tier2_512", "e1536", "e2048", "configs4_50x50"):
        assert k in d and "error" not in d[k] and d[k]["value"] > 0 and 0 < d[k]["frac"] < 1, (k, d.get(k))
    assert d["relaxed_order"]["exact_order"] is False and d["relaxed_order"]["parity"] == "none"
    assert d["extra_file"] == "bench_extra.json"
    full = json.load(open(side))
    assert full["line"] == d and len(full["extra"]) >= 8 and "slice_calibration" in full["headline"]["config"]


def test_bench_line_contract_small_batch():
    d = _bench("--envs", "64", "--steps", "4", "--warmup", "2", "--fuse", "2", "--no-extra")
    for k in CONTRACT_KEYS:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 2 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "cloth-substeps/s"
    c, r, b = d["config"], d["roofline"], d["cpu_baseline"]
    assert "workload" in c and "model" not in c and c["exact_order"] is True
    # value: the actions' substeps over the actions' share of the clock; blended: everything over the whole clock
    n_act = c["action_substeps_per_env_step"] * c["env_steps_executed"]
    # (the printed line carries six significant digits; bench_extra.json the full precision)
    assert abs(d["value"] * c["timed_region_s"] * c["action_time_frac"] - n_act) <= 1e-4 * n_act
    assert abs(c["blended_substeps_per_s"] * c["timed_region_s"] - c["substeps_per_env_step"] * c["env_steps_executed"]) <= 1e-4 * n_act
    assert 0.2 < c["action_time_frac"] < 1.0 and c["blended_substeps_per_s"] > d["value"] * c["action_time_frac"]
    assert abs(d["ms_per_step"] * c["steps_equivalent"] - 1e3 * c["timed_region_s"] * c["action_time_frac"]) <= 1e-1
    assert "action_substeps" in d["value_definition"] and "action_time_frac" in d["value_definition"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert abs(r["achieved"] - 1e-9 * r["alg_bytes_per_substep"] * r["action_substeps_per_launch"] / (1e-3 * r["kernel_ms_in_actions_avg"])) <= 1e-4 * r["achieved"]
    assert r["alg_bytes_per_substep"] == 49 * 625 and r["launches"] == 2 and r["kernel_ms_avg"] > 0
    assert r["dispatches_per_launch"] == 1                                   # (64 cloths: one generation of workgroups per launch)
    assert (r["traffic"] is None) == (r["traffic_source"] is None)        # (64 cloths: no committed PMC record -> null, and said so)
    assert b["kind"] == "port" and b["cores"] >= 1 and b["value"] > 0 and "sample" in b and b["unit"] == "cloth-substeps/s"
    assert "k_run_schedule<float,512,2,2,true,1,N25>" in c["variant"]        # the eight-wave LEAN build, specialised for the 25x25 grid


def test_relaxed_companion_is_labelled(monkeypatch):
    import bench
    rec = bench.run_workload(25, 64, "f32", "tier1", "fused", 4, 2, 2, 0, 1, 0, step_ms=40.0, relaxed=True)
    assert rec["config"]["exact_order"] is False and rec["config"]["parity"].startswith("none")
    assert "true,3>" in rec["config"]["variant"] and rec["value"] > 0
    # per handle (ABI 7): a handle created next to it in the same process steps in the reference's order
    rec2 = bench.run_workload(25, 64, "f32", "tier1", "fused", 4, 2, 2, 0, 1, 0, step_ms=40.0)
    assert rec2["config"]["exact_order"] is True and "true,1,N25>" in rec2["config"]["variant"]
