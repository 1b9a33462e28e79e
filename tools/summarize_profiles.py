#!/usr/bin/env python3
"""Turn a tools/collect_profiles.sh output directory (rocprofv3 rocpd databases + text logs) into the small
summaries kept under profiles/ (dev tool, runs anywhere: only needs sqlite3).

  python tools/summarize_profiles.py gpurun_out/prof_r2 profiles/r02
Also writes profiles/r02_traffic.json: the per-launch HBM bytes bench.py reports as roofline.traffic.
"""
import csv
import glob
import json
import os
import shutil
import sqlite3
import sys


def db_of(d):
    f = sorted(glob.glob(os.path.join(d, "*.db")))
    return sqlite3.connect(f[0]) if f else None


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
    # kernel trace: per-kernel statistics (what `rocprofv3 --stats` prints) + every stepper dispatch
    c = db_of(os.path.join(src, "trace"))
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                     "group by name order by sum(duration) desc").fetchall()
    tot = float(sum(r[2] for r in rows))
    with open(dst + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), "%.1f" % r[3], "%.4f" % (100.0 * r[2] / tot), int(r[4]), int(r[5])])
        # the stepper's dispatches of a bench run are: one short calibration launch, the warm-up launch(es), the TIMED launches (the
        # last `roofline.launches` of them): a row for the timed ones alone, which is what bench.py's kernel_ms_avg averages
        try:
            tb_ = json.loads([l for l in open(os.path.join(src, "bench_traced.json")).read().splitlines() if l.startswith("{")][-1])
            nl_ = int(tb_["roofline"]["launches"])
            dd_ = [r[0] for r in c.execute("select duration from kernels where name like '%k_run_schedule%' order by start").fetchall()][-nl_:]
            w.writerow(["k_run_schedule: the %d TIMED dispatches only (bench.py roofline.kernel_ms_avg = %.3f ms)" % (nl_, tb_["roofline"]["kernel_ms_avg"]),
                        len(dd_), int(sum(dd_)), "%.1f" % (sum(dd_) / len(dd_)), "", int(min(dd_)), int(max(dd_))])
        except (OSError, ValueError, KeyError, IndexError):
            pass
    disp = c.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, scratch_size from kernels "
                     "where name like '%k_run_schedule%' order by start").fetchall()
    out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-extra --no-cpu-baseline",
           "note": "vgprs below is rocprofv3's per-dispatch field (arch VGPR granules as the tool reports them); the compiler's "
                   "count per kernel variant is in *_kernel_resources.txt (tools/kernel_resources.sh)",
           "k_run_schedule_dispatches_ms": [round(d[1] / 1e6, 3) for d in disp],
           "k_run_schedule_launch": ({"grid_threads": disp[0][2], "workgroup": disp[0][3], "lds_bytes": disp[0][4],
                                      "vgprs": disp[0][5], "scratch_bytes": disp[0][6]} if disp else None)}
    for nm in ("bench.json", "bench_traced.json"):
        p = os.path.join(src, nm)
        if os.path.exists(p):
            txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if txt:
                out[nm[:-5]] = json.loads(txt[-1])
    json.dump(out, open(dst + "_bench.json", "w"), indent=1)
    # PMC passes: sum per counter over the stepper dispatches, plus per-dispatch HBM traffic
    pmc = {"note": "rocprofv3 --pmc, one run per counter group (never combined with tracing); sums over the "
                   "k_run_schedule dispatches of the profiled command", "counters": {}, "per_dispatch": {}}

    def per_dispatch(d):
        c_ = db_of(d)
        res = {}
        if c_ is None:
            return res
        for name, in c_.execute("select distinct counter_name from counters_collection").fetchall():
            res[name] = [r[0] for r in c_.execute("select sum(value) from counters_collection where counter_name=? and "
                                                  "kernel_name like '%k_run_schedule%' group by dispatch_id order by dispatch_id", (name,))]
        return res
    for d in sorted(glob.glob(os.path.join(src, "pmc_*")) + glob.glob(os.path.join(src, "sq_*"))):
        if not os.path.isdir(d):
            continue
        for name, vals in per_dispatch(d).items():
            pmc["counters"][name] = float(sum(vals))
            if name in ("FETCH_SIZE", "WRITE_SIZE"):
                pmc["per_dispatch"][name + "_KB"] = [round(v, 1) for v in vals]
            if name == "TCC_EA0_RDREQ_sum":
                pmc["per_dispatch"]["TCC_EA0_RDREQ_x64B_KB"] = [round(v * 64.0 / 1024.0, 1) for v in vals]
    # repeated passes: is the burst one dispatch shows now and then real? (it moves between dispatches and counters from pass to pass)
    for cname in ("FETCH_SIZE", "WRITE_SIZE"):
        for d in sorted(glob.glob(os.path.join(src, "pmcrep*_" + cname))):
            v = per_dispatch(d).get(cname)
            if v:
                pmc["per_dispatch"].setdefault(cname + "_KB_repeats", []).append([round(x, 1) for x in v])
    for d in sorted(glob.glob(os.path.join(src, "pmcx*_WRITE_SIZE_TCC_EA0_WRREQ_sum"))):      # both counters in ONE pass
        r = per_dispatch(d)
        if r.get("WRITE_SIZE") and r.get("TCC_EA0_WRREQ_sum"):
            pmc["per_dispatch"].setdefault("same_pass_WRITE_SIZE_KB_and_WRREQ", []).append(
                {"WRITE_SIZE_KB": [round(x, 1) for x in r["WRITE_SIZE"]], "TCC_EA0_WRREQ_sum": [round(x, 1) for x in r["TCC_EA0_WRREQ_sum"]]})
    json.dump(pmc, open(dst + "_pmc_summary.json", "w"), indent=1)
    # per-launch HBM traffic of the dominant kernel for bench.py's roofline.traffic: the timed launches are the last ones of
    # the profiled command; FETCH_SIZE is doubled (gfx950: it tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md).
    # One record per configuration bench.py reports a roofline for (keyed on mode, envs, grid, precision, init).
    records = []

    def traffic_record(prefix, out_json, command):
        p_ = os.path.join(src, out_json)
        if not os.path.exists(p_):
            return
        txt = [l for l in open(p_).read().splitlines() if l.startswith("{")]
        if not txt:
            return
        tb = json.loads(txt[-1])
        nl = int(tb.get("roofline", {}).get("launches", 0) or 0)
        g_ = int(tb.get("roofline", {}).get("dispatches_per_launch", 1) or 1)      # one kernel dispatch per generation of a time-sliced launch

        def grouped(v):
            # per LAUNCH: the launch's consecutive dispatches summed (counted from the end: calibration / warm-up launches come first)
            if not v or g_ <= 1:
                return v
            v = list(v)[len(v) % g_:]
            return [sum(v[i:i + g_]) for i in range(0, len(v), g_)]
        pd_ = lambda d: {k_: grouped(v_) for k_, v_ in per_dispatch(d).items()}      # noqa: E731
        fk = pd_(os.path.join(src, prefix + "FETCH_SIZE")).get("FETCH_SIZE")
        wk = pd_(os.path.join(src, prefix + "WRITE_SIZE")).get("WRITE_SIZE")
        if not (nl and fk and wk and len(fk) >= nl and len(wk) >= nl):
            return
        # the timed launches of every pass of the counter, pooled; the per-launch figure is their (lower) median and a dispatch
        # at more than twice the median is listed as an outlier (one dispatch of some passes shows a 10-14x burst in ONE counter,
        # a different dispatch and counter each time, with unchanged duration: profiles/README.md)
        def pooled(cname, first):
            vals = list(first[-nl:])
            if prefix == "pmc_":
                for d in sorted(glob.glob(os.path.join(src, "pmcrep*_" + cname)) + glob.glob(os.path.join(src, "pmcx*_" + cname + "_*"))):
                    v = pd_(d).get(cname)
                    if v and len(v) >= nl:
                        vals += v[-nl:]
            med = sorted(vals)[(len(vals) - 1) // 2]
            return med * 1024.0, [round(x, 1) for x in vals], [round(x, 1) for x in vals if x > 2.0 * med]
        fetch, f_all, f_out = pooled("FETCH_SIZE", fk)
        write, w_all, w_out = pooled("WRITE_SIZE", wk)
        cfg = tb["config"]
        records.append({"mode": "fused" if str(cfg["mode"]).startswith("fused") else "step", "envs": cfg["envs_per_gpu"], "n_side": cfg["n_side"],
                        "precision": tb["dtype"], "init": cfg["init"], "slice_ms": cfg.get("slice_ms"),
                        "fetch_size_bytes_per_launch_raw": fetch, "fetch_size_kb_of_the_timed_launches": f_all, "fetch_outliers_kb": f_out,
                        "write_size_bytes_per_launch": write, "write_size_kb_of_the_timed_launches": w_all, "write_outliers_kb": w_out,
                        "estimator": "lower median over the timed launches of all passes of the counter", "hbm_bytes_per_launch": 2.0 * fetch + write, "command": command,
                        "substeps_per_launch": tb["roofline"]["substeps_per_launch"],
                        "hbm_bytes_per_substep": (2.0 * fetch + write) / tb["roofline"]["substeps_per_launch"],
                        "variant": cfg.get("variant"),
                        "algorithmic_bytes_per_launch": tb["roofline"]["substeps_per_launch"] * tb["roofline"]["alg_bytes_per_substep"]})
    traffic_record("pmc_", "pmc_FETCH_SIZE.out", "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) -- python3 bench.py --no-extra --no-cpu-baseline")
    traffic_record("t2pmc_", "t2pmc_FETCH_SIZE.out", "... -- python3 bench.py --no-extra --no-cpu-baseline --init tier2 --steps 10")
    traffic_record("n50pmc_", "n50pmc_FETCH_SIZE.out", "... -- python3 bench.py --no-extra --no-cpu-baseline --n-side 50 --envs 1024 --steps 5 --warmup 0 --fuse 5 --step-ms 250")
    if records:
        json.dump({"records": records}, open(os.path.join(os.path.dirname(dst) or ".", os.path.basename(dst).split("_")[0] + "_traffic.json"), "w"), indent=1)
    # L2 hit / miss pass
    l2 = per_dispatch(os.path.join(src, "pmcl2_FETCH_SIZE_TCC"))
    if l2:
        pmc["per_dispatch"]["same_pass_FETCH_SIZE_KB_TCC_HIT_TCC_MISS"] = {k_: [round(x, 1) for x in v_] for k_, v_ in l2.items()}
        json.dump(pmc, open(dst + "_pmc_summary.json", "w"), indent=1)
    for nm, to in (("census.txt", "_census.txt"), ("phase_f32.txt", "_phase_profile_f32.txt"), ("phase_f64.txt", "_phase_profile_f64.txt"),
                   ("fused_balance.txt", "_fused_balance.txt"), ("fused_phases.txt", "_fused_phases.txt"),
                   ("fused_phases_f64.txt", "_fused_phases_f64.txt"), ("sweepstamps_f32.txt", "_sweepstamps_f32.txt"),
                   ("ablation.txt", "_phase_ablation.txt")):
        p = os.path.join(src, nm)
        if os.path.exists(p):
            shutil.copy(p, dst + to)
    for tag in ("1536", "2048"):
        cl = db_of(os.path.join(src, "trace" + tag))
        if cl is None:
            continue
        with open(dst + "_%s_cloths_dispatches.csv" % tag, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "DurationMs", "Workgroups", "WorkgroupSize", "LdsBytes", "ScratchBytes"])
            for r in cl.execute("select name, duration, grid_x, workgroup_x, lds_size, scratch_size from kernels where name like '%k_run_schedule%' order by start").fetchall():
                w.writerow([r[0], "%.3f" % (r[1] / 1e6), int(r[2] // max(r[3], 1)), r[3], r[4], r[5]])
            try:
                tb_ = json.loads([l for l in open(os.path.join(src, "bench%s_traced.json" % tag)).read().splitlines() if l.startswith("{")][-1])
                w.writerow(["bench.py of the same run: launches %d x dispatches_per_launch %d, roofline.kernel_ms_avg %.3f ms (spans a launch's dispatches), blended %.3f M substeps/s" %
                            (tb_["roofline"]["launches"], tb_["roofline"]["dispatches_per_launch"], tb_["roofline"]["kernel_ms_avg"], tb_["config"]["blended_substeps_per_s"] / 1e6)])
            except (OSError, ValueError, KeyError, IndexError):
                pass
    c50 = db_of(os.path.join(src, "trace50"))
    if c50 is not None:
        rows = c50.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                           "group by name order by sum(duration) desc").fetchall()
        tot = float(sum(r[2] for r in rows))
        with open(dst + "_50x50_kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r[0], r[1], int(r[2]), "%.1f" % r[3], "%.4f" % (100.0 * r[2] / tot), int(r[4]), int(r[5])])
    print(open(dst + "_kernel_stats.csv").read())
    print(json.dumps(pmc, indent=1)[:1500])


if __name__ == "__main__":
    main()
