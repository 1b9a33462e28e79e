#!/usr/bin/env python3
"""Turn a tools/collect_profiles.sh output directory (rocprofv3 rocpd databases + text logs) into the small
summaries kept under profiles/ (dev tool, runs anywhere: only needs sqlite3).

  python tools/summarize_profiles.py gpurun_out/prof_r1d profiles/r01_final
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
    disp = c.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, scratch_size from kernels "
                     "where name like '%k_run_schedule%' order by start").fetchall()
    out = {"command": "rocprofv3 --kernel-trace --stats -- python3 bench.py",
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
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        if not os.path.isdir(d):
            continue
        c = db_of(d)
        if c is None:
            continue
        for name, in c.execute("select distinct counter_name from counters_collection").fetchall():
            vals = [r[0] for r in c.execute("select sum(value) from counters_collection where counter_name=? and "
                                            "kernel_name like '%k_run_schedule%' group by dispatch_id order by dispatch_id", (name,))]
            pmc["counters"][name] = float(sum(vals))
            if name in ("FETCH_SIZE", "WRITE_SIZE"):
                pmc["per_dispatch"][name + "_KB"] = [round(v, 1) for v in vals]
    json.dump(pmc, open(dst + "_pmc_summary.json", "w"), indent=1)
    for nm, to in (("phase_f32.txt", "_phase_profile_f32.txt"), ("phase_f64.txt", "_phase_profile_f64.txt"),
                   ("bench_diag.txt", "_bench_diag.txt")):
        p = os.path.join(src, nm)
        if os.path.exists(p):
            shutil.copy(p, dst + to)
    print(open(dst + "_kernel_stats.csv").read())
    print(json.dumps(pmc, indent=1)[:1500])


if __name__ == "__main__":
    main()
