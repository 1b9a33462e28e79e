#!/bin/bash
# GPU box: run the full default bench into gpurun_out/$1/bench.json and print a compact summary (dev tool).
out=gpurun_out/${1:-bench}; shift
mkdir -p $out
t0=$(date +%s)
python3 bench.py "$@" > $out/bench.json 2> $out/bench.err
echo "bench rc=$? wall=$(( $(date +%s) - t0 )) s"; tail -3 $out/bench.err
python3 - $out/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["config"]
print("HEAD value (actions only) %.3f M  blended %.3f M  ms/step %.2f  frac %.4f  steps_eq %.2f  variant %s" % (d["value"] / 1e6, c.get("blended_substeps_per_s", 0) / 1e6, d["ms_per_step"], d["roofline"]["frac"], c["steps_equivalent"], c.get("variant")))
print("  line bytes", len(open(sys.argv[1]).read().strip()), "traffic", d["roofline"]["traffic"])
print("  cpu", d.get("cpu_baseline"))
import os
xf = os.path.join(os.path.dirname(os.path.abspath(sys.argv[1])), "..", "..", d.get("extra_file", "bench_extra.json"))
ex = json.load(open(xf))["extra"] if d.get("extra_file") and os.path.exists(xf) else []
for r in ex:
    rc = r.get("config") or {}
    print("  %-62s %s  blended %s  steps_eq %s  wall %.1f s  %s %s" % (r.get("label", "")[:62], ("%.3f M" % (r["value"] / 1e6)) if "value" in r and r.get("unit", "cloth-substeps/s") == "cloth-substeps/s" else r.get("value"),
          ("%.3f M" % (rc["blended_substeps_per_s"] / 1e6)) if rc.get("blended_substeps_per_s") else "-", rc.get("steps_equivalent"), r.get("wall_s", 0), rc.get("variant", ""), r.get("error", "")))
PY
