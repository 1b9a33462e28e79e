for r in 1 2; do for lib in "$@"; do out=$(CLOTHHIP_LIB=$PWD/$lib python3 bench.py --no-extra --no-cpu-baseline --precision f64 --steps 10 --step-ms 125 2>&1 | tail -1); echo "$lib run $r: $(echo "$out" | python3 -c 'import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d["roofline"]
    print("value %.3f M/s  kernel-only %.3f M/s" % (d["value"]/1e6, r["substeps_per_launch"]/r["kernel_ms_avg"]/1e3))
except Exception as e:
    print("FAILED", e)')"; done; done
