#!/usr/bin/env python3
"""Per-phase timing of one real pick-and-place action (the reference's oracle action of the seed-1337 tier-1
episode) replicated over E cloths: lift / up-rest / pull / grip-rest / rest, each as its own launch (dev tool)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import load_golden  # noqa: E402
from tools.microbench import cfg_from_golden  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=512)
    ap.add_argument("--precision", default="f32")
    args = ap.parse_args()
    from gym_cloth_amd import ClothBatch, make_schedules
    g = load_golden("g_env_tier1_1337.npz")
    k = int(g["n_reset_calls"])
    a = g["act"][k]
    x, y = a[0] / 2 + 0.5, a[1] / 2 + 0.5
    L = np.sqrt(a[2] ** 2 + a[3] ** 2)
    xr, yr = a[2] / (L + 1e-5) * 0.002, a[3] / (L + 1e-5) * 0.002
    ip = int(g["act_n_updates"][k]) - 1430
    b = ClothBatch(cfg_from_golden(g), n_envs=args.envs, precision=args.precision)
    b.set_state(g["act_pos0"][k], g["act_prev0"][k], g["act_pin0"][k], g["rest"])
    print("grabbed", b.grab_top([x, y])[0], "iters_pull", ip)
    E = args.envs
    phases = [("lift", 50, dict(n_up_end=50, n_uprest_end=50, n_pull_end=50, n_griprest_end=50, n_total=50, dz_up=0.0025)),
              ("uprest", 80, dict(n_griprest_end=80, n_total=80)),
              ("pull", ip, dict(n_pull_end=ip, n_griprest_end=ip, n_total=ip, dx_pull=xr, dy_pull=yr)),
              ("griprest", 300, dict(n_griprest_end=300, n_total=300)),
              ("rest0-200", 200, dict(n_total=200)),
              ("rest200-1000", 800, dict(n_total=800))]
    tot_ms = 0.0
    for name, n, f in phases:
        s = make_schedules(E, active=1, break_on_tear=1, **f)
        b.run(s)
        ms = b.last_kernel_ms
        st = b.debug_stats()[0] / float(n)
        tot_ms += ms
        print("%-13s %4d substeps %8.2f ms  %7.2f us/substep | sweeps %.2f windows %5.1f passes %6.1f corrected %5.1f" %
              (name, n, ms, ms * 1e3 / n, st[0], st[1], st[2], st[3]))
        if os.environ.get("CLOTHHIP_DEBUG_PHASES") and int(os.environ["CLOTHHIP_DEBUG_PHASES"]) & 32:
            raw = b.debug_stats()[0].astype(float) * 64          # sweep-stamps build: passes of the window sweep (quiet / correcting)
            nq, nc = max(raw[4 + 9] / 64, 1), max(raw[3] / 64, 1)
            print("      window sweep: %.1f quiet passes/substep at %.0f cycles, %.1f correcting at %.0f" %
                  (nq / n, raw[4 + 10] / nq, nc / n, raw[4 + 11] / nc))
        if os.environ.get("CLOTH_WINDOW_STAMPS"):      # dev build -DCLOTHHIP_WINDOW_STAMPS (with the sweep stamps): the LDS latency the sweep sees
            raw = b.debug_stats()[0].astype(float) * 64
            nwin = max(raw[4 + 2] / 64, 1)
            print("      read-ahead of a window (two 16-byte LDS reads, waited for at once): %.0f cycles incl. one stamp; two stamps back "
                  "to back: %.0f cycles  (%.1f windows/substep)" % (raw[4 + 1] / nwin, raw[4 + 3] / nwin, nwin / n))
    nsub = sum(p[1] for p in phases)
    print("total %d substeps %.2f ms -> %.2f us/substep -> %.2f M substeps/s at E=%d" %
          (nsub, tot_ms, tot_ms * 1e3 / nsub, E * nsub / tot_ms / 1e3, E))
    if args.precision == "f64":
        print("final max|pos - reference| =", np.abs(b.positions()[0] - g["act_pos1"][k]).max())


if __name__ == "__main__":
    main()
