#!/usr/bin/env python3
"""Kernel micro-benchmark / phase ablation on harvested reference states (dev tool, GPU only).
Mask bit 32 (e.g. --masks 47) prints per-phase shader-cycle stamps; that needs the profiling build of the library.

  python tools/microbench.py [--envs 512] [--precision f32] [--sub 200]
Regimes: 'rest' (settled cloth, nothing pinned), 'pull' (mid lateral pull, strain limiter busy),
'fold' (layers stacked, self-collision busy). Phase masks: bit0 hooke, bit1 collide, bit2 plane,
bit3 strain, bit4 disable the strain-sweep skipping.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import load_golden  # noqa: E402  (dev tool: uses the golden fixtures only)


def cfg_from_golden(g):
    c = g["cfg"]
    return {"cloth": {"num_width_points": c["n_side"], "num_height_points": c["n_side"], "width": c["width"],
                      "height": c["height"], "density": c["density"], "ks": c["ks"], "damping": c["damping"],
                      "thickness": c["thickness"], "plane_friction": c["plane_friction"],
                      "tear_thresh": c["tear_thresh"]},
            "frames_per_sec": c["frames_per_sec"], "simulation_steps": c["simulation_steps"],
            "env": {"grip_radius": c["grip_radius"]}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=512)
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--sub", type=int, default=200)
    ap.add_argument("--masks", default="15,31,1,3,7,9")
    ap.add_argument("--n50", action="store_true")
    args = ap.parse_args()
    from gym_cloth_amd import ClothBatch
    if args.n50:
        g = load_golden("g_traj_fold_50.npz")
        regimes = {"fold50-pull": (2, [0.0014142, 0.0014142, 0.0]), "fold50-rest": (6, None)}
    else:
        g = load_golden("g_traj_lift_pull_25.npz")
        regimes = {"settled": ("env", None), "rest": (14, None), "pull": (9, [0.0012, 0.0016, 0.0]),
                   "lift": (4, [0.0, 0.0, 0.0025])}
        gf = load_golden("g_traj_fold_25.npz")
        ge = load_golden("g_env_tier1_1337.npz")
    for name, (cp, delta) in regimes.items():
        for mask in [int(m) for m in args.masks.split(",")]:
            os.environ["CLOTHHIP_DEBUG_PHASES"] = str(mask)
            b = ClothBatch(cfg_from_golden(g), n_envs=args.envs, precision=args.precision)
            if cp == "env":
                st = (ge["post_pos"], ge["post_prev"], ge["post_pinned"], ge["rest"])
            else:
                st = (g["cp_pos"][cp], g["cp_prev"][cp], g["cp_pinned"][cp], g["rest"])
            b.set_state(*st)
            b.update(20, delta=delta)
            b.set_state(*st)
            b.update(args.sub, delta=delta)
            ms = b.last_kernel_ms
            st = b.debug_stats()[0] / float(args.sub)
            if mask & 32:       # needs the profiling build: make -C gym_cloth_amd/csrc stamps; CLOTHHIP_LIB=.../libclothhip_stamps.so
                names = ["adjust", "hooke", "insert", "ranges", "fill", "precheck", "cells: wait at barrier", "reset+plane", "prepass"]
                print("    cycles/substep (wave 0): " + "  ".join("%s %.0f" % (n, st[4 + i] * 64) for i, n in enumerate(names)))
                if os.environ.get("CLOTHHIP_SWEEP_STAMPS_LIB"):     # library built with -DCLOTHHIP_SWEEP_STAMPS instead of CELL_STAMPS
                    print("    window sweep (wave 0): passes without a correction %.0f  passes with one %.0f cycles/substep" %
                          (st[4 + 10] * 64, st[4 + 11] * 64))
                else:
                    print("    cells sweep (wave 0): cells over 16 members %.0f  small cells (four per pass) %.0f cycles/substep" %
                          (st[4 + 10] * 64, st[4 + 11] * 64))
            print("%-12s mask %2d: %8.2f us/substep  (%6.2f M substeps/s at E=%d)  per substep: sweeps %.2f windows %.1f passes %.1f corrected %.1f" %
                  (name, mask, ms * 1e3 / args.sub, args.envs * args.sub / ms / 1e3, args.envs, st[0], st[1], st[2], st[3]), flush=True)
            b.close()
    if not args.n50:
        os.environ["CLOTHHIP_DEBUG_PHASES"] = "15"
        b = ClothBatch(cfg_from_golden(gf), n_envs=args.envs, precision=args.precision)
        b.set_state(gf["cp_pos"][6], gf["cp_prev"][6], gf["cp_pinned"][6], gf["rest"])
        b.update(args.sub)
        ms = b.last_kernel_ms
        print("%-12s mask 15: %8.2f us/substep  (%6.2f M substeps/s)" % ("fold-rest", ms * 1e3 / args.sub,
                                                                        args.envs * args.sub / ms / 1e3))


if __name__ == "__main__":
    main()
