#!/usr/bin/env python3
"""tools/phase_profile.py with the per-phase shader-cycle stamps of every launch (dev tool). Needs the profiling build:
  make -C gym_cloth_amd/csrc stamps
  CLOTHHIP_LIB=$PWD/gym_cloth_amd/libclothhip_stamps.so CLOTHHIP_DEBUG_PHASES=47 python tools/phase_profile_stamps.py"""
import os, sys
sys.path.insert(0, os.getcwd())
sys.argv = ["phase_profile.py"]
import numpy as np
import tools.phase_profile as pp
from gym_cloth_amd import ClothBatch
orig = ClothBatch.run
names = ["adjust", "hooke", "insert", "ranges", "fill", "precheck", "cells-wait", "reset+plane", "prepass", "sweep", "big cells", "small cells"]
def run(self, s):
    r = orig(self, s)
    st = self.debug_stats()[0].astype(float)
    n = max(int(s["n_total"][0]), 1)
    print("      cycles/substep: " + "  ".join("%s %.0f" % (nm, st[4 + i] * 64 / n) for i, nm in enumerate(names)))
    return r
ClothBatch.run = run
pp.main()
