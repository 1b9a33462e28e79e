#!/usr/bin/env python3
"""Dev tool: for a kernel's ISA (tools/devbuild.sh -> /tmp/k_NAME.s) list the largest loops (backward branches) with the scratch loads /
stores, barriers and instructions inside each -- is a spill reload inside the substep loop or in the cold episode code around it?"""
import re, sys
for path in sys.argv[1:]:
    L = open(path).read().splitlines()
    lab = {}
    for n, l in enumerate(L):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: lab[m.group(1)] = n
    loops = []
    for n, l in enumerate(L):
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            t = lab.get(m.group(1) or m.group(2))
            if t is not None and t < n: loops.append((n - t, t, n))
    loops.sort(reverse=True)
    print(path)
    for span, a, b in loops[:6]:
        body = L[a:b + 1]
        ins = [x for x in body if x.startswith("\t") and not x.startswith("\t.") and not x.startswith("\t;")]
        print("  loop lines %6d..%6d: %5d instr, scratch_load %2d, scratch_store %2d, s_barrier %2d, ds_ %4d, s_sleep %d" % (
            a, b, len(ins), sum("scratch_load" in x for x in body), sum("scratch_store" in x for x in body), sum("s_barrier" in x for x in body),
            sum("\tds_" in x for x in body), sum("s_sleep" in x for x in body)))
