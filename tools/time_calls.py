#!/usr/bin/env python3
"""Per-call wall / device / host times of consecutive recur_b calls on one handle (cold call, cached regions): tools/time_calls.py [cells] [hoh]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.lattice import spread_sites
from rslmtoasa_amd.recursion import Recursion
n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
hoh = len(sys.argv) > 2
p = supercell_problem((n, n, n), hoh=hoh)
rec = Recursion(*objects_from(p, spread_sites(p["nn"].shape[0], 64), 50), device=0)
rec.set_option("graph", 0)
for i in range(5):
    t0 = time.perf_counter(); rec.recur_b(); w = time.perf_counter() - t0
    tm = rec.timing()
    print("call %d: wall %.1f ms  device %.1f ms  host %.2f ms" % (i, w * 1e3, tm["total_ms"], tm["host_ms"]), flush=True)
rec.close()
