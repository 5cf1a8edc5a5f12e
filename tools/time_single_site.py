#!/usr/bin/env python3
"""One site of the 22^3 cell, LL=50 (the SCF use case of a bulk calculation): where does the time go?"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.recursion import Recursion

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    c = int(os.environ.get("CELLS", "22"))
    lld = int(os.environ.get("LLD", "50"))
    p = supercell_problem((c, c, c))
    rec = Recursion(*objects_from(p, np.arange(1, n + 1, dtype=np.int32) * 97 % (c ** 3) + 1, lld))
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        rec.set_option(k, int(v))
    for it in range(5):
        t0 = time.time(); rec.recur_b(); w = time.time() - t0
        tm = rec.timing()
        print("call %d: wall %.1f ms  device %.1f ms  hop %.1f ms  host(region) %.1f ms" % (it, w * 1e3, tm["total_ms"], tm["hop_ms"], tm["host_ms"]))
    rec.close()
