#!/usr/bin/env python3
"""A/B of a library option on a mid-size problem: bitwise comparison of the coefficients + timing.  usage: ab_option.py key=val [key=val ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.recursion import Recursion

if __name__ == "__main__":
    c = int(os.environ.get("CELLS", "12")); lld = int(os.environ.get("LLD", "20")); n = int(os.environ.get("SITES", "64"))
    hoh = os.environ.get("HOH", "0") == "1"
    p = supercell_problem((c, c, c), hoh=hoh)
    sites = (np.arange(1, n + 1, dtype=np.int64) * 97 % (c ** 3) + 1).astype(np.int32)
    rec = Recursion(*objects_from(p, sites, lld))
    rec.recur_b()
    a0, b0 = rec.a_b.copy(), rec.b2_b.copy()
    t0 = rec.timing()
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        rec.set_option(k, int(v))
    rec.recur_b(); rec.recur_b()
    t1 = rec.timing()
    da = np.abs(rec.a_b - a0).max() / np.abs(a0).max(); db = np.abs(rec.b2_b - b0).max() / np.abs(b0).max()
    print("%s: bitwise a_b %s b2_b %s (max rel diff %.2e %.2e); device ms %.2f -> %.2f (hop %.2f -> %.2f)" % (" ".join(sys.argv[1:]), np.array_equal(rec.a_b, a0), np.array_equal(rec.b2_b, b0), da, db, t0["total_ms"], t1["total_ms"], t0["hop_ms"], t1["hop_ms"]))
    rec.close()
