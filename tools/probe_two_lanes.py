#!/usr/bin/env python3
"""Experiment: two half batches (2 x 32 sites) advanced concurrently by two engine handles on one GPU, against one handle with 64 sites.
The MFMA-bound SpMM of one half can share the CUs with the HBM-bound post-hop passes of the other when both fit a CU together
(s5_waves=4: one SpMM wave per SIMD; orth3=2: 256-register orthogonalisation waves).  usage: probe_two_lanes.py [key=val ...]"""
import os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.lattice import spread_sites
from rslmtoasa_amd.recursion import Recursion

if __name__ == "__main__":
    c = int(os.environ.get("CELLS", "22")); lld = int(os.environ.get("LLD", "50")); n = int(os.environ.get("SITES", "64")); lanes = int(os.environ.get("LANES", "2"))
    p = supercell_problem((c, c, c))
    sites = spread_sites(c ** 3, n)
    recs = [Recursion(*objects_from(p, sites[i::lanes], lld)) for i in range(lanes)]
    for r in recs:
        for kv in sys.argv[1:]:
            k, v = kv.split("=")
            r.set_option(k, int(v))
    def run(r):
        r.recur_b()
    for it in range(4):
        t0 = time.time()
        th = [threading.Thread(target=run, args=(r,)) for r in recs]
        for t in th: t.start()
        for t in th: t.join()
        w = time.time() - t0
        print("lanes=%d %s: call %d wall %.1f ms (%s device ms)" % (lanes, " ".join(sys.argv[1:]), it, w * 1e3, ", ".join("%.1f" % r.timing()["total_ms"] for r in recs)))
    for r in recs: r.close()
