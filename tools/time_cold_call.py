#!/usr/bin/env python3
"""First (cold) recursion call on a large cell: host time of the region search and order lists against the device time.  tools/time_cold_call.py [cells] [sites]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.lattice import spread_sites, supercell_positions
from rslmtoasa_amd.recursion import Recursion

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 46
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    t0 = time.perf_counter(); p = supercell_problem((n, n, n)); kk = p["nn"].shape[0]; t_lat = time.perf_counter() - t0
    ham, lat, ctl, en = objects_from(p, spread_sites(kk, ns), 50)
    lat.cr = supercell_positions((n, n, n))
    t0 = time.perf_counter(); rec = Recursion(ham, lat, ctl, en, device=0); t_new = time.perf_counter() - t0
    for rep in ("cold", "warm", "warm"):
        t0 = time.perf_counter(); rec.recur_b(); w = time.perf_counter() - t0
        tm = rec.timing()
        print("%d atoms, %d sites, LL=50, %s call: wall %.0f ms, device %.0f ms, host (region search, order lists, transfers) %.0f ms" % (kk, ns, rep, 1e3 * w, tm["total_ms"], tm["host_ms"]), flush=True)
    print("(python lattice tables %.1f s, engine set-up incl. set_lattice %.2f s)" % (t_lat, t_new))
    rec.close()
