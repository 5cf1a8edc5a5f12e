#!/usr/bin/env python3
"""Does running two independent half-batches concurrently (two contexts = two HIP streams, one host thread each) overlap the
matrix-pipe-bound H|psi> kernel of one with the HBM-bound Gram / orthogonalisation kernels of the other?
Prints wall time for 2 x 32 sites back to back and for the same two calls issued from two threads."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.lattice import spread_sites
from rslmtoasa_amd.recursion import Recursion

def main(nhalf=32, lld=50, cells=22):
    p = supercell_problem((cells,) * 3)
    kk = p["nn"].shape[0]
    sites = spread_sites(kk, 2 * nhalf)
    recs = []
    for h in range(2):
        ham, lat, ctl, en = objects_from(p, sites[h * nhalf:(h + 1) * nhalf], lld, emin=-3.0, emax=1.8)
        recs.append(Recursion(ham, lat, ctl, en))
    for r in recs:
        r.recur_b(); r.recur_b()
    t0 = time.perf_counter()
    for r in recs:
        r.recur_b()
    t_seq = time.perf_counter() - t0
    best = None
    for rep in range(3):
        th = [threading.Thread(target=r.recur_b) for r in recs]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    print("2 x %d sites: back to back %.1f ms, two threads / two streams %.1f ms" % (nhalf, t_seq * 1e3, best * 1e3))
    for r in recs: r.close()

if __name__ == "__main__":
    main()
