#!/usr/bin/env python3
"""Cost of per-atom operator blocks (`hall`, the first nmax atoms; hamiltonian.f90:1618): the same periodic bcc cell with nmax = 0, 15, 200, 1000
atoms carrying their own (identical) copies of the stencil.  Block Lanczos, 16 sites, LL = 20."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem
from rslmtoasa_amd.lattice import spread_sites
from rslmtoasa_amd.recursion import Recursion

if __name__ == "__main__":
    dims = (16, 16, 16)
    for hoh in (False, True):
        base = supercell_problem(dims, hoh=hoh)
        kk = base["nn"].shape[0]
        irec = spread_sites(kk, 16)
        ref = None
        for nmax in (0, 15, 200, 1000):
            p = dict(base, nmax=nmax)
            if nmax:
                p["hall"] = np.asfortranarray(np.repeat(base["ee"][:, :, :, :1], nmax, axis=3))
                if hoh:
                    p["hallo"] = np.asfortranarray(np.repeat(base["eeo"][:, :, :, :1], nmax, axis=3))
            rec = Recursion(*objects_from(p, irec, 20), device=0)
            t0 = time.perf_counter(); rec.update_hamiltonian(); t_set = time.perf_counter() - t0
            rec.recur_b()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); rec.recur_b(); ts.append(time.perf_counter() - t0)
            tm = rec.timing()
            if ref is None:
                ref = rec.a_b.copy()
            dev = float(np.abs(rec.a_b - ref).max() / np.abs(ref).max())
            print("hoh=%d kk=%d nmax=%-5d set_hamiltonian %.2f ms   recur_b %.1f ms (H|psi> %.1f ms in %d launches)   max deviation from nmax=0: %.1e"
                  % (hoh, kk, nmax, 1e3 * t_set, 1e3 * min(ts), tm["hop_ms"], tm["hop_launches"], dev), flush=True)
            rec.close()
