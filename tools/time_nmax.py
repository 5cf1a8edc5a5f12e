#!/usr/bin/env python3
"""Cost of per-atom operator blocks (`hall`, the first nmax atoms; hamiltonian.f90:1618): the same periodic bcc cell with nmax = 0 ... 1000
atoms carrying their own (identical) copies of the stencil, with their groups formed per atom (s5_octet = 0) and over 8 chains (s5_octet = 1).
Block Lanczos, LL = 20.   tools/time_nmax.py [sites] [inside]   (inside: the recursion sites are the first atoms of the impurity region itself,
as in the reference's impurity runs, instead of being spread over the cell)"""
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
        nsites = int(sys.argv[1]) if len(sys.argv) > 1 else 16
        irec = np.arange(1, nsites + 1, dtype=np.int32) if len(sys.argv) > 2 else spread_sites(kk, nsites)
        ref = None
        for nmax in (0, 15, 32, 64, 200, 1000):
            p = dict(base, nmax=nmax)
            if nmax:
                p["hall"] = np.asfortranarray(np.repeat(base["ee"][:, :, :, :1], nmax, axis=3))
                if hoh:
                    p["hallo"] = np.asfortranarray(np.repeat(base["eeo"][:, :, :, :1], nmax, axis=3))
            rec = Recursion(*objects_from(p, irec, 20), device=0)
            t0 = time.perf_counter(); rec.update_hamiltonian(); t_set = time.perf_counter() - t0
            res = {}
            for octet in (0, 1):
                rec.set_option("s5_octet", octet)
                rec.recur_b()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter(); rec.recur_b(); ts.append(time.perf_counter() - t0)
                res[octet] = (1e3 * min(ts), int(rec.timing()["octet_launches"]))
            if ref is None:
                ref = rec.a_b.copy()
            dev = float(np.abs(rec.a_b - ref).max() / np.abs(ref).max())
            print("hoh=%d kk=%d sites=%d nmax=%-5d set_hamiltonian %.2f ms   recur_b %.1f ms per-atom groups, %.1f ms over chains (%d of the launches)   max deviation from nmax=0: %.1e"
                  % (hoh, kk, len(irec), nmax, 1e3 * t_set, res[0][0], res[1][0], res[1][1], dev), flush=True)
            rec.close()
