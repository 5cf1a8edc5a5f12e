#!/usr/bin/env python3
"""Timing of the non-default recursion variants at BASELINE config-1 size (22^3 atoms, 64 sites, LL=50):
Chebyshev, block+hoh, Chebyshev+hoh.  Prints device ms, H|psi> ms and nominal TFLOP/s."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from, supercell_problem, load_golden
from rslmtoasa_amd.lattice import spread_sites
from rslmtoasa_amd.recursion import Recursion

def run(label, hoh, kind, nsites=64, lld=50, cells=22):
    p = supercell_problem((cells,) * 3, hoh=hoh) if hoh else supercell_problem((cells,) * 3)
    kk = p["nn"].shape[0]
    sites = spread_sites(kk, nsites)
    ham, lat, ctl, en = objects_from(p, sites, lld, emin=-3.0, emax=1.8)
    rec = Recursion(ham, lat, ctl, en)
    for _ in range(2):
        (rec.chebyshev_recur if kind == "cheb" else rec.recur_b)()
    tm = rec.timing()
    nbm, ast = tm["block_multiplies"], tm["atom_steps"]
    extra = 2 if kind == "cheb" else 5
    flop = 46656.0 * (nbm + extra * ast)
    print("%-22s total %.1f ms  hop %.1f ms (%d launches)  -> %.1f TF/s whole, hop %.1f TF/s nominal" % (
        label, tm["total_ms"], tm["hop_ms"], tm["hop_launches"], flop / tm["total_ms"] * 1e-9, 46656.0 * nbm / tm["hop_ms"] * 1e-9))
    rec.close()

if __name__ == "__main__":
    run("block", False, "block")
    run("chebyshev", False, "cheb")
    try:
        run("block hoh", True, "block")
        run("chebyshev hoh", True, "cheb")
    except TypeError as e:
        print("hoh supercell helper not available:", e)
