#!/usr/bin/env python3
"""Timing of the Green-function stage at BASELINE config-1 size: 64 sites x 2510 energies x LL=50 (rsrec_block_green)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import GOLD, objects_from, supercell_problem
from rslmtoasa_amd.green import Green
from rslmtoasa_amd.lattice import spread_sites
from rslmtoasa_amd.recursion import Recursion

if __name__ == "__main__":
    nsites, lld, nen = 64, 50, 2510
    p = supercell_problem((22, 22, 22))
    rec = Recursion(*objects_from(p, spread_sites(p["nn"].shape[0], nsites), lld))
    rec.recur_b()
    t_rec = rec.timing()["total_ms"]
    rec.zsqr()
    with np.load(os.path.join(GOLD, "bccFe_nsp2_block_green.npz")) as z:
        a_inf = np.repeat(z["a_inf"][:, :, :1], nsites, axis=2); b_inf = np.repeat(z["b_inf"][:, :, :1], nsites, axis=2)
        ene = float(z["ene_full_first"]) + float(z["ene_full_step"]) * np.arange(nen)
    gr = Green(rec, ene)
    for _ in range(2):
        t0 = time.time(); gr.block_green(a_inf, b_inf); wall = time.time() - t0
        tm = rec.timing()
    flop = nsites * nen * (lld - 1) * 18.0 ** 3 * 8 * (2 + 8.0 / 3)      # two products + inversion per level
    print("recursion %.1f ms | block_green: kernel %.1f ms, with transfers %.1f ms, wall %.1f ms  (%.2f TFLOP/s in the kernel; g0 = %.0f MB)" % (
        t_rec, tm["hop_ms"], tm["total_ms"], wall * 1e3, flop / tm["hop_ms"] * 1e-9, gr.g0.nbytes / 1e6))
    print("LDOS min %.3e (eta = 0: zero outside the band)" % gr.ldos().min())
    rec.close()
