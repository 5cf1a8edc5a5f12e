#!/usr/bin/env python3
"""Host cost of handing a rebuilt Hamiltonian to the engine (rsrec_set_hamiltonian: copies, l.s fold, hoh operator set, swizzle into
MFMA fragment streams, uploads) -- what a device-side assembly could remove.  Every SCF iteration pays it once."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden_with_inputs, objects_from, problem_dict, supercell_problem
from rslmtoasa_amd.recursion import Recursion

if __name__ == "__main__":
    cases = [("bcc Fe stencil (1 type, 15 slots)", supercell_problem((12, 12, 12)), np.array([1], np.int32)),
             ("bcc Fe stencil, hoh", supercell_problem((12, 12, 12), hoh=True), np.array([1], np.int32))]
    for name in ("B2FeCo_block", "B2FeCo_block_hoh", "fccCu001_block_hoh"):
        g = load_golden_with_inputs(name)
        cases.append((name + " (%d classes, %d slots)" % (int(g["nmax"]) + g["ee"].shape[3], int(g["nn"][:, 0].max())), problem_dict(g), g["irec"]))
    for label, p, irec in cases:
        rec = Recursion(*objects_from(p, irec, 12))
        ts = []
        for _ in range(20):
            t0 = time.perf_counter(); rec.update_hamiltonian(); ts.append(time.perf_counter() - t0)
        rec.recur_b(); tm = rec.timing()
        t0 = time.perf_counter(); rec.recur_b(); w = time.perf_counter() - t0
        # one SCF iteration as the reference drives it: new blocks, then ONE recursion call (self.f90:777-806)
        it = {}
        for opt in (1, 2):
            rec.set_option("spmm5", opt)
            tt = []
            for _ in range(6):
                t0 = time.perf_counter(); rec.update_hamiltonian(); rec.recur_b(); tt.append(time.perf_counter() - t0)
            it[opt] = 1e3 * np.median(tt[1:])
        print("%-48s set_hamiltonian %.3f ms (min %.3f)   recur_b (lld 12, %d site(s)) %.2f ms   set + recur_b per SCF iteration: %.2f ms (spmm5=1) %.2f ms (spmm5=2)"
              % (label, 1e3 * np.median(ts), 1e3 * min(ts), len(irec), 1e3 * w, it[1], it[2]))
        rec.close()
