#!/usr/bin/env python3
"""Randomised cross-check of the stages behind the recursion against the CPU oracle (checker): random Hermitian operators on a periodic
bcc cell (full 18x18 hopping blocks with spin-flip parts, +- hoh), random sites / depth / energy mesh / complex increment eta /
sym_term; per case the engine's zsqr, terminator, block Green function, the resident LDOS pipeline (rsrec_block_ldos) and the Chebyshev
Green function are compared with the oracle's run on the ENGINE's coefficients (so that only the stage under test differs).
tools/fuzz_stages.py [seconds] [first seed]; exit code 1 on a failure."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import objects_from
from test_gpu_random_operator import random_problem
from test_gpu_ldos import ldos_from_g0
from oracle import oracle
from rslmtoasa_amd.green import Green
from rslmtoasa_amd.recursion import Recursion


def rel(x, ref, floor=0.0):
    """max |x - ref| / max |ref|; NaNs (energies outside the Chebyshev window give them in the reference too) must sit at the same places"""
    nx, nr = np.isnan(x), np.isnan(ref)
    if not np.array_equal(nx, nr):
        return np.inf
    if nr.all():
        return 0.0
    return float(np.abs(x[~nr] - ref[~nr]).max() / max(np.abs(ref[~nr]).max(), floor, 1e-300))


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, ncase, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        hoh = bool(rng.integers(0, 2))
        p = random_problem(seed, hoh, scale=float(rng.choice([0.03, 0.08])))
        kk = p["nn"].shape[0]
        nsites = int(rng.choice([1, 2, 7]))
        lld = int(rng.integers(4, 13))
        irec = rng.integers(1, kk + 1, nsites).astype(np.int32)
        nen = int(rng.integers(3, 40))
        width = float(rng.choice([1.0, 3.0, 8.0]))
        ene = np.sort(rng.uniform(-width, width, nen))
        eta = complex(rng.choice([0.0, 0.002, 0.05]), rng.choice([0.0, 0.004, 0.02]))
        sym = bool(rng.integers(0, 2))
        tag = "seed %d: hoh=%d sites=%d lld=%d nen=%d width=%.0f eta=%s sym_term=%d" % (seed, hoh, nsites, lld, nen, width, eta, sym)
        try:
            rec = Recursion(*objects_from(p, irec, lld, nsp=4, emin=-6.0, emax=6.0), device=0)
            rec.recur_b()
            gr = Green(rec, ene, sym_term=sym)
            res = gr.block_ldos(eta=eta)                                  # from the resident coefficients (b2_b still B^2)
            b2 = rec.b2_b[:, :, :, :nsites].copy()
            rec.zsqr()
            errs = {"zsqr": rel(rec.b2_b[:, :, :, :nsites], oracle.zsqr(b2))}
            a_inf, b_inf, a0, b0 = gr.terminator(nsites=nsites)
            ao, bo, a0o, b0o = oracle.terminator(rec.a_b[:, :, :, :nsites], rec.b2_b[:, :, :, :nsites])
            errs["term"] = max(rel(a_inf, ao), rel(b_inf, bo))
            g0 = gr.block_green(a_inf, b_inf, eta=eta, nsites=nsites).copy()
            g0o = np.stack([oracle.block_green(rec.a_b[:, :, :, s], rec.b2_b[:, :, :, s], ene, a_inf[:, :, s], b_inf[:, :, s], eta=eta, sym_term=sym) for s in range(nsites)], axis=3)
            errs["green"] = rel(g0, g0o)
            # conditioning of the continued fraction on this mesh: the oracle's own answer for energies shifted by 1e-14 (a mesh point next to a
            # pole of a short chain on the real axis amplifies rounding by 1e6 and more, in either code)
            g0s = np.stack([oracle.block_green(rec.a_b[:, :, :, s], rec.b2_b[:, :, :, s], ene * (1 + 1e-14) + 1e-15, a_inf[:, :, s], b_inf[:, :, s], eta=eta, sym_term=sym) for s in range(nsites)], axis=3)
            sens = rel(g0s, g0o)
            dt, da, dl = ldos_from_g0(g0o)
            # the densities are -Im g0_jj / pi: outside the band (or with a real eta) that imaginary part is rounding noise of |g0|, and near a
            # pole of a short chain the inverse is ill-conditioned in both codes -- the scale of the comparison is |g0|, not the density
            fl = float(np.abs(g0o).max() / np.pi)
            errs["ldos"] = max(rel(res["dosial"], dl, fl), rel(res["dosia"], da, 18 * fl), rel(res["dtot"], dt, 18 * nsites * fl), rel(res["a_inf"], a_inf), rel(res["b_inf"], b_inf))
            rec.chebyshev_recur()
            gc = gr.chebyshev_green(nsites=nsites)
            gco = np.stack([oracle.chebyshev_green(rec.mu_n[:, :, :, s], ene, -6.0, 6.0) for s in range(nsites)], axis=3)
            errs["cheb_green"] = rel(gc, gco)
            rec.close()
            tol = {"zsqr": 1e-12, "term": 1e-12, "green": max(1e-10, 10 * sens), "ldos": max(1e-10, 10 * sens), "cheb_green": 1e-11}
            if sens > 1e-11:
                tag += " (ill-conditioned mesh point: oracle sensitivity %.1e)" % sens
            ok = all(errs[k] <= tol[k] for k in tol)
            print("%s %s  %s" % ("ok  " if ok else "FAIL", tag, " ".join("%s %.1e" % kv for kv in errs.items())), flush=True)
            bad += 0 if ok else 1
        except Exception as e:
            print("EXC  %s  %r" % (tag, e), flush=True)
            bad += 1
        ncase += 1
        seed += 1
    print("%d cases, %d failures, %.0f s" % (ncase, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
