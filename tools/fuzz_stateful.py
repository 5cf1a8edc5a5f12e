#!/usr/bin/env python3
"""Stateful randomised check: ONE engine handle lives through a random sequence of what an SCF / post-processing run does to it -- new
operator values on the same block structure, a different block structure (blocks zeroed, collinear <-> spin-mixing, hoh on / off), other
recursion sites, another lattice altogether, positions given or not, options toggled, block assembly on the device in between -- and after
every change block Lanczos and Chebyshev moments are compared with the oracle (checker).  What it is after: anything cached inside the
handle (regions, captured graphs, operator schedules, device-assembled blocks, lazily built kernel tables) surviving a change it
should not survive.   tools/fuzz_stateful.py [seconds] [first seed]; exit code 1 on a failure."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import RTOL, objects_from, rel_err
import test_gpu_spmm_random as tsr
from oracle import oracle
from rslmtoasa_amd.recursion import Recursion, chebyshev_scaling

OPTIONS = {"kernels": (0, 1, 2), "spmm5": (0, 1, 2), "s5_lds": (0, 1, 2), "s5_queue": (0, 1, 2), "graph": (0, 1, 2), "orth3": (1, 2), "batch": (0, 1, 3),
           "chain_fold": (1, 2), "s5_host_emit": (0, 1), "s5_octet": (0, 1, 1), "s5_spin_xcd": (0, 1), "side_stream": (0, 1), "cheb_fused": (0, 1), "s5_split": (0, 3), "s5_waves": (8, 12), "sat_pct": (40, 80, 100), "orth_oop": (0, 1)}


def new_problem(rng):
    kk = int(rng.integers(20, 300)); nslots = int(rng.choice([5, 9, 15, 19, 27, 31])); ntype = int(rng.integers(1, 4))
    nmax = int(rng.choice([0, 0, 2, 6])); hoh = bool(rng.integers(0, 2)); collinear = bool(rng.integers(0, 2))
    p = tsr.random_problem(rng, kk, nslots, ntype, nmax, hoh, collinear)
    return p, dict(kk=kk, nslots=nslots, ntype=ntype, nmax=nmax, collinear=collinear)


def new_values(rng, p, meta, restructure):
    """new operator blocks on the same lattice; restructure: change which blocks / quadrants are non-zero and whether hoh is on"""
    hoh = bool(rng.integers(0, 2)) if restructure else bool(p["hoh"])
    collinear = bool(rng.integers(0, 2)) if restructure else meta["collinear"]
    q = tsr.random_problem(rng, meta["kk"], meta["nslots"], meta["ntype"], meta["nmax"], hoh, collinear)
    q["nn"], q["iz"] = p["nn"], p["iz"]
    if restructure and rng.random() < 0.5:                       # some hopping slots vanish for every class (absent blocks in the schedule)
        for s in rng.choice(np.arange(1, meta["nslots"]), size=min(3, meta["nslots"] - 1), replace=False):
            for k in ("ee", "eeo", "hall", "hallo"):
                if k in q:
                    q[k][:, :, s] = 0
    meta["collinear"] = collinear
    return q


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    t0, nstep, bad = time.time(), 0, 0
    p, meta = new_problem(rng)
    lld = 6
    irec = rng.integers(1, meta["kk"] + 1, 3).astype(np.int32)
    rec = Recursion(*objects_from(p, irec, lld, emin=-60.0, emax=60.0), device=0)
    while time.time() - t0 < budget:
        action = rng.choice(["values", "structure", "sites", "lattice", "options", "positions", "assemble", "repeat"], p=[0.25, 0.15, 0.15, 0.1, 0.15, 0.05, 0.05, 0.1])
        if action == "lattice" or (action == "sites" and rng.random() < 0.2):
            if action == "lattice":
                p, meta = new_problem(rng)
            lld = int(rng.integers(2, 9))
            irec = rng.integers(1, meta["kk"] + 1, int(rng.choice([1, 2, 5, 70]))).astype(np.int32)
            ham, lat, ctl, en = objects_from(p, irec, lld, emin=-60.0, emax=60.0)
            if rng.random() < 0.5:
                lat.cr = np.asfortranarray(rng.standard_normal((3, meta["kk"])))
            rec.hamiltonian, rec.lattice, rec.control = ham, lat, ctl
            rec.restore_to_default(); rec.update_lattice(); rec.update_hamiltonian()
        elif action == "sites":
            irec = rng.integers(1, meta["kk"] + 1, len(irec)).astype(np.int32)
            rec.lattice.irec = irec
        elif action in ("values", "structure"):
            p = new_values(rng, p, meta, action == "structure")
            rec.hamiltonian = objects_from(p, irec, lld)[0]
            rec.update_hamiltonian()
        elif action == "options":
            for k in rng.choice(list(OPTIONS), size=3, replace=False):
                rec.set_option(str(k), int(rng.choice(OPTIONS[str(k)])))
            rec.update_hamiltonian()
        elif action == "positions":
            rec.lattice.cr = np.asfortranarray(rng.standard_normal((3, meta["kk"])))
            rec.update_lattice()
        elif action == "assemble":                               # unrelated blocks pass through the device-assembly buffers
            hm = np.asfortranarray(rng.standard_normal((9, 9, meta["nslots"], 4, meta["ntype"])) + 0j)
            rec._assemble(0, hm, np.ones((meta["nslots"], meta["ntype"]), np.int32) if p["hoh"] else None,
                          np.asfortranarray(rng.standard_normal((18, 18, meta["ntype"])) + 0j) if p["hoh"] else None)
        tag = "step %d %-9s kk=%d slots=%d types=%d nmax=%d hoh=%d collinear=%d sites=%d lld=%d" % (nstep, action, meta["kk"], meta["nslots"], meta["ntype"], meta["nmax"], p["hoh"], meta["collinear"], len(irec), lld)
        try:
            o = oracle.Oracle(p)
            # a chain whose Krylov space runs out ends the reference's run ('Diagonalization error', recursion.f90:1942): the oracle raises, the
            # engine returns RSREC_ERR_EIG (tools/fuzz_recursion.py) -- both or neither
            engine_fatal = False
            try:
                rec.recur_b()
            except Exception as e:
                if "Diagonalization error" not in str(e):
                    raise
                engine_fatal = True
            n = len(irec)
            try:
                a_o, b_o = o.block_lanczos(irec, lld)
                oracle_fatal = False
            except oracle.DiagonalizationError:
                oracle_fatal = True
            if engine_fatal or oracle_fatal:
                ok = engine_fatal == oracle_fatal
                print("%s %s  (oracle fatal=%s, engine fatal=%s)" % ("ok  " if ok else "EDGE", tag, oracle_fatal, engine_fatal), flush=True)
                nstep += 1
                continue                                         # (an EDGE -- rounding decides the sign of a vanishing eigenvalue -- is reported, not counted)
            errs = [rel_err(rec.a_b[:, :, :, :n], a_o), rel_err(rec.b2_b[:, :, :, :n], b_o)]
            if rng.random() < 0.5:
                rec.chebyshev_recur()
                mu_o, div = o.chebyshev(irec, lld, *chebyshev_scaling(-60.0, 60.0))
                errs.append(rel_err(rec.mu_n[:, :, :, :n], mu_o))
            ok = max(errs) < RTOL
            verdict = "ok  " if ok else "FAIL"
            if not ok:
                # chains of a random directed graph can die out: B_n^2 nearly singular, rounding amplified by its condition number per level in
                # the oracle as in the engine (tools/fuzz_recursion.py): reported as ILL while the error stays below eps * cond^2
                cond = 1.0
                for l in range(lld):
                    for sidx in range(n):
                        try:
                            ev = np.linalg.eigvalsh(0.5 * (b_o[:, :, l, sidx] + b_o[:, :, l, sidx].conj().T))
                            cond = max(cond, ev.max() / max(abs(ev.min()), 1e-300))
                        except Exception:
                            cond = np.inf
                if max(errs) < 1e-16 * cond ** 2:
                    verdict, ok = "ILL ", True
                    tag += " cond(B^2)=%.1e" % cond
            print("%s %s  worst %.1e" % (verdict, tag, max(errs)), flush=True)
        except Exception as e:
            ok = False
            print("EXC  %s  %r" % (tag, e), flush=True)
        bad += 0 if ok else 1
        nstep += 1
    rec.close()
    print("%d steps, %d failures, %.0f s" % (nstep, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
