#!/usr/bin/env python3
"""Randomised cross-check of the engine against the CPU oracle (checker): random ragged lattices (missing neighbours, several atom types,
impurity atoms, 1..31 slots), random operators (collinear / spin-mixing, +- hoh), random sites / depth / batch and a random set of
library options for every case; block Lanczos and Chebyshev moments must agree with the oracle at the parity bar.  Time-bounded:
tools/fuzz_recursion.py [seconds] [first seed].  Prints one line per case and every failure with its seed; exit code 1 on a failure."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "8")      # the oracle's OpenMP team: the box shows more cores than its cgroup grants
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import RTOL, objects_from, rel_err
from test_gpu_spmm_random import random_problem
from oracle import oracle
from rslmtoasa_amd.recursion import Recursion, chebyshev_scaling

OPTIONS = {"kernels": (0, 1, 2), "spmm5": (0, 1, 2), "s5_lds": (0, 1, 2), "s5_queue": (0, 1, 2), "s5_run_min": (0, 1), "graph": (0, 1, 2), "orth3": (1, 2),
           "batch": (0, 1, 3), "chain_fold": (1, 2), "s5_host_emit": (0, 1), "s5_octet": (0, 1, 1), "s5_spin_xcd": (0, 1), "side_stream": (0, 1), "cheb_fused": (0, 1), "s5_waves": (4, 8, 12), "s5_split": (0, 3), "sat_pct": (40, 80, 100), "orth_oop": (0, 1)}

if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0, ncase, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        kk = int(rng.integers(20, 400))
        nslots = int(rng.choice([1, 2, 5, 9, 15, 19, 27, 31]))
        ntype = int(rng.integers(1, 4))
        nmax = int(rng.choice([0, 0, 1, 3, 9]))
        hoh, collinear = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        nsites = int(rng.choice([1, 2, 5, 11]))
        lld = int(rng.integers(2, 9))
        p = random_problem(rng, kk, nslots, ntype, min(nmax, kk), hoh, collinear)
        irec = rng.integers(1, kk + 1, nsites).astype(np.int32)
        opts = {k: int(rng.choice(v)) for k, v in OPTIONS.items() if rng.random() < 0.4}
        tag = "seed %d: kk=%d slots=%d types=%d nmax=%d hoh=%d collinear=%d sites=%d lld=%d opts=%s" % (seed, kk, nslots, ntype, nmax, hoh, collinear, nsites, lld, opts)
        try:
            rec = Recursion(*objects_from(p, irec, lld, emin=-60.0, emax=60.0), device=0)
            for k, v in opts.items():
                rec.set_option(k, v)
            rec.update_hamiltonian()                              # (options that shape the operator tables apply from the next hand-over)
            o = oracle.Oracle(p)
            errs = []
            noct = 0
            # A chain whose Krylov space runs out ends the reference's run ('Diagonalization error', recursion.f90:1942; the oracle
            # raises, the engine returns RSREC_ERR_EIG).  Whether a vanishing eigenvalue of B^2 comes out as -1e-17 (fatal) or +1e-17
            # (garbage, but finite) is rounding: a case where only one of the two ends is reported as EDGE, not judged.
            try:
                a_o, b_o = o.block_lanczos(irec, lld)
                oracle_fatal = False
            except oracle.DiagonalizationError:
                oracle_fatal = True
            engine_fatal = False
            for rep in range(2):                                  # twice: the second call reuses cached regions / a captured graph
                try:
                    rec.recur_b()
                except Exception as e:
                    if "Diagonalization error" not in str(e):
                        raise
                    engine_fatal = True
                    break
                noct += int(rec.timing()["octet_launches"])
                if not oracle_fatal:
                    errs += [rel_err(rec.a_b, a_o), rel_err(rec.b2_b, b_o)]
            if oracle_fatal or engine_fatal:
                rec.close()
                print("%s %s  (oracle fatal=%s, engine fatal=%s)" % ("ok  " if oracle_fatal == engine_fatal else "EDGE", tag, oracle_fatal, engine_fatal), flush=True)
                ncase += 1; seed += 1
                continue
            rec.chebyshev_recur()
            mu_o, div = o.chebyshev(irec, lld, *chebyshev_scaling(-60.0, 60.0))
            errs.append(rel_err(rec.mu_n, mu_o))
            rec.close()
            extra = " [octet launches %d]" % noct if noct else ""
            if nslots >= 5 and rng.random() < 0.3:                # (one or two slots: the pair chains break down after a level or two, nothing to compare)
                # recur_b_ij / chebyshev_recur_ij: four seeded chains per random atom pair (one pair may be i == j)
                npair = int(rng.integers(1, 4))
                pairs = rng.integers(1, kk + 1, (npair, 2)).astype(np.int32)
                if rng.random() < 0.3:
                    pairs[0, 1] = pairs[0, 0]
                ham, lat, ctl, en = objects_from(p, [1], lld, emin=-60.0, emax=60.0)
                lat.ijpair = pairs
                rp = Recursion(ham, lat, ctl, en, device=0)
                for k, v in opts.items():
                    rp.set_option(k, v)
                rp.update_hamiltonian()
                slots, sa, sc = rp._pair_seeds(pairs, skip_diagonal_repeats=True)
                a_p, b_p = o.block_lanczos_seeded(sa, sc, lld, fatal_ok=True)
                try:
                    rp.recur_b_ij()
                    pair_fatal = False
                except Exception as e:
                    if "Diagonalization error" not in str(e):
                        raise
                    pair_fatal = True
                cond_p = 1.0                                       # (pair chains on a random directed graph often die out: see the ILL rule below)
                for l in range(lld):
                    for c in range(b_p.shape[3]):
                        try:
                            ev = np.linalg.eigvalsh(0.5 * (b_p[:, :, l, c] + b_p[:, :, l, c].conj().T))
                            cond_p = max(cond_p, ev.max() / max(abs(ev.min()), 1e-300))
                        except Exception:
                            cond_p = np.inf
                if pair_fatal or not np.isfinite(a_p).all():
                    extra = " +pairs(%s)" % ("both end in the fatal error" if pair_fatal and not np.isfinite(a_p).all() else "EDGE: one of engine / oracle ends in the fatal error")
                elif cond_p < 1e3:
                    errs += [rel_err(rp.a_b[:, :, :, slots], a_p), rel_err(rp.b2_b[:, :, :, slots], b_p)]
                    extra = " +pairs"
                else:
                    extra = " +pairs(ill-conditioned, not compared)"
                rp.close()
            if collinear and not hoh and rng.random() < 0.3:
                # scalar Haydock recursion (nsp = 1): 18 orbital chains per site on the spin-diagonal 9x9 blocks
                from helpers import rel_err_rows
                p1 = dict(p, nsp=1)
                rs = Recursion(*objects_from(p1, irec, lld, nsp=1, llsp=lld), device=0)
                rs.recur()
                a_s, b_s = oracle.Oracle(p1).scalar_lanczos(irec, lld, lld)
                errs += [rel_err_rows(rs.a[:, :, :nsites, 0], a_s), rel_err_rows(rs.b2[:, :, :nsites, 0], b_s)]
                rs.close()
                extra += " +scalar"
            tag += extra
            ok = div == 0 and max(errs) < RTOL
            verdict = "ok  " if ok else "FAIL"
            if not ok and div == 0:
                # random directed graphs with one or two slots give chains that die out: B_n^2 is then nearly singular and the
                # recursion amplifies rounding by its condition number at every level (in the oracle as in the engine).  Such a case
                # is reported as ILL, not as a failure, while the error stays below eps * cond^2.
                cond = 1.0
                for l in range(lld):
                    for sidx in range(nsites):
                        try:
                            ev = np.linalg.eigvalsh(0.5 * (b_o[:, :, l, sidx] + b_o[:, :, l, sidx].conj().T))
                            cond = max(cond, ev.max() / max(abs(ev.min()), 1e-300))
                        except Exception:
                            cond = np.inf
                # (an unknown condition number -- eigvalsh failed, NaN -- excuses nothing, and no condition number excuses more than 1e-6;
                # tests/test_gpu_breakdown.py holds the round-3 ILL seeds to the COMPILED REFERENCE's own sensitivity instead)
                if np.isfinite(cond) and max(errs) < min(1e-16 * cond ** 2, 1e-6):
                    verdict, ok = "ILL ", True
                    tag += " cond(B^2)=%.1e" % cond
            print("%s %s  worst %.1e" % (verdict, tag, max(errs)), flush=True)
            bad += 0 if ok else 1
        except Exception as e:                                     # an option set the engine refuses is a finding too
            print("EXC  %s  %r" % (tag, e), flush=True)
            bad += 1
        ncase += 1
        seed += 1
    print("%d cases, %d failures, %.0f s" % (ncase, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
