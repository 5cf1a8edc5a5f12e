#!/usr/bin/env python3
"""Randomised cross-checks of the remaining entry points against the CPU oracle / numpy restatements (checkers):
  local-axis  rsrec_block_lanczos_local_axis with random SU(2) spin rotations per site: the oracle runs every site on blocks it rotated
              itself, H' = R^H H R for ee / eeo / enim (not lsham), as rotate_to_local_axis does (hamiltonian.f90:2442-2465)
  kubo        rsrec_kubo_moments with random velocity operators, seeds, phases, depth, +- hoh on random ragged lattices
  apply       rsrec_apply_operator (plain / hoh operator on whole vectors) against the numpy loops of tests/test_gpu_spmm_random.py
  assemble    rsrec_assemble_blocks with random hmag / obarm / neighbour types against oracle.assemble_blocks
tools/fuzz_misc.py [seconds] [first seed]; exit code 1 on a failure."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "8")
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import RTOL, objects_from, rel_err
import test_gpu_random_operator as tro
import test_gpu_spmm_random as tsr
from oracle import oracle
from rslmtoasa_amd.recursion import Recursion, chebyshev_scaling


def su2(rng):
    q = rng.standard_normal(4); q /= np.linalg.norm(q)
    u = np.array([[q[0] + 1j * q[3], q[2] + 1j * q[1]], [-q[2] + 1j * q[1], q[0] - 1j * q[3]]])
    return np.kron(u, np.eye(9))                       # spin-major 18x18: orbitals 1-9 up, 10-18 down


def case_local_axis(rng, seed):
    hoh = bool(rng.integers(0, 2))
    p = tro.random_problem(seed, hoh)
    kk = p["nn"].shape[0]
    n = int(rng.integers(1, 5)); lld = int(rng.integers(3, 9))
    irec = rng.choice(kk, n, replace=False).astype(np.int32) + 1
    rot = np.stack([su2(rng) for _ in range(n)], axis=2)
    rec = Recursion(*objects_from(p, irec, lld, nsp=4), device=0)
    rec.recur_b_local_axis(rot)
    worst = 0.0
    for s in range(n):
        R = rot[:, :, s]
        q = dict(p)
        for k in ("ee", "eeo", "enim"):
            if k in p:
                q[k] = np.asfortranarray(np.einsum("ji,jk...,kl->il...", R.conj(), p[k], R))
        a_o, b_o = oracle.Oracle(q).block_lanczos(irec[s:s + 1], lld)
        worst = max(worst, rel_err(rec.a_b[:, :, :, s:s + 1], a_o), rel_err(rec.b2_b[:, :, :, s:s + 1], b_o))
    rec.close()
    return worst, RTOL, "hoh=%d sites=%d lld=%d" % (hoh, n, lld)


def ragged(rng):
    kk = int(rng.integers(30, 250)); nslots = int(rng.choice([2, 5, 9, 15, 19, 27, 31])); ntype = int(rng.integers(1, 4))
    nmax = int(rng.choice([0, 0, 2, 5])); hoh = bool(rng.integers(0, 2)); collinear = bool(rng.integers(0, 2))
    return tsr.random_problem(rng, kk, nslots, ntype, nmax, hoh, collinear), "kk=%d slots=%d types=%d nmax=%d hoh=%d collinear=%d" % (kk, nslots, ntype, nmax, hoh, collinear)


def case_kubo(rng, seed):
    p, tag = ragged(rng)
    kk, ns, nt = p["nn"].shape[0], p["ee"].shape[2], p["ee"].shape[3]
    cond_ll = int(rng.integers(2, 7)); nvec = int(rng.integers(1, 12)); nseed = int(rng.choice([1, 3, kk]))      # up to 8 vectors share a launch
    vbatch, lchunk = int(rng.choice([0, 0, 1, 3])), int(rng.choice([0, 0, 1, 2]))
    blk = lambda: np.asfortranarray(0.2 * (rng.standard_normal((18, 18, ns, nt)) + 1j * rng.standard_normal((18, 18, ns, nt))))
    v_a, v_b = blk(), blk()
    vo_a, vo_b = (blk(), blk()) if p["hoh"] else (None, None)
    seeds = np.stack([rng.choice(kk, nseed, replace=False) + 1 for _ in range(nvec)]).astype(np.int32)
    coefs = np.exp(2j * np.pi * rng.random((nvec, nseed))) / np.sqrt(nseed)
    emin, emax = -60.0, 60.0
    rec = Recursion(*objects_from(p, [1], cond_ll, emin=emin, emax=emax), device=0)
    rec.set_option("kubo_vbatch", vbatch); rec.set_option("kubo_lchunk", lchunk)
    mu = rec.compute_moments_stochastic(v_a, v_b, cond_ll, vo_a=vo_a, vo_b=vo_b, seeds=seeds, coefs=coefs)
    rec.close()
    a, b = chebyshev_scaling(emin, emax)
    ref = oracle.Oracle(p).kubo_moments(seeds, coefs, cond_ll, a, b, v_a, v_b, vo_a, vo_b)
    err = max(np.abs(mu[..., i] - ref[..., i]).max() / max(np.abs(ref[..., i]).max(), 1e-300) for i in range(nvec))
    return float(err), RTOL, tag + " cond_ll=%d nvec=%d nseed=%d vbatch=%d lchunk=%d" % (cond_ll, nvec, nseed, vbatch, lchunk)


def case_apply(rng, seed):
    p, tag = ragged(rng)
    kk = p["nn"].shape[0]
    rec = Recursion(*objects_from(p, [1], 4), device=0)
    x = np.asfortranarray(rng.standard_normal((18, 18, kk)) + 1j * rng.standard_normal((18, 18, kk)))
    a, b = float(rng.uniform(0.5, 3.0)), float(rng.uniform(-1.0, 1.0))
    want = tsr.ham_vec_numpy(p, x, a, b)
    got = rec.ham_hoh_vec_matmul(x, a, b) if p["hoh"] else rec.ham_vec_matmul(x, a, b)
    err = np.abs(got - want).max() / np.abs(want).max()
    if p["hoh"]:
        ls = np.stack([p["lsham"][:, :, t - 1] for t in p["iz"]], axis=2)
        plain = (tsr.apply_blocks(p, x, False) + np.einsum("ijk,jlk->ilk", ls, x) - b * x) / a
        err = max(err, np.abs(rec.ham_vec_matmul(x, a, b) - plain).max() / np.abs(plain).max())
    rec.close()
    return float(err), 5e-13, tag


def case_assemble(rng, seed):
    p, tag = ragged(rng)
    ncls, nsl, ntype = int(rng.integers(1, 20)), int(rng.integers(1, 32)), int(rng.integers(1, 5))
    hm = np.asfortranarray(rng.standard_normal((9, 9, nsl, 4, ncls)) + 1j * rng.standard_normal((9, 9, nsl, 4, ncls)))
    ty = rng.integers(0, ntype + 1, (nsl, ncls)).astype(np.int32)
    ob = np.asfortranarray(rng.standard_normal((18, 18, ntype)) + 1j * rng.standard_normal((18, 18, ntype)))
    rec = Recursion(*objects_from(p, [1], 4), device=0)
    part = int(rng.integers(0, 2))
    b, bo = rec._assemble(part, hm, ty if p["hoh"] else None, ob if p["hoh"] else None)
    rec.close()
    rb, rbo = oracle.assemble_blocks(hm, ty if p["hoh"] else None, ob if p["hoh"] else None)
    err = 0.0 if np.array_equal(b, rb) else np.inf
    if p["hoh"]:
        err = max(err, np.abs(bo - rbo).max() / np.abs(rbo).max())
    return float(err), 1e-14, "classes=%d slots=%d types=%d hoh=%d part=%d" % (ncls, nsl, ntype, p["hoh"], part)


CASES = {"local-axis": case_local_axis, "kubo": case_kubo, "apply": case_apply, "assemble": case_assemble}

if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else list(CASES)
    t0, ncase, bad, count = time.time(), 0, 0, {k: 0 for k in CASES}
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        kind = only[seed % len(only)]
        try:
            err, tol, tag = CASES[kind](rng, seed)
            ok = err <= tol
            print("%s seed %d %-10s %s  err %.1e" % ("ok  " if ok else "FAIL", seed, kind, tag, err), flush=True)
        except Exception as e:
            ok = False
            print("EXC  seed %d %-10s %r" % (seed, kind, e), flush=True)
        bad += 0 if ok else 1
        count[kind] += 1
        ncase += 1
        seed += 1
    print("%d cases %s, %d failures, %.0f s" % (ncase, count, bad, time.time() - t0))
    sys.exit(1 if bad else 0)
