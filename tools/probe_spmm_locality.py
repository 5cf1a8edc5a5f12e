#!/usr/bin/env python3
"""Diagnostic: how fast is the SpMM kernel when every neighbour gather hits cache?

Builds an artificial neighbour table in which the 14 neighbours of atom i are the atoms of its own group of 8
(perfect locality: each group reads 8 distinct blocks instead of 120) and seeds one atom per group so the whole
lattice is active from level 1.  Same flops per launch as the real bcc lattice; only the gather addresses differ.
The numbers it prints separate "MFMA issue bound" from "L2-miss traffic bound" for k_mfma_spmm."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden, objects_from, supercell_problem
from rslmtoasa_amd.recursion import Recursion
import ctypes as C


def run(p, seeds, coefs, lld, nch, label):
    ham, lat, ctl, en = objects_from(p, [1], lld)
    rec = Recursion(ham, lat, ctl, en)
    sa = np.ascontiguousarray(np.tile(seeds, (nch, 1)), dtype=np.int32)
    sc = np.ascontiguousarray(np.tile(coefs, (nch, 1)), dtype=np.complex128)
    a_b = np.zeros((18, 18, lld, nch), np.complex128, order="F")
    b2_b = np.zeros_like(a_b)
    for _ in range(2):
        rc = rec._L.rsrec_block_lanczos_seeded(rec._h, nch, sa.shape[1], sa.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), lld,
                                               a_b.ctypes.data_as(C.c_void_p), b2_b.ctypes.data_as(C.c_void_p))
        assert rc == 0
    tm = rec.timing()
    hop_flop = 46656.0 * tm["block_multiplies"]
    print("%-28s hop %.1f ms  %.1f TF/s (nominal)   total %.1f ms" % (label, tm["hop_ms"], hop_flop / tm["hop_ms"] * 1e-9, tm["total_ms"]))
    rec.close()


if __name__ == "__main__":
    lld, nch = 12, 64
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
    p = supercell_problem((n, n, n))
    kk = p["nn"].shape[0]
    seeds = np.arange(1, kk + 1, 8, dtype=np.int32)          # one seed per group of 8 consecutive atoms
    coefs = np.full(len(seeds), 1.0 / np.sqrt(len(seeds)), dtype=np.complex128)
    run(p, seeds, coefs, lld, nch, "bcc %d^3 (real gathers)" % n)
    q = dict(p)
    nn = p["nn"].copy()
    base = (np.arange(kk) // 8) * 8
    for s in range(1, 15):
        nn[:, s] = np.minimum(base + (s % 8), kk - 1) + 1
    q["nn"] = nn
    run(q, seeds, coefs, lld, nch, "clique-of-8 (all cache hits)")
    # neighbours at index distance +-(8 m + 1), m = 1..7: 120 distinct blocks per group like the real lattice, but every block is used by atoms
    # within a window of 224 consecutive atoms: the gathers miss L1 and hit L2 (no fabric traffic beyond the compulsory 1x)
    r = dict(p)
    nn = p["nn"].copy()
    idx = np.arange(kk)
    for s in range(1, 15):
        d = (8 * ((s + 1) // 2) + 1) * (1 if s % 2 else -1)      # 9, -9, 17, -17, ...: connects all residues mod 8
        nn[:, s] = (idx + d) % kk + 1
    r["nn"] = nn
    run(r, seeds, coefs, lld, nch, "index-near neighbours (L2 hits)")
