"""The C-ABI library loads on a CPU-only box and exports every symbol include/rsrec.h declares (no compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rslmtoasa_amd import _lib
from rslmtoasa_amd.lattice import active_region_sizes, bcc_supercell, spread_sites
from rslmtoasa_amd.recursion import chebyshev_scaling, site_partition
from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "rsrec.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rsrec_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), "librsrec.so does not export " + s
    assert sorted(syms) == _lib.exported_symbols()
    assert L.rsrec_version() >= 100


def test_site_partition_matches_oracle(oracle_lib):
    for n in (1, 5, 64, 97):
        for p in (1, 2, 4, 8):
            for r in range(p):
                assert site_partition(r, p, n) == oracle_lib.site_partition(r, p, n)


def test_chebyshev_scaling_matches_reference_fixture():
    g = load_golden("bccFe_nsp2_cheb")
    a, b = chebyshev_scaling(g["emin"], g["emax"])
    assert a == g["acheb"] and b == g["bcheb"]


def test_supercell_table_and_region_growth():
    vec = load_golden("bccFe_nsp2_block")["slot_vec"]
    nn = bcc_supercell((22, 22, 22), vec)
    assert nn.shape == (10648, 16) and np.all(nn[:, 0] == 15) and np.all(nn[:, 15] == 0)
    assert nn[:, 1:15].min() >= 1 and nn[:, 1:15].max() <= 10648
    # every slot is a permutation of the atoms (translation), and slot m / its inverse slot are mutual
    for m in range(1, 15):
        assert len(np.unique(nn[:, m])) == 10648
    # region growth quoted in SURVEY.md 8(d)
    assert active_region_sizes(nn, 1, 10) == [15, 65, 175, 369, 671, 1105, 1695, 2465, 3439, 4641]
    assert list(spread_sites(10648, 4)) == [1, 2663, 5325, 7987]


def test_no_cpu_fallback_when_no_device():
    """On a box without a GPU rsrec_create must fail loudly (no silent CPU path)."""
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present")
    except ImportError:
        pass
    L = _lib.lib()
    h = C.c_void_p()
    rc = L.rsrec_create(C.byref(h), 0)
    assert rc != 0 and not h.value
