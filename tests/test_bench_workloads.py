"""bench.py's workload builder and work bookkeeping on the CPU (no GPU): the lattices of every --workload, the reference's flop count
(SURVEY.md 8d figures), and the real-basis property recorded in DESIGN.md section 3 (tools/realbasis.py)."""
import importlib.util
import os
import sys

import numpy as np
import pytest

from helpers import load_golden_with_inputs, problem_dict, supercell_problem

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_module(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, [path]
    try:
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    return m


class Args:
    def __init__(self, **kw):
        self.sites, self.cells, self.lld, self.hoh, self.spin_mixing, self.no_positions, self.recur, self.workload = 64, 22, 50, False, False, False, "block", "bcc"
        self.cells_given = False
        self.__dict__.update(kw)


def test_workloads_and_reference_flop_count():
    b = load_module(os.path.join(ROOT, "bench.py"), "bench_mod")
    W = b.build_workload(Args(), 1)
    assert W["key"] == "block_c22_s64_l50" and W["nn"].shape == (10648, 16) and len(W["irec"]) == 64 and W["nmax"] == 0 and W["ntype"] == 1
    # one chain of the 22^3 cell, LL = 50: 6 144 075 block multiplies, 420 252 post-hop atom-steps
    assert b.algorithmic_work(W["nn"], 1, 49, False) == (6144075.0, 420252.0)
    Wf = b.build_workload(Args(workload="fccCu001", recur="chebyshev"), 1)
    assert Wf["key"] == "chebyshev_fccCu001_s64_l50" and Wf["nn"].shape == (9318, 20) and Wf["ntype"] == 3 and Wf["cr"].shape == (3, 9318)
    Wi = b.build_workload(Args(workload="B2FeCo", hoh=True), 2)
    assert Wi["key"] == "block_hoh_B2FeCo_s64_l50" and Wi["nmax"] == 15 and len(Wi["irec"]) == 128
    assert list(Wi["irec"][:15]) == list(range(1, 16)) and len(set(Wi["irec"].tolist())) == 128 and Wi["irec"].max() <= 4152      # the impurity region first, no site twice
    m, a = b.algorithmic_work(Wi["nn"], 1, 49, True)
    assert m > 2 * a > 0                                            # hoh: two passes per level


def test_default_workload_is_the_north_star_cell(monkeypatch):
    """`python bench.py` with no flags measures the cell BASELINE.json's target is quoted on: 46^3 = 97 336 atoms, 64 sites, LL = 50."""
    b = load_module(os.path.join(ROOT, "bench.py"), "bench_mod3")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse_args()
    assert (a.cells, a.sites, a.lld, a.recur, a.workload, a.gpus) == (46, 64, 50, "block", "bcc", 1) and not a.cells_given
    monkeypatch.setattr(sys, "argv", ["bench.py", "--workload", "kubo"])
    assert b.parse_args().cells == 20
    monkeypatch.setattr(sys, "argv", ["bench.py", "--cells", "22"])
    a = b.parse_args()
    assert a.cells == 22 and a.cells_given


def test_periodic_fcc_table_of_the_kubo_workload():
    from helpers import load_golden
    from rslmtoasa_amd.lattice import bcc_supercell
    b = load_module(os.path.join(ROOT, "bench.py"), "bench_mod2")
    z = load_golden("fccPt_kubo")
    nn = bcc_supercell((6, 6, 6), z["slot_vec"], primitive=np.array(b.FCC_PRIMITIVE))
    nb = int(nn[0, 0])
    assert nb == 19 and nn.shape == (216, 20)
    for m in range(1, nb):                                          # every slot is a translation: a permutation of the atoms
        assert len(set(nn[:, m].tolist())) == 216
    d = np.linalg.norm(z["slot_vec"][1:nb], axis=1)
    assert np.sum(np.isclose(d, np.sqrt(0.5))) == 12 and np.sum(np.isclose(d, 1.0)) == 6      # 12 nearest + 6 second neighbours of fcc


@pytest.mark.parametrize("name", ["bccFe_nsp2_block_hoh", "B2FeCo_block_hoh", "fccCu001_block_hoh", "bccFe_nsp4_block", "fccPt_kubo"])
def test_hopping_blocks_are_real_in_a_common_basis(name):
    """every collinear case of the reference admits ONE 9x9 unitary C per spin with C^H H_s C real for all hopping
    blocks (the two-centre integrals in real harmonics); the on-site block with spin-orbit stays complex."""
    rb = load_module(os.path.join(ROOT, "tools", "realbasis.py"), "realbasis_mod")
    g = load_golden_with_inputs(name)
    p = problem_dict(g)
    q, U = rb.transform_operator(p)
    assert q is not None and np.abs(U.conj().T @ U - np.eye(18)).max() < 1e-12
    nb = int(p["nn"][:, 0].max())
    assert np.abs(q["ee"][:, :, 1:nb].imag).max() == 0.0
    back = np.einsum("ab,bcst,cd->adst", U, q["ee"], U.conj().T)
    assert np.abs(back - p["ee"])[:, :, :nb].max() < 1e-12 * np.abs(p["ee"]).max()
    if np.abs(p["lsham"]).max() > 0:
        assert np.abs(q["lsham"].imag).max() > 1e-6 * np.abs(q["lsham"]).max()      # l.s is not real there
