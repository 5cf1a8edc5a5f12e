"""the O(N) cell-list neighbour search (fortran/lattice_cells.f90, overriding lattice%nncal) against the
reference's own all-pairs search, through the reference's whole pre-processing (build_data, bravais, build_surf_full,
newclu, structb): the resulting lattice%nn table and the `map` / `clust` files must be byte-identical.

CPU only.  oracle/_ref/nncal_check.x (tests/fortran/nncal_check.f90 + the compiled reference modules, fortran/build.sh) runs one
mode per process; inputs are the reference's test cases held as data under tests/golden/scf/inputs."""
import filecmp
import os
import shutil
import subprocess

import pytest

from oracle.make_fixtures import patch_namelist
from rslmtoasa_amd._proc import run_with_unlimited_stack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INPUTS = os.path.join(ROOT, "tests", "golden", "scf", "inputs")
EXE = os.path.join(ROOT, "oracle", "_ref", "nncal_check.x")

PERIODIC = {"lattice": {"pbc": ".true.", "b1": ".true.", "b2": ".true.", "b3": ".true.", "n1": "10", "n2": "10", "n3": "10"}}
SLAB = {"lattice": {"pbc": ".true.", "b1": ".true.", "b2": ".true.", "b3": ".false.", "n1": "12", "n2": "9", "n3": "7"}}
# 3 a = 8.58 A against a 3.0 A cut-off: two cells along x, each the other's only neighbour (must not be visited twice).  A periodic
# box below two cut-offs is not a valid input of the reference itself (remd stops with "VECTOR NOT FOUND" at n1 = 2).
TINY = {"lattice": {"pbc": ".true.", "b1": ".true.", "b2": ".true.", "b3": ".true.", "n1": "3", "n2": "5", "n3": "9"}}
CASES = [("bulk_bccFe", {}), ("impurity_B2FeCo", {}), ("surface_fccCu001", {}), ("regression_bccFe_lanczos", {}),
         ("bulk_bccFe", PERIODIC), ("bulk_bccFe", SLAB), ("bulk_bccFe", TINY)]


def run(mode, case, patch, where):
    work = where / mode
    shutil.copytree(os.path.join(INPUTS, case), work)
    inp = work / "input.nml"
    inp.write_text(patch_namelist(inp.read_text(), patch))
    r = run_with_unlimited_stack([EXE, mode], cwd=work, env={"OMP_NUM_THREADS": "4"}, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    return work, r.stdout


@pytest.mark.parametrize("case,patch", CASES, ids=["%s%s" % (c, "".join("_%s%s%s" % (p["lattice"]["n1"], p["lattice"]["n2"], p["lattice"]["n3"]) for p in [q] if q)) for c, q in CASES])
def test_cell_list_search_reproduces_the_reference_tables(case, patch, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/nncal_check.x not built (needs the compiled reference: build container only)")
    ref, _ = run("ref", case, patch, tmp_path)
    new, out = run("cells", case, patch, tmp_path)
    assert filecmp.cmp(ref / "nn_ref.bin", new / "nn_cells.bin", shallow=False)
    for f in ("map", "clust"):
        assert filecmp.cmp(ref / f, new / f, shallow=False), f
    # the search really was the cell list (the override passes tiny or partial-update problems on to the parent)
    ev, total = [int(x) for x in out.split("pairs evaluated")[1].split("of")]
    assert 0 < ev < total and (total < 1e6 or ev < total / 10)
