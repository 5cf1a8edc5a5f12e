"""End-to-end drop-in test: the reference's own SCF workflow (compiled reference modules) with the GPU recursion type
(fortran/recursion_gpu.f90) swapped in, against the reference's committed expected values
(tests/scf/references/*/ref.json and tests/regression/bccFe_lanczos/Fe.nml.ref, copied as data to tests/golden/scf/).

Two builds of the boundary, both linked in the build container and travelling to the GPU box with oracle/_ref/:
  * "dropin": oracle/_ref/rslmto_dropin.x (fortran/build_dropin.sh) -- the reference's OWN main program, linked from its unmodified
    main.f90 / calculation.f90 / self.f90 ... with the GPU types behind the reference's module names (fortran/shadow/): the zero-edit
    drop-in BASELINE.json's north_star asks for.  It runs every case of the manifest, the conductivity post-processing included
    (`post_processing = 'conductivity'` is the reference's own branch, calculation.f90:206).
  * "driver": oracle/_ref/rslmto_gpu.x / kubo_gpu.x (fortran/build.sh) -- the reference's object code + a driver program of ours that
    declares the GPU types explicitly (the edited-declarations integration; also the host of the flows the reference's program has no
    switch for: LDOS only, the library's own communicator).
Comparison rule = the reference's own (tests/run_test.py:201-219): a value fails only if BOTH abs and rel differences exceed the
tolerance."""
import json
import os
import re
import shutil
import subprocess

import pytest

from helpers import require_built
from oracle.make_fixtures import patch_namelist
from rslmtoasa_amd._proc import run_with_unlimited_stack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCF = os.path.join(ROOT, "tests", "golden", "scf")
EXE = os.path.join(ROOT, "oracle", "_ref", "rslmto_gpu.x")
DROPIN = os.path.join(ROOT, "oracle", "_ref", "rslmto_dropin.x")
MANIFEST = json.load(open(os.path.join(SCF, "manifest.json")))

pytestmark = pytest.mark.gpu


def read_nml_value(path, key, index=None):
    txt = open(path).read()
    m = re.search(r"(?im)^\s*%s(\(:\))?\s*=\s*(.+)$" % re.escape(key), txt)
    assert m, "%s not found in %s" % (key, path)
    vals = [float(v.replace("D", "E").replace("d", "e")) for v in re.findall(r"[-+]?[0-9]*\.?[0-9]+(?:[eEdD][-+]?[0-9]+)?", m.group(2))]
    return vals[0] if index is None else vals[index - 1]


def close(got, exp, abs_tol, rel_tol):
    d = abs(got - exp)
    return d <= abs_tol or d <= rel_tol * abs(exp)


def fortran_float(tok):
    """A Fortran es16.6 field: exponents of three digits are written without the 'E' (-1.880039+133)."""
    m = re.fullmatch(r"([-+]?[0-9.]+)([-+][0-9]{3})", tok)
    return float(m.group(1) + "e" + m.group(2)) if m else float(tok.replace("D", "E").replace("d", "e"))


@pytest.mark.parametrize("build", ["dropin", "driver"])
@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_scf_workflow_with_gpu_recursion(name, build, tmp_path):
    case = MANIFEST[name]
    exe = DROPIN if build == "dropin" else os.path.join(os.path.dirname(EXE), case.get("exe", "rslmto_gpu.x"))    # kubo_gpu.x: the conductivity post-processing workflow
    require_built(exe)
    work = tmp_path / "run"
    shutil.copytree(os.path.join(SCF, case["inputs"]), work)
    inp = work / "input.nml"
    inp.write_text(patch_namelist(inp.read_text(), case["patch"]))
    # RSREC_REPORT: the library prints its life-time counters when the process exits (the reference's main program cannot be asked)
    r = run_with_unlimited_stack([exe], cwd=work, env={"OMP_NUM_THREADS": "8", "RSREC_REPORT": "1"}, timeout=1500, scrub=False)   # the child drives the GPU itself
    log = r.stdout + r.stderr
    assert r.returncode == 0, log[-3000:]
    assert "fatal" not in log.lower(), log[-3000:]                      # tests/run_test.py:119-131
    m = re.search(r"rsrec report: library_calls=(\d+)", log)
    assert m and int(m.group(1)) >= 1, log[-2000:]                       # the GPU library did the work (there is no other path, but say so)
    if "exe" in case:
        if build == "driver":
            assert "compute_moments_stochastic wall time" in log
            print("\n".join(l for l in log.splitlines() if "compute_moments_stochastic wall time" in l))
    elif "'block'" in str(case["patch"]):
        # block recursions: the Green function (green%bgreen) also ran on the GPU (fortran/green_gpu.f90); its timer region is
        # listed in the reference's own timing report
        assert "bgreen-gpu" in log, log[-3000:]
        # ... and the densities of states of calculate_fermi (totaldos.out below) came from the device LDOS stage
        # (fortran/bands_gpu.f90) except in local-axis runs, whose resident coefficients are the un-rotated ones
        assert ("ldos-gpu" in log) == ("local_axis" not in str(case["patch"])), log[-3000:]
    if "exe" not in case:
        # ee / eeo / hall / hallo were assembled on the device (fortran/hamiltonian_gpu.f90) and the recursion took them from there
        n_asm = int(re.search(r"device_assemblies=(\d+)", log).group(1))
        n_dev = int(re.search(r"operator_arrays_from_device=(\d+)", log).group(1))
        hoh, imp = ".true." in str(case["patch"].get("hamiltonian", {}).get("hoh", "")), "impurity" in case["inputs"]
        assert n_asm >= 1 and n_dev == (2 if hoh else 1) * (2 if imp else 1), (n_asm, n_dev, log[-1500:])
    if name == "Regression_bccFe_lanczos" and build == "dropin":
        # scalar recursion: green%sgreen took dos%density from the device (fortran/dos_gpu.f90 behind the reference's density_of_states_mod)
        assert "density-gpu" in log, log[-3000:]
    if "'chebyshev'" in str(case["patch"]) and "exe" not in case:
        assert "chebyshev-green-gpu" in log, log[-3000:]
    at, rt = case["abs_tol"], case["rel_tol"]
    bad = []
    for fn, keys in case["expected"].get("nml", {}).items():
        for key, exp in keys.items():
            if isinstance(exp, dict):
                for idx, e in exp.items():
                    got = read_nml_value(work / fn, key, int(idx))
                    if not close(got, e, at, rt):
                        bad.append((fn, key, idx, got, e))
            else:
                got = read_nml_value(work / fn, key)
                if not close(got, exp, at, rt):
                    bad.append((fn, key, got, exp))
    for fn, rows in case["expected"].get("text", {}).items():
        lines = (work / fn).read_text().splitlines()
        for row, cols in rows.items():
            vals = lines[int(row) - 1].split()
            for col, e in cols.items():
                got = fortran_float(vals[int(col) - 1])
                if not close(got, e, at, rt):
                    bad.append((fn, row, col, got, e))
    assert not bad, bad


LDOS_CASES = ["Example_bulk_bccFe_nsp2_block", "Example_impurity_B2FeCo_block_hoh", "Example_surface_fccCu001_block_hoh"]


@pytest.mark.parametrize("name", LDOS_CASES)
def test_density_of_states_without_g0(name, tmp_path):
    """A flow that stops at the densities of states (recur_b -> zsqr -> block_green -> calculate_fermi: calculation.f90:700-712,
    self.f90:821-833) with `bands_gpu` + `green_gpu%defer_g0`: dtot / dosia / dosial come from rsrec_block_ldos on the coefficients
    the recursion left on the device, and g0 (13 MB per site) is never produced.  Checked against the same flow with the reference's
    own host reduction over a downloaded g0: the three files the reference writes must agree to the printed digits."""
    require_built(EXE)
    case = MANIFEST[name]
    outs = {}
    # (the device run also brings up the library's own communicator through a file -- one rank: the box has one GPU)
    for mode, env in (("device", {"RSREC_LDOS_ONLY": "1", "RSREC_DEFER_G0": "1", "RSREC_RANK": "0", "RSREC_NRANKS": "1", "RSREC_COMM_FILE": str(tmp_path / "comm.id")}),
                      ("host", {"RSREC_LDOS_ONLY": "1", "RSREC_HOST_LDOS": "1", "RSREC_HOST_HAM": "1"})):
        work = tmp_path / mode
        shutil.copytree(os.path.join(SCF, case["inputs"]), work)
        inp = work / "input.nml"
        inp.write_text(patch_namelist(inp.read_text(), case["patch"]))
        r = run_with_unlimited_stack([EXE], cwd=work, env=dict(env, OMP_NUM_THREADS="8"), timeout=1500, scrub=False)
        log = r.stdout + r.stderr
        assert r.returncode == 0 and "fatal" not in log.lower(), log[-3000:]
        if mode == "device":
            assert "ldos-only: device_ldos_calls=1 g0_pending=T" in log and "library communicator: rank 0 of 1" in log, log[-2000:]
            assert "ldos-gpu" in log and "bgreen-gpu" not in log, log[-3000:]            # no Green-function download in this flow
            assert "device_assemblies=0" not in log and "operator_arrays_from_device=0" not in log, log[-2000:]
        else:
            assert "ldos-only: device_ldos_calls=0 g0_pending=F" in log, log[-2000:]
            assert "bgreen-gpu" in log and "ldos-gpu" not in log, log[-3000:]
            assert "device_assemblies=0" in log and "operator_arrays_from_device=0" in log, log[-2000:]      # the reference's host build_bulkham / build_locham
        outs[mode] = {fn: [[fortran_float(v) for v in line.split()] for line in (work / fn).read_text().splitlines()]
                      for fn in sorted(os.listdir(work)) if fn == "totaldos.out" or fn.endswith("_dos.out")}
    assert set(outs["device"]) == set(outs["host"]) and len(outs["device"]) >= 3
    for fn, rows in outs["device"].items():
        ref = outs["host"][fn]
        assert len(rows) == len(ref) > 1000, fn
        worst = max(abs(a - b) for ra, rb in zip(rows, ref) for a, b in zip(ra, rb))
        assert worst <= 1.0e-5 + 1e-12, (fn, worst)                                      # files carry 5 decimals: last-digit rounding at most
