#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY (oracle side) -- generates tests/golden/*.npz.

Runs ONLY in the build container (needs /root/reference and oracle/_ref built by
oracle/build_ref.sh).  For every case it copies the reference test case's *input data files*
(input.nml + <label>.nml) into a scratch directory, patches the namelist the way the reference's
own test runner does (tests/run_test.py:79-83 applies the `namelists` dict of tests/scf/cases.json),
runs oracle/_ref/dump_fixture.x (the compiled reference + our dump driver) and stores the
recursion inputs and the reference's outputs at full precision as a compressed .npz.

Supercell cases (BASELINE.json configs 1/2) feed a synthetic periodic bcc lattice + the dumped
Fe stencil through oracle/_ref/ref_kernel.x (the compiled reference recursion routines).

usage: python oracle/make_fixtures.py [case ...]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import fixture_io as fio  # noqa: E402
from rslmtoasa_amd._proc import run_with_unlimited_stack  # noqa: E402

REF = os.environ.get("RSREC_REFERENCE", "/root/reference")
GOLD = os.path.join(ROOT, "tests", "golden")

# name -> (case dir under reference tests, namelist patch, extra)
CASES = {
    # tests/scf/cases.json "Example_bulk_bccFe_nsp2_block"
    "bccFe_nsp2_block": ("tests/scf/cases/bulk/bccFe", {"control": {"nsp": 2, "recur": "'block'", "lld": 20}, "hamiltonian": {"hoh": ".false."}}),
    # "Example_bulk_bccFe_nsp2_block_hoh"
    "bccFe_nsp2_block_hoh": ("tests/scf/cases/bulk/bccFe", {"control": {"nsp": 2, "recur": "'block'", "lld": 20}, "hamiltonian": {"hoh": ".true."}}),
    # non-collinear + SOC variant of the same case (cases.json nsp=4)
    "bccFe_nsp4_block": ("tests/scf/cases/bulk/bccFe", {"control": {"nsp": 4, "recur": "'block'", "lld": 12}, "hamiltonian": {"hoh": ".false."}}),
    # Chebyshev (cases.json uses lld=100, energy window -3..1.8; lld shortened to keep the file small)
    "bccFe_nsp2_cheb": ("tests/scf/cases/bulk/bccFe", {"control": {"nsp": 2, "recur": "'chebyshev'", "lld": 20}, "hamiltonian": {"hoh": ".false."}, "energy": {"energy_min": -3.0, "energy_max": 1.8}}),
    "bccFe_nsp2_cheb_hoh": ("tests/scf/cases/bulk/bccFe", {"control": {"nsp": 2, "recur": "'chebyshev'", "lld": 20}, "hamiltonian": {"hoh": ".true."}, "energy": {"energy_min": -3.0, "energy_max": 1.8}}),
    # the only scalar-Lanczos case of the reference (tests/regression/bccFe_lanczos: nsp=1, lld=16)
    "bccFe_nsp1_lanczos": ("tests/regression/bccFe_lanczos", {}),
    # impurity: per-atom hall blocks for the first nmax atoms
    "B2FeCo_block": ("tests/scf/cases/impurity/B2FeCo", {"control": {"nsp": 2, "recur": "'block'", "lld": 12}, "hamiltonian": {"hoh": ".false."}}),
    "B2FeCo_block_hoh": ("tests/scf/cases/impurity/B2FeCo", {"control": {"nsp": 2, "recur": "'block'", "lld": 12}, "hamiltonian": {"hoh": ".true."}}),
    # surface: 3 types, 2 recursion sites, fcc stencil (19 slots)
    "fccCu001_block_hoh": ("tests/scf/cases/surface/fccCu001", {"control": {"nsp": 2, "recur": "'block'", "lld": 12}, "hamiltonian": {"hoh": ".true."}}),
    "fccCu001_cheb": ("tests/scf/cases/surface/fccCu001", {"control": {"nsp": 2, "recur": "'chebyshev'", "lld": 12}, "hamiltonian": {"hoh": ".false."}, "energy": {"energy_min": -3.0, "energy_max": 1.8}}),
    # SURVEY C4 at its full depth (lld = 50 -> 102 moments); outputs only, the inputs are those of fccCu001_cheb
    "fccCu001_cheb50": ("tests/scf/cases/surface/fccCu001", {"control": {"nsp": 2, "recur": "'chebyshev'", "lld": 50}, "hamiltonian": {"hoh": ".false."}, "energy": {"energy_min": -3.0, "energy_max": 1.8}}),
    # scalar Haydock recursion beyond the one bulk case: three atom types (surface) and per-atom impurity blocks (nmax = 15)
    "fccCu001_nsp1_lanczos": ("tests/scf/cases/surface/fccCu001", {"control": {"nsp": 1, "recur": "'lanczos'", "lld": 12, "llsp": 12}, "hamiltonian": {"hoh": ".false."}}),
    "B2FeCo_nsp1_lanczos": ("tests/scf/cases/impurity/B2FeCo", {"control": {"nsp": 1, "recur": "'lanczos'", "lld": 12, "llsp": 12}, "hamiltonian": {"hoh": ".false."}}),
}
# non-collinear run with per-site spin frames (hamiltonian%local_axis = T, recursion.f90:1830-1832): four sites (Mn, Ga, Pt1, Pt2)
# whose moment directions are set to four different axes in the atoms' potential files (MOM_PATCH)
CASES["Pt2MnGa_nsp4_local_axis"] = ("tests/scf/cases/bulk/Pt2MnGa", {"control": {"nsp": 4, "recur": "'block'", "lld": 10}, "hamiltonian": {"hoh": ".false.", "local_axis": ".true."}})
CASES["Pt2MnGa_nsp4_local_axis_hoh"] = ("tests/scf/cases/bulk/Pt2MnGa", {"control": {"nsp": 4, "recur": "'block'", "lld": 10}, "hamiltonian": {"hoh": ".true.", "local_axis": ".true."}})
MOM_PATCH = {"Pt2MnGa_nsp4_local_axis": {"Mn.nml": (0.6, 0.0, 0.8), "Ga.nml": (0.0, 0.0, 1.0), "Pt1.nml": (0.0, 0.8, 0.6), "Pt2.nml": (-0.36, 0.48, 0.8)}}
MOM_PATCH["Pt2MnGa_nsp4_local_axis_hoh"] = MOM_PATCH["Pt2MnGa_nsp4_local_axis"]
OUTPUTS_ONLY = {"fccCu001_cheb50": "fccCu001_cheb"}      # name -> fixture that holds the (identical) inputs
# Green-function-only variants: same recursion inputs as the base case (tests pair them with <base>.npz), other terminator options
GREEN_ONLY = {
    # control%sym_term = .true.: orbital-independent terminator (green.f90:1263-1275)
    "bccFe_nsp2_block_symterm": ("tests/scf/cases/bulk/bccFe", {"control": {"nsp": 2, "recur": "'block'", "lld": 20, "sym_term": ".true."}, "hamiltonian": {"hoh": ".false."}}),
}


def patch_namelist(text, patch):
    """Minimal stand-in for f90nml.patch: set `key = value` inside `&group ... /` (add if absent)."""
    for group, kv in patch.items():
        m = re.search(r"(?ims)^\s*&%s\b(.*?)^\s*/" % re.escape(group), text)
        if not m:
            body = "".join("%s = %s\n" % (k, v) for k, v in kv.items())
            text += "\n&%s\n%s/\n" % (group, body)
            continue
        body = m.group(1)
        for k, v in kv.items():
            pat = re.compile(r"(?im)^(\s*%s\s*=\s*)[^!\n]*" % re.escape(k))
            if pat.search(body):
                body = pat.sub(lambda mm: mm.group(1) + str(v) + " ", body, count=1)
            else:
                body = body.rstrip("\n") + "\n%s = %s\n" % (k, v)
        text = text[:m.start(1)] + body + text[m.end(1):]
    return text


def run_ref(exe, cwd, threads=8, timeout=None):
    """One of the compiled-reference programs of oracle/_ref, with the unlimited stack its automatic arrays need (no shell hop)."""
    return run_with_unlimited_stack([exe], cwd=cwd, env={"OMP_NUM_THREADS": str(threads)}, timeout=timeout)


def run_case(name):
    case_dir, patch = CASES[name]
    scratch = tempfile.mkdtemp(prefix="rsrec_fx_%s_" % name)
    try:
        for fn in os.listdir(os.path.join(REF, case_dir)):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(REF, case_dir, fn), os.path.join(scratch, fn))
                os.chmod(os.path.join(scratch, fn), 0o644)
        p = os.path.join(scratch, "input.nml")
        txt = patch_namelist(open(p).read(), patch)
        open(p, "w").write(txt)
        for fn, mom in MOM_PATCH.get(name, {}).items():
            q = os.path.join(scratch, fn)
            t = re.sub(r"(?im)^(\s*mom\s*=\s*)[^\n]*", lambda mm: mm.group(1) + "%.16g, %.16g, %.16g" % mom, open(q).read(), count=1)
            open(q, "w").write(t)
        r = run_ref(os.path.join(HERE, "_ref", "dump_fixture.x"), scratch)
        if r.returncode != 0 or not os.path.exists(os.path.join(scratch, "fixture.bin")):
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("dump_fixture failed for " + name)
        d = fio.read_fixture_bin(os.path.join(scratch, "fixture.bin"))
        extra = {"source_case": np.array(case_dir), "namelist_patch": np.array(repr(patch))}
        la = os.path.join(scratch, "local_axis.bin")
        if os.path.exists(la):
            # global-frame blocks replace the (last site's frame) arrays of fixture.bin; moments and rotation matrices are added
            import struct
            with open(la, "rb") as f:
                magic, nrec = struct.unpack("<ii", f.read(8))
                assert magic == 0x4c415831 and nrec == d["nrec"]
                d["ee"] = fio._rd(f, np.complex128, d["ee"].shape)
                if d["hoh"]:
                    d["eeo"] = fio._rd(f, np.complex128, d["eeo"].shape)
                    d["enim"] = fio._rd(f, np.complex128, d["enim"].shape)
                if d["nmax"] > 0:
                    d["hall"] = fio._rd(f, np.complex128, d["hall"].shape)
                    if d["hoh"]:
                        d["hallo"] = fio._rd(f, np.complex128, d["hallo"].shape)
                moms, rots = [], []
                for _ in range(nrec):
                    moms.append(fio._rd(f, np.float64, (3,)))
                    rots.append(fio._rd(f, np.complex128, (18, 18)))
                assert f.read(1) == b""
            extra.update(local_axis=np.array(1), mom=np.stack(moms, axis=1), rot=np.stack(rots, axis=2), mom_patch=np.array(repr(MOM_PATCH.get(name))))
            d.pop("green", None)
        if name == "bccFe_nsp2_block":
            extra["slot_vec"] = slot_vectors(d)
        if name in OUTPUTS_ONLY:
            base = fio.load_golden(os.path.join(GOLD, OUTPUTS_ONLY[name] + ".npz"))
            for k in fio.INPUT_KEYS:
                if k in base:
                    assert np.array_equal(base[k], d[k]), "inputs of %s differ from %s: %s" % (name, OUTPUTS_ONLY[name], k)
            d = {k: v for k, v in d.items() if k not in fio.INPUT_KEYS}
            extra["inputs_from"] = np.array(OUTPUTS_ONLY[name])
        fio.save_golden(os.path.join(GOLD, name + ".npz"), d, extra)
        print("%-24s kk=%d nmax=%d nrec=%d lld=%d kind=%d hoh=%d -> %.1f KB" % (
            name, d["kk"], d["nmax"], d["nrec"], d["lld"], d["kind"], d["hoh"],
            os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1024))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


DENSITY_CASES = ["bccFe_nsp1_lanczos", "fccCu001_nsp1_lanczos", "B2FeCo_nsp1_lanczos"]
DENSITY_STRIDE = 25        # every 25th energy of the reference's mesh is kept


def run_density_case(name):
    """<name>_density.npz: inputs and the reference's outputs of the stage behind the scalar recursion of the same run as <name>.npz --
    dos%density (density_of_states.f90:248-363, bprldos :370-404) for every (site, direction) and the g0 green%sgreen makes of it
    (green.f90:628-705) -- on a sub-sampled set of energies (every energy is an independent continued fraction)."""
    case_dir, patch = CASES[name]
    scratch = tempfile.mkdtemp(prefix="rsrec_dn_%s_" % name)
    try:
        for fn in os.listdir(os.path.join(REF, case_dir)):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(REF, case_dir, fn), os.path.join(scratch, fn))
                os.chmod(os.path.join(scratch, fn), 0o644)
        p = os.path.join(scratch, "input.nml")
        txt = patch_namelist(open(p).read(), patch)
        open(p, "w").write(txt)
        r = run_ref(os.path.join(HERE, "_ref", "dump_fixture.x"), scratch)
        if r.returncode != 0:
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("dump_fixture failed for " + name)
        d = fio.read_fixture_bin(os.path.join(scratch, "fixture.bin"))
        g = d["density"]
        idx = np.arange(0, g["nen"], DENSITY_STRIDE, dtype=np.int32)
        out = dict(lld=d["lld"], llmax=d["llmax"], nrec=d["nrec"], nmdir=g["nmdir"], nen_full=g["nen"], ene_idx=idx, ene=g["ene"][idx],
                   dw_l=g["dw_l"], cshi=g["cshi"], a=g["a"], b2=g["b2"], tdens=g["tdens"][:, idx], g0=g["g0"][:, :, idx, :],
                   source_case=np.array(case_dir), namelist_patch=np.array(repr(patch)))
        path = os.path.join(GOLD, name + "_density.npz")
        np.savez_compressed(path, **out)
        print("%-24s density: nen=%d kept=%d nrec=%d nmdir=%d llmax=%d -> %.1f KB" % (name, g["nen"], len(idx), d["nrec"], g["nmdir"], d["llmax"], os.path.getsize(path) / 1024))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


GREEN_CASES = ["bccFe_nsp2_block", "bccFe_nsp4_block", "B2FeCo_block_hoh", "fccCu001_block_hoh", "bccFe_nsp2_cheb", "fccCu001_cheb"]
GREEN_STRIDE = 40          # every 40th energy of the reference's mesh (channels_ldos + 10 points) is kept


def run_green_case(name):
    """<name>_green.npz: inputs and the reference's output of green%block_green for the same run as <name>.npz:
    a_b (reference coefficients), sqrt(B^2) after zsqr, the terminator (get_terminf), the energy mesh and g0 on a
    sub-sampled set of energies (every energy is an independent continued fraction, green.f90:1257-1336)."""
    case_dir, patch = CASES[name] if name in CASES else GREEN_ONLY[name]
    scratch = tempfile.mkdtemp(prefix="rsrec_gx_%s_" % name)
    try:
        for fn in os.listdir(os.path.join(REF, case_dir)):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(REF, case_dir, fn), os.path.join(scratch, fn))
                os.chmod(os.path.join(scratch, fn), 0o644)
        p = os.path.join(scratch, "input.nml")
        txt = patch_namelist(open(p).read(), patch)
        open(p, "w").write(txt)
        r = run_ref(os.path.join(HERE, "_ref", "dump_fixture.x"), scratch)
        if r.returncode != 0:
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("dump_fixture failed for " + name)
        d = fio.read_fixture_bin(os.path.join(scratch, "fixture.bin"))
        g = d["green"]
        idx = np.arange(0, g["nen"], GREEN_STRIDE, dtype=np.int32)
        if d["kind"] == fio.KIND_CHEB:
            out = dict(lld=d["lld"], nrec=d["nrec"], nen_full=g["nen"], ene_idx=idx, ene=g["ene"][idx], emin=d["emin"], emax=d["emax"],
                       mu_n=d["mu_n"], g0=g["g0"][:, :, idx, :], source_case=np.array(case_dir), namelist_patch=np.array(repr(patch)))
            path = os.path.join(GOLD, name + "_green.npz")
            np.savez_compressed(path, **out)
            print("%-24s chebyshev_green: nen=%d kept=%d nrec=%d lld=%d -> %.1f KB" % (name, g["nen"], len(idx), d["nrec"], d["lld"], os.path.getsize(path) / 1024))
            return
        out = dict(lld=d["lld"], nrec=d["nrec"], nen_full=g["nen"], sym_term=g["sym_term"], ene_idx=idx, ene=g["ene"][idx],
                   ene_full_first=g["ene"][0], ene_full_step=g["ene"][1] - g["ene"][0],
                   a_inf=g["a_inf"], b_inf=g["b_inf"], a_b=d["a_b"], b_sqrt=g["b_sqrt"], g0=g["g0"][:, :, idx, :],
                   source_case=np.array(case_dir), namelist_patch=np.array(repr(patch)))
        if "eta" in g:      # block_green_eta: 1-based energy points, complex increments, g(18,18,neta,nrec)
            out.update(eta_points=g["eta_points"], eta=g["eta"], g_eta=g["g_eta"], ene_eta=g["ene"][g["eta_points"] - 1])
        path = os.path.join(GOLD, name + "_green.npz")
        np.savez_compressed(path, **out)
        print("%-24s green: nen=%d kept=%d nrec=%d lld=%d sym_term=%d -> %.1f KB" % (name, g["nen"], len(idx), d["nrec"], d["lld"], g["sym_term"],
                                                                                   os.path.getsize(path) / 1024))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


def slot_vectors(d):
    """Displacement vector (units of alat) of every neighbour slot, from an interior atom
    (slot m is the same displacement for every atom of a type: lattice.f90:2823-2893)."""
    nn, cr = d["nn"], d["cr"]
    nb = nn[0, 0]
    i = next(i for i in range(d["kk"]) if nn[i, 0] == nb and np.all(nn[i, 1:nb] > 0))
    v = np.zeros((nb, 3))
    for m in range(1, nb):
        v[m] = cr[:, nn[i, m] - 1] - cr[:, i]
    return v


def supercell_case(name, dims, lld, kind, hoh, nsites, stencil="bccFe_nsp2_block", threads=8, save_inputs=False, pairs=None):
    """Reference recursion routines (ref_kernel.x) on a synthetic periodic bcc supercell."""
    from rslmtoasa_amd.lattice import bcc_supercell, spread_sites
    st = fio.load_golden(os.path.join(GOLD, stencil + ".npz"))
    if hoh:
        sth = fio.load_golden(os.path.join(GOLD, stencil + "_hoh.npz"))
    nn = bcc_supercell(dims, st["slot_vec"])
    kk = nn.shape[0]
    irec = spread_sites(kk, nsites) if pairs is None else np.asarray(pairs, dtype=np.int32).ravel()   # pair variants: (i1,j1,i2,j2,...)
    p = dict(nn=nn, iz=np.ones(kk, np.int32), irec=irec, lld=lld, nsp=2, hoh=hoh, kind=kind,
             ee=st["ee"], lsham=st["lsham"], emin=-3.0, emax=1.8)
    if hoh:
        p.update(ee=sth["ee"], lsham=sth["lsham"], eeo=sth["eeo"], enim=sth["enim"])
    scratch = tempfile.mkdtemp(prefix="rsrec_sc_%s_" % name)
    try:
        fio.write_kernel_in(os.path.join(scratch, "kernel_in.bin"), p)
        r = run_ref(os.path.join(HERE, "_ref", "ref_kernel.x"), scratch, threads)
        if r.returncode != 0:
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("ref_kernel failed for " + name)
        open(os.path.join(GOLD, name + ".timer.txt"), "w").write(
            "# g_timer report of the compiled reference (oracle/_ref/ref_kernel.x), %d OpenMP threads, build container\n" % threads + r.stdout)
        out = fio.read_kernel_out(os.path.join(scratch, "kernel_out.bin"), lld, len(irec))
        meta = dict(dims=np.array(dims), lld=lld, kind=kind, hoh=int(hoh), nsp=2, irec=p["irec"], stencil=np.array(stencil),
                    emin=-3.0, emax=1.8)
        if pairs is not None:
            meta.update(pairs=np.asarray(pairs, dtype=np.int32))
        if kind in (fio.KIND_CHEB, fio.KIND_CHEB_IJ):
            meta.update(acheb=(1.8 + 3.0) / float(np.float32(2) - np.float32(0.3)), bcheb=(1.8 - 3.0) / 2)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **meta, **out)
        print("%-24s kk=%d lld=%d nsites=%d -> %.1f KB" % (name, kk, lld, nsites, os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1024))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


SUPERCELLS = {
    # BASELINE.json configs[0]: ~128-atom supercell, LL=30 (pbc 4x4x8)
    "sc_4x4x8_block": dict(dims=(4, 4, 8), lld=30, kind=fio.KIND_BLOCK, hoh=False, nsites=2),
    "sc_4x4x8_block_hoh": dict(dims=(4, 4, 8), lld=30, kind=fio.KIND_BLOCK, hoh=True, nsites=1),
    "sc_4x4x8_cheb": dict(dims=(4, 4, 8), lld=30, kind=fio.KIND_CHEB, hoh=False, nsites=1),
    # BASELINE.json configs[1]: 22^3 = 10 648 atoms, LL=50 (one site; outputs only, inputs are regenerated)
    "sc_22_block": dict(dims=(22, 22, 22), lld=50, kind=fio.KIND_BLOCK, hoh=False, nsites=1),
    # pair variants used by the exchange post-processing (recur_b_ij :1655, chebyshev_recur_ij :2376): first neighbours, i == j, distant pair
    "sc_4x4x8_block_ij": dict(dims=(4, 4, 8), lld=12, kind=fio.KIND_BLOCK_IJ, hoh=False, nsites=0, pairs=[(1, 2), (7, 7), (5, 40)]),
    "sc_4x4x8_cheb_ij": dict(dims=(4, 4, 8), lld=12, kind=fio.KIND_CHEB_IJ, hoh=False, nsites=0, pairs=[(1, 2), (7, 7), (5, 40)]),
    "sc_4x4x8_cheb_ij_hoh": dict(dims=(4, 4, 8), lld=12, kind=fio.KIND_CHEB_IJ, hoh=True, nsites=0, pairs=[(1, 2), (7, 7)]),
}


KUBO_CASES = {
    # tests/postproc/cases.json "Example_exchange_conductivity_fccPt" / "_hoh" (cond_type = 'spin', per_type), cell and moment count
    # reduced so that the fixture stays small: n = 8^3 fcc cell, cond_ll = 10 (the case runs 20^3, cond_ll = 50)
    "fccPt_kubo": ("tests/postproc/cases/conductivity/fccPt", {"lattice": {"n1": 8, "n2": 8, "n3": 8}, "control": {"cond_ll": 10, "lld": 10}, "hamiltonian": {"hoh": ".false."}}),
    "fccPt_kubo_hoh": ("tests/postproc/cases/conductivity/fccPt", {"lattice": {"n1": 8, "n2": 8, "n3": 8}, "control": {"cond_ll": 10, "lld": 10}, "hamiltonian": {"hoh": ".true."}}),
    # cond_calctype = 'random_vec' (recursion.f90:1101-1140; round 4): two random-phase vectors over all atoms.  The fixture carries the
    # random numbers the routine drew (dump_kubo.f90 version 3) next to its moments.
    "fccPt_kubo_random": ("tests/postproc/cases/conductivity/fccPt", {"lattice": {"n1": 8, "n2": 8, "n3": 8},
                                                                      "control": {"cond_ll": 6, "lld": 6, "cond_calctype": "'random_vec'", "random_vec_num": 2},
                                                                      "hamiltonian": {"hoh": ".false."}}),
}


def run_kubo_case(name):
    """<name>.npz: inputs and output of recursion%compute_moments_stochastic (recursion.f90:979) from the compiled reference
    (oracle/_ref/dump_kubo.x = the reference modules + oracle/dump_kubo.f90, which replays calculation.f90:960-1052)."""
    import struct
    case_dir, patch = KUBO_CASES[name]
    scratch = tempfile.mkdtemp(prefix="rsrec_kubo_%s_" % name)
    try:
        for fn in os.listdir(os.path.join(REF, case_dir)):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(REF, case_dir, fn), os.path.join(scratch, fn))
                os.chmod(os.path.join(scratch, fn), 0o644)
        p = os.path.join(scratch, "input.nml")
        txt = open(p).read()
        # the case file holds TWO &hamiltonian groups; a namelist read takes the first one, so the second (hoh only) is dropped and
        # the patch goes into the first
        head, sep, tail = txt.rpartition("&hamiltonian")
        if "&hamiltonian" in head:
            txt = head + tail[tail.index("/") + 1:]
        # the case file still carries two keys the reference's namelists no longer declare (js_alpha, cond_type): the read of the
        # group stops there with an error the reference only logs, silently dropping every later key.  They are removed here so
        # that the whole group (hoh, cond_calctype) is read.
        txt = "\n".join(l for l in txt.splitlines() if not re.match(r"\s*(js_alpha|cond_type)\s*=", l)) + "\n"
        txt = patch_namelist(txt, patch)
        open(p, "w").write(txt)
        r = run_ref(os.path.join(HERE, "_ref", "dump_kubo.x"), scratch)
        if r.returncode != 0 or not os.path.exists(os.path.join(scratch, "kubo.bin")):
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("dump_kubo failed for " + name)
        with open(os.path.join(scratch, "kubo.bin"), "rb") as f:
            magic, version = struct.unpack("<ii", f.read(8))
            assert magic == 0x4b55424f
            kk, nncols, nmax, ntype, cond_ll, nsp, hoh, nslots, nvec = struct.unpack("<9i", f.read(36))
            a, b = struct.unpack("<2d", f.read(16))
            rd = fio._rd
            d = dict(kk=kk, nmax=nmax, ntype=ntype, cond_ll=cond_ll, nsp=nsp, hoh=hoh, nslots=nslots, acheb=a, bcheb=b)
            d["iz"] = rd(f, np.int32, (kk,)); d["nn"] = rd(f, np.int32, (kk, nncols)); d["atlist"] = rd(f, np.int32, (ntype,))
            for k, shape in (("ee", (18, 18, nslots, ntype)), ("lsham", (18, 18, ntype)), ("eeo", (18, 18, nslots, ntype)), ("enim", (18, 18, ntype)),
                             ("v_a", (18, 18, nslots, ntype)), ("v_b", (18, 18, nslots, ntype)), ("vo_a", (18, 18, nslots, ntype)), ("vo_b", (18, 18, nslots, ntype))):
                d[k] = rd(f, np.complex128, shape)
            d["mu_nm"] = rd(f, np.complex128, (18, 18, cond_ll, cond_ll, nvec))
            if version >= 2:            # positions: only to name the displacement vector of every neighbour slot (supercell generator)
                d["cr"] = rd(f, np.float64, (3, kk))
                alat = struct.unpack("<d", f.read(8))[0]
                d["slot_vec"] = slot_vectors(d)          # from an interior atom of the (free) cluster
                d.pop("cr")
                d["alat"] = alat
            if version >= 3:            # random_vec: the random number of every (atom, vector)
                d["rng"] = rd(f, np.float64, (kk, nvec))
            assert f.read(1) == b""
        if not hoh:
            for k in ("eeo", "enim", "vo_a", "vo_b"):
                d.pop(k)
        d["source_case"] = np.array(case_dir); d["namelist_patch"] = np.array(repr(patch))
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, **d)
        wall = [l for l in r.stdout.splitlines() if "wall time" in l]
        print("%-24s kk=%d nslots=%d cond_ll=%d hoh=%d |mu|max=%.3e -> %.1f KB  (%s)" % (name, kk, nslots, cond_ll, hoh, np.abs(d["mu_nm"]).max(),
                                                                                   os.path.getsize(path) / 1024, wall[-1].strip() if wall else ""))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


ORBITAL_CASES = {
    # chebyshev_orbital_mod loops over ALL atoms as seeds (kk x lld whole-lattice products): a small free fcc Pt cluster
    "fccPt_orbital": ("tests/postproc/cases/conductivity/fccPt", {"lattice": {"n1": 5, "n2": 5, "n3": 5}, "control": {"lld": 8}, "hamiltonian": {"hoh": ".false."}}),
    "fccPt_orbital_hoh": ("tests/postproc/cases/conductivity/fccPt", {"lattice": {"n1": 5, "n2": 5, "n3": 5}, "control": {"lld": 8}, "hamiltonian": {"hoh": ".true."}}),
    # a MAGNETIC case (round 4): ferromagnetic bcc Fe with spin-orbit coupling (nsp = 2, l.s on site), a 748-atom free cluster of the
    # reference's bulk/bccFe case: the orbital moment does not vanish, unit 50 carries 7 significant digits of the trace of the moments
    # (the non-magnetic fcc Pt cases above leave 1e-13 noise there)
    "bccFe_orbital": ("tests/scf/cases/bulk/bccFe", {"lattice": {"rc": "20"}, "control": {"lld": 8}}),
}


def run_orbital_case(name):
    """<name>.npz: inputs of recursion%chebyshev_orbital_mod (recursion.f90:2834) and what the compiled reference writes: unit 50
    (`fort.50`: E - E_F, -Lz(E)/pi integrated, -lz(E)/pi energy-resolved; 3es16.6 = 7 significant digits) and, per seed atom, the
    list-directed sums it prints (sum(left_vec), sum(psiref), sum(left_vec1), sum(left_vec2): full precision).  The moments
    themselves are a local variable of the routine and never leave it."""
    import struct
    case_dir, patch = ORBITAL_CASES[name]
    scratch = tempfile.mkdtemp(prefix="rsrec_orb_%s_" % name)
    try:
        for fn in os.listdir(os.path.join(REF, case_dir)):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(REF, case_dir, fn), os.path.join(scratch, fn))
                os.chmod(os.path.join(scratch, fn), 0o644)
        p = os.path.join(scratch, "input.nml")
        txt = open(p).read()
        head, sep, tail = txt.rpartition("&hamiltonian")          # (see run_kubo_case: two &hamiltonian groups, undeclared keys)
        if "&hamiltonian" in head:
            txt = head + tail[tail.index("/") + 1:]
        txt = "\n".join(l for l in txt.splitlines() if not re.match(r"\s*(js_alpha|cond_type)\s*=", l)) + "\n"
        txt = patch_namelist(txt, patch)
        open(p, "w").write(txt)
        r = run_with_unlimited_stack([os.path.join(HERE, "_ref", "dump_kubo.x")], cwd=scratch, env={"OMP_NUM_THREADS": "8", "RSREC_DUMP_ORBITAL": "1"})
        if r.returncode != 0 or not os.path.exists(os.path.join(scratch, "orbital.bin")):
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("dump_kubo (orbital) failed for " + name)
        with open(os.path.join(scratch, "orbital.bin"), "rb") as f:
            magic, version = struct.unpack("<ii", f.read(8))
            assert magic == 0x4f52424d
            kk, nncols, nmax, ntype, lld, nsp, hoh, nslots, nv, nv1 = struct.unpack("<10i", f.read(40))
            a, b, alat, fermi = struct.unpack("<4d", f.read(32))
            rd = fio._rd
            d = dict(kk=kk, nmax=nmax, ntype=ntype, lld=lld, nsp=nsp, hoh=hoh, nslots=nslots, acheb=a, bcheb=b, alat=alat, fermi=fermi, nv1=nv1)
            d["iz"] = rd(f, np.int32, (kk,)); d["nn"] = rd(f, np.int32, (kk, nncols))
            for k, shape in (("ee", (18, 18, nslots, ntype)), ("lsham", (18, 18, ntype)), ("eeo", (18, 18, nslots, ntype)), ("enim", (18, 18, ntype))):
                d[k] = rd(f, np.complex128, shape)
            d["cr"] = rd(f, np.float64, (3, kk))
            d["ene"] = rd(f, np.float64, (nv,))
            assert f.read(1) == b""
        if not hoh:
            d.pop("eeo"); d.pop("enim")
        rows = [[float(v) for v in l.split()] for l in open(os.path.join(scratch, "fort.50")).read().splitlines() if l.strip()]
        d["fort50"] = np.array(rows)
        assert d["fort50"].shape == (nv, 3)
        # per-seed sums: list-directed complex numbers "(re,im)", four per seed, possibly wrapped over lines
        flat = re.sub(r"\s+", "", r.stdout)
        c = re.findall(r"\(([-+0-9.eEdD]+),([-+0-9.eEdD]+)\)", flat)
        if len(c) == 4 * kk:
            d["seed_sums"] = np.array([complex(float(x.replace("D", "E")), float(y.replace("D", "E"))) for x, y in c]).reshape(kk, 4)
        d["source_case"] = np.array(case_dir); d["namelist_patch"] = np.array(repr(patch))
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, **d)
        wall = [l for l in r.stdout.splitlines() if "wall time" in l]
        print("%-24s kk=%d lld=%d hoh=%d nv=%d seed_sums=%s |lz|max=%.3e -> %.1f KB  (%s)" % (name, kk, lld, hoh, nv, "seed_sums" in d, np.abs(d["fort50"][:, 2]).max(),
                                                                                      os.path.getsize(path) / 1024, wall[-1].strip() if wall else ""))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


HMAG_CASES = ["bccFe_nsp2_block_hoh", "bccFe_nsp4_block", "fccCu001_block_hoh", "B2FeCo_block_hoh"]


def run_hmag_case(name):
    """<name>_hmag.npz: the INPUTS of the reference's build_bulkham / build_locham (hamiltonian.f90:1553-1667) for the run of <name>.npz:
    the four 9x9 parts chbar_nc leaves in `hmag` per class atom, the atom type behind every neighbour slot, `obarm`.  The outputs they
    pin are the ee / eeo / hall / hallo arrays of <name>.npz (checked here to be the same run)."""
    import struct
    case_dir, patch = CASES[name]
    scratch = tempfile.mkdtemp(prefix="rsrec_hm_%s_" % name)
    try:
        for fn in os.listdir(os.path.join(REF, case_dir)):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(REF, case_dir, fn), os.path.join(scratch, fn))
                os.chmod(os.path.join(scratch, fn), 0o644)
        q = os.path.join(scratch, "input.nml")
        txt = patch_namelist(open(q).read(), patch)
        open(q, "w").write(txt)
        r = run_with_unlimited_stack([os.path.join(HERE, "_ref", "dump_fixture.x")], cwd=scratch, env={"OMP_NUM_THREADS": "8", "RSREC_DUMP_HMAG": "1"})
        if r.returncode != 0 or not os.path.exists(os.path.join(scratch, "hmag.bin")):
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("dump_fixture (hmag) failed for " + name)
        d = fio.read_fixture_bin(os.path.join(scratch, "fixture.bin"))
        base = fio.load_golden(os.path.join(GOLD, name + ".npz"))
        for k in ("ee", "eeo", "hall", "hallo"):
            if k in base:
                assert np.array_equal(base[k], d[k]), "%s of this run differs from %s.npz" % (k, name)
        with open(os.path.join(scratch, "hmag.bin"), "rb") as f:
            magic, ntype, nmax, nsl, hoh = struct.unpack("<5i", f.read(20))
            assert magic == 0x484d4731 and ntype == d["ntype"] and nsl == d["ee"].shape[2]
            ncls = ntype + (nmax if d["nmax"] > 0 else 0)
            nr, ji, hm = [], [], []
            for _ in range(ncls):
                nr.append(struct.unpack("<i", f.read(4))[0])
                ji.append(fio._rd(f, np.int32, (nsl,)))
                hm.append(fio._rd(f, np.complex128, (9, 9, nsl, 4)))
            obarm = fio._rd(f, np.complex128, (18, 18, ntype))
            assert f.read(1) == b""
        out = {"hmag_type": np.stack(hm[:ntype], axis=4), "nbr_type_type": np.stack(ji[:ntype], axis=1), "nr_type": np.array(nr[:ntype], np.int32),
               "obarm": obarm, "hoh": np.array(hoh), "source_case": np.array(case_dir), "namelist_patch": np.array(repr(patch))}
        if ncls > ntype:
            out.update(hmag_atom=np.stack(hm[ntype:], axis=4), nbr_type_atom=np.stack(ji[ntype:], axis=1), nr_atom=np.array(nr[ntype:], np.int32))
        np.savez_compressed(os.path.join(GOLD, name + "_hmag.npz"), **out)
        print("%-24s ntype=%d nmax=%d nslots=%d -> %.1f KB" % (name + "_hmag", ntype, nmax if ncls > ntype else 0, nsl, os.path.getsize(os.path.join(GOLD, name + "_hmag.npz")) / 1024))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


POSITION_CASES = {"fccCu001_cr": "fccCu001_cheb", "B2FeCo_cr": "B2FeCo_block_hoh"}


def run_position_case(name):
    """<name>.npz: lattice%cr(3,kk) of the reference run behind the fixture POSITION_CASES[name] (the other fixtures drop it: the recursion's
    arithmetic never reads positions).  The Fortran drop-in hands cr to rsrec_set_positions as a locality hint; bench.py does the same for
    the two multi-class workloads with this file."""
    base_name = POSITION_CASES[name]
    case_dir, patch = CASES[base_name]
    scratch = tempfile.mkdtemp(prefix="rsrec_cr_%s_" % name)
    try:
        for fn in os.listdir(os.path.join(REF, case_dir)):
            if fn.endswith(".nml"):
                shutil.copy(os.path.join(REF, case_dir, fn), os.path.join(scratch, fn))
                os.chmod(os.path.join(scratch, fn), 0o644)
        q = os.path.join(scratch, "input.nml")
        txt = patch_namelist(open(q).read(), patch)
        open(q, "w").write(txt)
        r = run_ref(os.path.join(HERE, "_ref", "dump_fixture.x"), scratch)
        if r.returncode != 0 or not os.path.exists(os.path.join(scratch, "fixture.bin")):
            print(r.stdout[-3000:], r.stderr[-3000:])
            raise RuntimeError("dump_fixture failed for " + name)
        d = fio.read_fixture_bin(os.path.join(scratch, "fixture.bin"))
        base = fio.load_golden(os.path.join(GOLD, base_name + ".npz"))
        assert np.array_equal(base["nn"], d["nn"]) and np.array_equal(base["iz"], d["iz"]), "lattice of this run differs from %s.npz" % base_name
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), cr=d["cr"], lattice_of=np.array(base_name))
        print("%-24s kk=%d -> %.1f KB" % (name, d["kk"], os.path.getsize(os.path.join(GOLD, name + ".npz")) / 1024))
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


def spread_case(name, threads=(1, 2, 8)):
    """<name>_spread.npz: the compiled reference's OWN run-to-run spread on a supercell case -- the same ref_kernel.x run at
    several OpenMP thread counts (its reductions are `omp reduction` sums, recursion.f90:1638-1645: the summation order depends
    on the thread count).  Tests bound the GPU's deviation on ill-conditioned late levels by a multiple of this spread."""
    cfg = SUPERCELLS[name]
    outs = {}
    for t in threads:
        tmpname = "%s__t%d" % (name, t)
        supercell_case(tmpname, threads=t, **cfg)
        with np.load(os.path.join(GOLD, tmpname + ".npz"), allow_pickle=False) as z:
            for k in fio.OUTPUT_KEYS:
                if k in z.files:
                    outs["%s_t%d" % (k, t)] = z[k]
        os.remove(os.path.join(GOLD, tmpname + ".npz"))
        os.remove(os.path.join(GOLD, tmpname + ".timer.txt"))
    outs["threads"] = np.array(threads)
    np.savez_compressed(os.path.join(GOLD, name + "_spread.npz"), **outs)
    print("%-24s reference spread over threads %s" % (name + "_spread", threads))


# Seeds of the round-3 fuzz campaign (tools/fuzz_recursion.py) whose first verdict was FAIL and which the checker afterwards classified by
# the ORACLE's own conditioning (gpurun_out/fuzz1.log, fuzz3.log).  Round 4 gives them a judge that is neither the engine nor the oracle:
# the compiled reference, run at 1, 2 and 8 OpenMP threads (its `omp reduction` sums change order with the thread count,
# recursion.f90:1638-1645) -- the spread between those runs is what the reference itself can reproduce of these chains.
#   block: the lattice / operator / sites the seed gives the generator (same draws as the fuzzer's: everything up to `irec`);
#   pairs: the same, plus atom pairs -- the fuzzer drew them AFTER its option set, whose table has changed since, so the original
#          pairs are not recoverable; three pairs are drawn from default_rng(seed + 10**6) instead (one of them i == j).
FUZZ_SEEDS = {36: "block", 75: "block", 783: "block", 1511: "block", 1766: "block", 20187: "pairs", 20215: "pairs", 20441: "pairs"}


def fuzz_seed_case(seed, threads=(1, 2, 8)):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_spmm_random import random_problem
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
    mode = FUZZ_SEEDS[seed]
    irec = irec[:5]                                                    # at most five chains per seed (fixture size)
    runs = [(fio.KIND_BLOCK, irec)]
    d = dict(p, seed=seed, kk=kk, nslots=nslots, ntype=ntype, lld=lld, collinear=int(collinear), irec=irec[:5], threads=np.array(threads))
    if mode == "pairs":
        pr = np.random.default_rng(seed + 10 ** 6)
        pairs = pr.integers(1, kk + 1, (3, 2)).astype(np.int32)
        pairs[0, 1] = pairs[0, 0]
        d["pairs"] = pairs
        runs = [(fio.KIND_BLOCK_IJ, pairs.ravel())]
    # One reference process per (site | pair) and thread count: a chain whose Krylov space is exhausted ends the reference's run --
    # sqrt of a rounding-negative eigenvalue of B^2 (recursion.f90:1950) gives NaN, the next zheev fails and the routine calls
    # g_logger%fatal ('Diagonalization error', :1942) -- and must not take the other chains of the seed with it.
    # ok_k<kind>_t<threads>[unit] = 1 if that run completed.
    # Besides the thread counts (which only reorder the reductions -- and not even that on regions of a few atoms): three runs on inputs
    # perturbed at the rounding level, every block element times (1 + 1e-15 xi), xi uniform in (-1, 1).  What the reference's own answer
    # moves by under such a perturbation is its sensitivity to rounding, whatever the arithmetic order: cond x 1e-15.
    def perturbed(q, k):
        pr = np.random.default_rng(seed * 1000 + k)
        out = dict(q)
        for name in ("ee", "lsham", "eeo", "enim", "hall", "hallo"):
            if name in q:
                out[name] = np.asfortranarray(q[name] * (1.0 + 1e-15 * pr.uniform(-1.0, 1.0, q[name].shape)))
        return out
    variants = [("t%d" % t, t, 0) for t in threads] + [("p%d" % k, 8, k) for k in (1, 2, 3)]
    d["variants"] = np.array([v[0] for v in variants])
    for kind, sites in runs:
        per = 2 if kind == fio.KIND_BLOCK_IJ else 1                     # irec entries per unit (pair = two atoms -> four chains)
        nunit, nch = len(sites) // per, (4 if per == 2 else 1)
        for tag, t, pk in variants:
            ok = np.zeros(nunit, np.int32)
            acc = {}
            for u in range(nunit):
                q = dict(p, irec=np.asarray(sites[per * u:per * u + per], np.int32), lld=lld, kind=kind, emin=-60.0, emax=60.0)
                if pk:
                    q = perturbed(q, pk)
                scratch = tempfile.mkdtemp(prefix="rsrec_fuzz_%d_" % seed)
                try:
                    fio.write_kernel_in(os.path.join(scratch, "kernel_in.bin"), q)
                    r = run_ref(os.path.join(HERE, "_ref", "ref_kernel.x"), scratch, t)
                    if r.returncode != 0:
                        if "Diagonalization error" not in (r.stdout + r.stderr) and "did not converge" not in (r.stdout + r.stderr):
                            print(r.stdout[-2000:], r.stderr[-2000:])
                            raise RuntimeError("ref_kernel failed for fuzz seed %d in an unexpected way" % seed)
                        continue
                    out = fio.read_kernel_out(os.path.join(scratch, "kernel_out.bin"), lld, per)
                    ok[u] = 1
                    for k in ("a_b", "b2_b", "mu_n"):
                        if k in out:
                            full = acc.setdefault(k, np.full(out[k].shape[:3] + (nunit * nch,), np.nan + 0j, np.complex128, order="F"))
                            full[:, :, :, nch * u:nch * (u + 1)] = out[k][:, :, :, :nch]
                finally:
                    shutil.rmtree(scratch, ignore_errors=True)
            d["ok_" + tag] = ok
            for k, v in acc.items():
                d["%s_%s" % (k, tag)] = v
    # what travels: the 8-thread run as THE reference answer, and per (level, chain) the largest relative distance between two of the
    # runs (per 18x18 matrix, as tests/helpers.py level_errors measures it); inf where a run did not complete or is not finite
    def level_err(x, y):
        dd = np.abs(x - y).max(axis=(0, 1)); rr = np.abs(y).max(axis=(0, 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            e = np.where(rr > 0, dd / rr, np.where(dd > 0, np.inf, 0.0))
        return np.where(np.isfinite(e), e, np.inf)
    for k in ("a_b", "b2_b"):
        arrs = [d.pop("%s_%s" % (k, v[0]), None) for v in variants]
        shape = next(a.shape for a in arrs if a is not None) if any(a is not None for a in arrs) else None
        if shape is None:
            continue
        arrs = [a if a is not None else np.full(shape, np.nan + 0j) for a in arrs]
        spread = np.zeros(shape[2:])
        for i in range(len(arrs)):
            for j in range(i + 1, len(arrs)):
                spread = np.maximum(spread, np.maximum(level_err(arrs[i], arrs[j]), level_err(arrs[j], arrs[i])))
        d[k + "_ref"] = arrs[len(threads) - 1]                           # the unperturbed 8-thread run
        d[k + "_spread"] = spread
    path = os.path.join(GOLD, "fuzz_seed_%d.npz" % seed)
    np.savez_compressed(path, **d)
    print("fuzz_seed_%-6d %-6s kk=%d slots=%d types=%d nmax=%d hoh=%d collinear=%d sites=%d lld=%d -> %.1f KB" % (
        seed, mode, kk, nslots, ntype, nmax, hoh, collinear, nsites, lld, os.path.getsize(path) / 1024))


def krylov_exhaustion_case(name="krylov_2x2x2", dims=(2, 2, 2), llds=range(6, 15)):
    """What the reference does when the Krylov space runs out: block Lanczos from one site of an 8-atom periodic bcc cell (144
    orbitals = 8 blocks of 18: after level 8 nothing is left), for a range of depths.  Records, per depth, whether the compiled
    reference's run completed or ended in g_logger%fatal('Diagonalization error') (recursion.f90:1942), and the coefficients of the
    deepest run that completed.  (The 128-atom cell of BASELINE config 0 never gets there: run to LL = 140 the reference loses
    orthogonality long before level 128 and simply carries on.)"""
    from rslmtoasa_amd.lattice import bcc_supercell
    st = fio.load_golden(os.path.join(GOLD, "bccFe_nsp2_block.npz"))
    nn = bcc_supercell(dims, st["slot_vec"])
    kk = nn.shape[0]
    ok, msgs, best = [], [], None
    for lld in llds:
        p = dict(nn=nn, iz=np.ones(kk, np.int32), irec=np.array([1], np.int32), lld=lld, nsp=2, hoh=0, kind=fio.KIND_BLOCK, ee=st["ee"], lsham=st["lsham"], emin=-3.0, emax=1.8)
        scratch = tempfile.mkdtemp(prefix="rsrec_krylov_")
        try:
            fio.write_kernel_in(os.path.join(scratch, "kernel_in.bin"), p)
            r = run_ref(os.path.join(HERE, "_ref", "ref_kernel.x"), scratch, 8)
            fatal = "Diagonalization error" in (r.stdout + r.stderr)
            assert r.returncode == 0 or fatal, (r.stdout + r.stderr)[-2000:]
            ok.append(int(r.returncode == 0))
            if r.returncode == 0:
                best = (lld, fio.read_kernel_out(os.path.join(scratch, "kernel_out.bin"), lld, 1))
        finally:
            shutil.rmtree(scratch, ignore_errors=True)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), dims=np.array(dims), kk=kk, llds=np.array(list(llds)), ok=np.array(ok), lld_best=best[0],
                        a_b=best[1]["a_b"], b2_b=best[1]["b2_b"], stencil=np.array("bccFe_nsp2_block"))
    print("%-24s kk=%d depths %s completed %s" % (name, kk, list(llds), ok))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    want = sys.argv[1:] or (list(CASES) + list(SUPERCELLS) + [c + "_green" for c in GREEN_CASES + list(GREEN_ONLY)] + [c + "_density" for c in DENSITY_CASES] + list(KUBO_CASES) + list(ORBITAL_CASES) + [c + "_hmag" for c in HMAG_CASES] + list(POSITION_CASES)
                            + ["sc_4x4x8_block_spread", "sc_4x4x8_block_hoh_spread"] + ["fuzz_seed_%d" % q for q in FUZZ_SEEDS] + ["krylov_2x2x2"])
    for n in want:
        if n == "krylov_2x2x2":
            krylov_exhaustion_case()
        elif n.startswith("fuzz_seed_"):
            fuzz_seed_case(int(n[len("fuzz_seed_"):]))
        elif n in KUBO_CASES:
            run_kubo_case(n)
        elif n in ORBITAL_CASES:
            run_orbital_case(n)
        elif n in POSITION_CASES:
            run_position_case(n)
        elif n.endswith("_hmag"):
            run_hmag_case(n[:-len("_hmag")])
        elif n.endswith("_spread"):
            spread_case(n[:-len("_spread")])
        elif n.endswith("_green"):
            run_green_case(n[:-len("_green")])
        elif n.endswith("_density"):
            run_density_case(n[:-len("_density")])
        elif n in CASES:
            run_case(n)
        else:
            supercell_case(n, **SUPERCELLS[n])
