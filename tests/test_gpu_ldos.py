"""The LDOS stage on the device (SURVEY.md 8 a10 + a13): terminator kernel and the resident-coefficient pipeline
recursion -> zsqr -> get_terminf -> bgreen -> calculate_fermi's reduction (rsrec_terminator, rsrec_block_ldos).

Checker: the compiled reference's own outputs held by tests/golden/*_green.npz (a_inf, b_inf from recursion%get_terminf,
g0 from green%block_green on every 40th energy of its mesh).  dtot / dosia / dosial are formed from that g0 with the
expressions of bands.f90:258-268."""
import numpy as np
import pytest

from helpers import RTOL, load_golden, objects_from, problem_dict
from rslmtoasa_amd.green import Green
from rslmtoasa_amd.recursion import Recursion
from test_gpu_green import GREEN_CASES, base_case, load_green

pytestmark = pytest.mark.gpu


def ldos_from_g0(g0):
    """bands.f90:258-268 on a g0(18,18,nen,nsites) array: dosial(ia,j,i), dosia(ia,i), dtot(i)."""
    d = np.arange(18)
    gim = g0[d, d].imag                                    # (18, nen, nsites)
    dosial = (-gim / np.pi).transpose(2, 0, 1)             # (nsites, 18, nen)
    dosia = (-(gim[:9] + gim[9:]) / np.pi).sum(axis=0).T   # (nsites, nen)
    return dosia.sum(axis=0), dosia, dosial


@pytest.mark.parametrize("name", GREEN_CASES)
def test_terminator_matches_reference(name):
    """get_terminf / get_cinf / bpopt / emami on the GPU: the reference's coefficients in, the reference's a_inf, b_inf out.
    The routine is a chain of bisections -- every comparison must fall the same way, so agreement is to rounding or not at all."""
    z = load_green(name)
    g = load_golden(base_case(name))
    rec = Recursion(*objects_from(problem_dict(g), g["irec"], g["lld"], nsp=g["nsp"]), device=0)
    n = int(z["nrec"])
    rec.a_b[:, :, :, :n] = z["a_b"]
    rec.b2_b[:, :, :, :n] = z["b_sqrt"]
    a_inf, b_inf, a0, b0 = Green(rec, z["ene"]).terminator(nsites=n)
    assert np.abs(a_inf - z["a_inf"]).max() <= 1e-13 * np.abs(z["a_inf"]).max()
    assert np.abs(b_inf - z["b_inf"]).max() <= 1e-13 * np.abs(z["b_inf"]).max()
    d = np.arange(18)
    assert np.allclose(a0, z["a_inf"][d, d].mean(axis=0), rtol=1e-13) and np.allclose(b0, z["b_inf"][d, d].mean(axis=0), rtol=1e-13)
    # the quirks of get_terminf (:2114-2131): no NaN survives, diagonal never zero
    assert np.isfinite(a_inf).all() and np.isfinite(b_inf).all() and np.all(a_inf[d, d] != 0) and np.all(b_inf[d, d] != 0)
    rec.close()


def test_terminator_many_sites_in_one_call(oracle_lib):
    """64 sites in one launch (the all-sites call of the drop-in's block_green): every site equals what a one-site call gives, bit
    for bit, and the CPU oracle's values.  Sites are the bulk bcc Fe coefficients with the levels scaled site by site, so that no
    two sites run the same bisections."""
    z = load_green("bccFe_nsp2_block")
    g = load_golden("bccFe_nsp2_block")
    ns = 64
    rec = Recursion(*objects_from(problem_dict(g), np.repeat(g["irec"], ns), g["lld"], nsp=g["nsp"]), device=0)
    scale = 1.0 + 0.003 * np.arange(ns)
    rec.a_b[:, :, :, :ns] = z["a_b"][:, :, :, :1] * scale
    rec.b2_b[:, :, :, :ns] = z["b_sqrt"][:, :, :, :1] * np.sqrt(scale)
    gr = Green(rec, z["ene"])
    a_all, b_all, a0, b0 = gr.terminator(nsites=ns)
    a_o, b_o, _, _ = oracle_lib.terminator(rec.a_b[:, :, :, :ns], rec.b2_b[:, :, :, :ns])
    assert np.abs(a_all - a_o).max() <= 1e-13 * np.abs(a_o).max() and np.abs(b_all - b_o).max() <= 1e-13 * np.abs(b_o).max()
    keep_a, keep_b = rec.a_b.copy(), rec.b2_b.copy()
    for s in (0, 17, 63):
        rec.a_b[:, :, :, :1] = keep_a[:, :, :, s:s + 1]
        rec.b2_b[:, :, :, :1] = keep_b[:, :, :, s:s + 1]
        a1, b1, _, _ = gr.terminator(nsites=1)
        assert np.array_equal(a1[:, :, 0], a_all[:, :, s]) and np.array_equal(b1[:, :, 0], b_all[:, :, s])
    rec.close()


@pytest.mark.parametrize("name", GREEN_CASES)
def test_resident_ldos_pipeline(name):
    """recur_b on the GPU, then ONE call for the whole LDOS stage from the coefficients left on the device: terminator and
    densities of states against the reference run (its coefficients differ from the GPU's by rounding, 1e-14)."""
    z = load_green(name)
    g = load_golden(base_case(name))
    rec = Recursion(*objects_from(problem_dict(g), g["irec"], g["lld"], nsp=g["nsp"]), device=0)
    rec.recur_b()
    n = int(z["nrec"])
    gr = Green(rec, z["ene"], sym_term=bool(z["sym_term"]))
    r = gr.block_ldos()
    assert np.abs(r["a_inf"] - z["a_inf"]).max() <= 1e-9 * np.abs(z["a_inf"]).max()      # continuous in the coefficients
    assert np.abs(r["b_inf"] - z["b_inf"]).max() <= 1e-9 * np.abs(z["b_inf"]).max()
    dtot, dosia, dosial = ldos_from_g0(z["g0"])
    # scale: the largest LDOS of the mesh.  (Outside the band Im g0 is rounding noise of either code, 1e-18 with eta = 0: a
    # per-energy relative bound has no meaning there; inside the band the two agree to 1e-13.)
    scale = np.abs(dosial).max()
    assert np.abs(r["dosial"] - dosial).max() <= 1e-10 * scale
    assert np.abs(r["dosia"] - dosia).max() <= 1e-10 * scale
    assert np.abs(r["dtot"] - dtot).max() <= 1e-10 * n * scale
    # b2_b of the recursion is still B^2 (the stage takes its own square root): the host path gives the same numbers
    rec.zsqr()
    g0 = gr.block_green(r["a_inf"], r["b_inf"], nsites=n)
    dt2, da2, dl2 = ldos_from_g0(g0)
    assert np.abs(r["dosial"] - dl2).max() <= 1e-12 * np.abs(dl2).max()
    assert np.abs(r["dtot"] - dt2).max() <= 1e-12 * np.abs(dt2).max()
    rec.close()


def test_ldos_images_are_zero_padded():
    """The images the ranks all-reduce (bands.f90:271-274): this rank's sites at their global positions, zeros elsewhere."""
    name = "fccCu001_block_hoh"                                                          # two sites
    z = load_green(name)
    g = load_golden(name)
    rec = Recursion(*objects_from(problem_dict(g), g["irec"], g["lld"], nsp=g["nsp"]), device=0)
    rec.recur_b()
    gr = Green(rec, z["ene"])
    n, ntot, off = 2, 5, 2
    host = gr.block_ldos(site_offset=off, nsites_total=ntot)
    local = gr.block_ldos()
    assert np.array_equal(host["dosial"][off:off + n], local["dosial"]) and np.array_equal(host["dosia"][off:off + n], local["dosia"])
    assert np.array_equal(host["dtot"], local["dtot"])
    mask = np.ones(ntot, bool); mask[off:off + n] = False
    assert np.all(host["dosial"][mask] == 0) and np.all(host["dosia"][mask] == 0)
    rec.close()


DEVICE_OUTPUT_SCRIPT = r"""
import sys, numpy as np, torch
torch.cuda.init(); torch.cuda.set_device(0)          # torch's HIP runtime first, as in bench.py
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from helpers import load_golden, objects_from, problem_dict
from test_gpu_green import load_green
from rslmtoasa_amd.green import Green
from rslmtoasa_amd.recursion import Recursion
name = "fccCu001_block_hoh"
z, g = load_green(name), load_golden(name)
rec = Recursion(*objects_from(problem_dict(g), g["irec"], g["lld"], nsp=g["nsp"]), device=0)
rec.recur_b()
gr = Green(rec, z["ene"])
n, nen, ntot, off = 2, len(z["ene"]), 5, 2
host = gr.block_ldos(site_offset=off, nsites_total=ntot)
t_tot = torch.full((nen,), -1.0, dtype=torch.float64, device="cuda")
t_ia = torch.full((nen, ntot), -1.0, dtype=torch.float64, device="cuda")              # Fortran (ntot, nen)
t_ial = torch.full((nen, 18, ntot), -1.0, dtype=torch.float64, device="cuda")         # Fortran (ntot, 18, nen)
torch.cuda.synchronize()
gr.block_ldos(site_offset=off, nsites_total=ntot, out=(t_tot.data_ptr(), t_ia.data_ptr(), t_ial.data_ptr()))
assert np.array_equal(t_tot.cpu().numpy(), host["dtot"])
assert np.array_equal(t_ia.cpu().numpy().T, host["dosia"])
assert np.array_equal(t_ial.cpu().numpy().transpose(2, 1, 0), host["dosial"])
# the packed diagonal coefficients, same contract
a_img = torch.full((ntot, 18, g["lld"]), -1.0, dtype=torch.float64, device="cuda")
b_img = torch.full((ntot, 18, g["lld"]), -1.0, dtype=torch.float64, device="cuda")
torch.cuda.synchronize()
rec.pack_diag(off, ntot, a_img.data_ptr(), b_img.data_ptr())
a = a_img.cpu().numpy()
assert np.array_equal(a[off:off + n], rec.a[:g["lld"], :, :n, 0].transpose(2, 1, 0)) and np.all(a[:off] == 0) and np.all(a[off + n:] == 0)
ah = np.zeros((g["lld"], 18, ntot), order="F"); bh = np.zeros_like(ah)
rec.pack_diag(off, ntot, ah, bh)
assert np.array_equal(ah.transpose(2, 1, 0), a) and np.array_equal(bh.transpose(2, 1, 0), b_img.cpu().numpy())
rec.close()
print("DEVICE_OUTPUT_OK")
"""


def test_device_outputs_match_host_outputs():
    """Outputs handed over as DEVICE buffers (the tensors a collective would reduce) receive the same numbers as host arrays.
    Own process: torch's HIP runtime has to be initialised before librsrec's (as in bench.py); the other tests of this session
    have already started librsrec's."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", DEVICE_OUTPUT_SCRIPT, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DEVICE_OUTPUT_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_ldos_positive_on_full_mesh():
    """Full 2510-point mesh of the reference with a small positive broadening: every orbital-resolved LDOS is positive, and the
    stage is repeatable bit for bit."""
    z = load_green("bccFe_nsp2_block")
    g = load_golden("bccFe_nsp2_block")
    rec = Recursion(*objects_from(problem_dict(g), g["irec"], g["lld"], nsp=g["nsp"]), device=0)
    rec.recur_b()
    ene = float(z["ene_full_first"]) + float(z["ene_full_step"]) * np.arange(int(z["nen_full"]))
    gr = Green(rec, ene)
    r1 = gr.block_ldos(eta=1e-3j)
    r2 = gr.block_ldos(eta=1e-3j)
    assert (r1["dosial"] > 0).all() and np.array_equal(r1["dosial"], r2["dosial"]) and np.array_equal(r1["dtot"], r2["dtot"])
    rec.close()


def test_ldos_needs_resident_coefficients():
    from rslmtoasa_amd import _lib
    g = load_golden("bccFe_nsp2_block")
    rec = Recursion(*objects_from(problem_dict(g), g["irec"], g["lld"], nsp=g["nsp"]), device=0)
    with pytest.raises(_lib.RsrecError):
        Green(rec, np.linspace(-1, 1, 4)).block_ldos()
    rec.close()
