"""The stage behind the scalar recursion on the GPU -- dos%density with bprldos (density_of_states.f90:248-404) as rsrec_scalar_density,
and the g0 green%sgreen (green.f90:628-705) makes of it -- against the compiled reference's outputs for the same coefficients
(tests/golden/*_density.npz) and against the CPU oracle on perturbed chains."""
import os

import numpy as np
import pytest

from helpers import GOLD, load_golden, objects_from
from rslmtoasa_amd.green import Green
from rslmtoasa_amd._lib import RsrecError
from rslmtoasa_amd.recursion import Recursion

pytestmark = pytest.mark.gpu

CASES = ["bccFe_nsp1_lanczos", "fccCu001_nsp1_lanczos", "B2FeCo_nsp1_lanczos"]


def scalar_recursion(name):
    p = load_golden(name)
    rec = Recursion(*objects_from(p, p["irec"], int(p["lld"]), nsp=1), device=0)
    return p, rec


@pytest.mark.parametrize("name", CASES)
def test_density_and_sgreen_match_reference(name):
    """recur (scalar Haydock recursion) -> density -> sgreen, all on the device, against the reference's tdens and g0."""
    with np.load(os.path.join(GOLD, name + "_density.npz"), allow_pickle=False) as z:
        p, rec = scalar_recursion(name)
        rec.recur()
        n = int(z["nrec"])
        assert np.abs(rec.a[:, :, :n, 0] - z["a"][:, :, :, 0]).max() <= 1e-10 * np.abs(z["a"]).max()
        gr = Green(rec, z["ene"])
        t = gr.density(z["dw_l"], z["cshi"], nsites=n)
        ref = z["tdens"]
        assert t.shape == ref.shape
        # the chains are the GPU recursion's own (1e-10 from the reference's): the fractions amplify that near the band edges
        assert np.abs(t - ref).max() <= 1e-7 * np.abs(ref).max()
        # ... and from the reference's own coefficients the stage itself agrees to rounding
        rec.a[:, :, :n, 0] = z["a"][:, :, :, 0]
        rec.b2[:, :, :n, 0] = z["b2"][:, :, :, 0]
        t = gr.density(z["dw_l"], z["cshi"], nsites=n)
        assert np.abs(t - ref).max() <= 1e-12 * np.abs(ref).max()
        g0 = gr.sgreen(z["dw_l"], z["cshi"], nsites=n)
        assert np.abs(g0 - z["g0"]).max() <= 1e-12 * np.abs(z["g0"]).max()
        rec.close()


def test_density_matches_oracle_on_perturbed_chains_and_parameters(oracle_lib):
    """Many sites, three directions, dw_l /= 1 and cshi /= 0, energies on both sides of every band: GPU against the CPU restatement."""
    rng = np.random.default_rng(11)
    with np.load(os.path.join(GOLD, "fccCu001_nsp1_lanczos_density.npz"), allow_pickle=False) as z:
        p, rec = scalar_recursion("fccCu001_nsp1_lanczos")
        llmax, nsites, nmd = z["a"].shape[0], 7, 3
        a = np.asfortranarray(z["a"][:, :, [0, 1, 0, 1, 0, 1, 0], :][:, :, :, [0, 0, 0]] + 0.05 * rng.standard_normal((llmax, 18, nsites, nmd)))
        b2 = np.asfortranarray(z["b2"][:, :, [0, 1, 0, 1, 0, 1, 0], :][:, :, :, [0, 0, 0]] * (1.0 + 0.1 * rng.random((llmax, 18, nsites, nmd))))
        dw = np.asfortranarray(1.0 + 0.2 * rng.random((18, nsites)))
        cs = np.asfortranarray(0.05 * rng.standard_normal((18, nsites)))
        ene = np.linspace(-1.5, 1.5, 333)
        want = oracle_lib.scalar_density(a, b2, ene, dw, cs, llmax)
        rec.a = a; rec.b2 = b2
        gr = Green(rec, ene)
        got = gr.density(dw, cs, nsites=nsites, nmdir=nmd)
        assert got.shape == want.shape == (18, 333, nsites, nmd)
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
        # a shorter chain than the arrays hold (control%lld < llmax)
        rec.control.lld = llmax - 3
        want = oracle_lib.scalar_density(a, b2, ene, dw, cs, llmax - 3)
        got = gr.density(dw, cs, nsites=nsites, nmdir=nmd)
        assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
        rec.close()


def test_density_on_deep_chains(oracle_lib):
    """lld = 150 (the band-edge kernel then runs fewer threads per workgroup: one LDS column per thread): GPU against the CPU restatement."""
    rng = np.random.default_rng(5)
    p, rec = scalar_recursion("bccFe_nsp1_lanczos")
    lld, n = 150, 3
    a = np.asfortranarray(0.01 * rng.standard_normal((lld, 18, n, 1)))          # a weakly disordered chain: a band about [-1, 1]
    b2 = np.asfortranarray(0.25 * (1.0 + 0.02 * rng.random((lld, 18, n, 1))))
    ene = np.linspace(-1.2, 1.2, 97)
    rec.a = a; rec.b2 = b2; rec.control.lld = lld
    want = oracle_lib.scalar_density(a, b2, ene, np.ones((18, n)), np.zeros((18, n)), lld)
    got = Green(rec, ene).density(nsites=n)
    assert np.abs(want).max() > 0.1 and np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    rec.close()


def test_density_argument_errors():
    p, rec = scalar_recursion("bccFe_nsp1_lanczos")
    gr = Green(rec, np.linspace(-1, 1, 5))
    rec.control.lld = rec.a.shape[0] + 1                     # deeper than the arrays
    with pytest.raises(RsrecError):
        gr.density()
    rec.control.lld = 1
    with pytest.raises(RsrecError):
        gr.density()
    rec.close()
