"""Random Hermitian operators: every hopping block is a full 18x18 complex matrix with spin-flip entries (a non-collinear
magnet with spin-orbit hopping), which none of the reference-derived fixtures has (there only the on-site block mixes spins).
Exercises the spin-flip schedule entries of k_spmm5 and the full pattern of k_spmm4 for EVERY slot, block Lanczos and Chebyshev,
with and without hoh, against the CPU oracle.  Parity unpinned by the reference (no such case exists there): the oracle is the
checker, and it is pinned on all reference fixtures by tests/test_oracle_golden.py."""
import numpy as np
import pytest

from helpers import RTOL, load_golden, objects_from, rel_err
from rslmtoasa_amd.lattice import bcc_supercell
from rslmtoasa_amd.recursion import Recursion, chebyshev_scaling

pytestmark = pytest.mark.gpu


def random_problem(seed, hoh, scale=0.08):
    rng = np.random.default_rng(seed)
    vec = load_golden("bccFe_nsp2_block")["slot_vec"]
    nb = vec.shape[0]
    nn = bcc_supercell((4, 4, 4), vec)
    opp = [int(np.argmin(np.abs(vec + vec[s]).sum(axis=1))) for s in range(nb)]      # slot of the opposite displacement
    assert all(np.allclose(vec[opp[s]], -vec[s]) for s in range(1, nb))

    def blocks():
        b = scale * (rng.standard_normal((18, 18, nb)) + 1j * rng.standard_normal((18, 18, nb)))
        b[:, :, 0] = 0.5 * (b[:, :, 0] + b[:, :, 0].conj().T)                            # on-site: Hermitian
        for s in range(1, nb):
            if opp[s] > s:
                b[:, :, opp[s]] = b[:, :, s].conj().T                                      # H(j,i) = H(i,j)^H
        return np.asfortranarray(b[:, :, :, None])
    ls = scale * (rng.standard_normal((18, 18)) + 1j * rng.standard_normal((18, 18)))
    p = dict(nn=nn, iz=np.ones(nn.shape[0], np.int32), ee=blocks(), lsham=np.asfortranarray(0.5 * (ls + ls.conj().T))[:, :, None], hoh=int(hoh), nsp=4, nmax=0)
    if hoh:
        en = scale * (rng.standard_normal((18, 18)) + 1j * rng.standard_normal((18, 18)))
        p.update(eeo=0.3 * blocks(), enim=np.asfortranarray(0.5 * (en + en.conj().T))[:, :, None])
    return p


@pytest.mark.parametrize("variant", ["coop", "kp"])
@pytest.mark.parametrize("hoh", [False, True])
@pytest.mark.parametrize("seed", [1, 2])
def test_full_complex_hopping_blocks(seed, hoh, variant, oracle_lib):
    p = random_problem(seed, hoh)
    sites = np.array([1, 30, 64], dtype=np.int32)
    lld = 8
    rec = Recursion(*objects_from(p, sites, lld, nsp=4, emin=-6.0, emax=6.0), device=0)
    rec.set_option("kernels", 2)
    rec.set_option("spmm5", 2 if variant == "kp" else 1)        # kp: k_spmm5 (the default); coop: k_spmm4<4> (small launches of the plain operator when asked for)
    rec.recur_b()
    rec.chebyshev_recur()
    o = oracle_lib.Oracle(p)
    a_o, b_o = o.block_lanczos(sites, lld)
    mu_o, rc = o.chebyshev(sites, lld, *chebyshev_scaling(-6.0, 6.0))
    assert rc == 0
    assert rel_err(rec.a_b[:, :, :, :3], a_o) < RTOL
    assert rel_err(rec.b2_b[:, :, :, :3], b_o) < RTOL
    assert rel_err(rec.mu_n[:, :, :, :3], mu_o) < RTOL
    rec.close()
