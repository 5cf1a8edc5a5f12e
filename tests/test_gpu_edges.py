"""Edge cases of the recursion drivers on the GPU, against the CPU oracle (which mirrors the reference statement by statement):
shortest recursions (lld = 1, 2), repeated and last-atom seeds, more chains than one batch, zero sites."""
import numpy as np
import pytest

from helpers import RTOL, objects_from, rel_err, supercell_problem
from rslmtoasa_amd.recursion import Recursion, chebyshev_scaling

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lld", [1, 2, 3])
@pytest.mark.parametrize("hoh", [False, True])
def test_shortest_recursions(lld, hoh, oracle_lib):
    p = supercell_problem((4, 4, 8), hoh=hoh)
    kk = p["nn"].shape[0]
    sites = np.array([1, kk, 1, 77], dtype=np.int32)          # first atom, last atom, a repeated site
    rec = Recursion(*objects_from(p, sites, lld, emin=-3.0, emax=1.8), device=0)
    rec.recur_b()
    rec.chebyshev_recur()
    o = oracle_lib.Oracle(p)
    a_o, b_o = o.block_lanczos(sites, lld)
    mu_o, rc = o.chebyshev(sites, lld, *chebyshev_scaling(-3.0, 1.8))
    assert rc == 0
    if lld > 1:
        assert rel_err(rec.a_b[:, :, :, :4], a_o) < RTOL
    else:
        assert np.all(rec.a_b[:, :, :, :4] == 0) and np.all(a_o == 0)      # a_b(:,:,lld) = 0 (recursion.f90:1836)
    assert rel_err(rec.b2_b[:, :, :, :4], b_o) < RTOL
    assert rel_err(rec.mu_n[:, :, :, :4], mu_o) < RTOL
    assert np.array_equal(rec.a_b[:, :, :, 0], rec.a_b[:, :, :, 2])       # the repeated site gives bit-identical coefficients
    rec.close()


def test_more_chains_than_one_batch_and_zero_sites(oracle_lib):
    p = supercell_problem((4, 4, 8))
    kk = p["nn"].shape[0]
    sites = np.arange(1, kk + 1, dtype=np.int32)              # every atom of the cell: 128 chains = two batches of 64
    rec = Recursion(*objects_from(p, sites, 6), device=0)
    rec.recur_b()
    for s in range(1, kk):                                    # periodic cell of one type: all sites equivalent
        assert rel_err(rec.a_b[:, :, :, s], rec.a_b[:, :, :, 0]) < RTOL
    a_o, b_o = oracle_lib.Oracle(p).block_lanczos(sites[:2], 6)
    assert rel_err(rec.a_b[:, :, :, :2], a_o) < RTOL and rel_err(rec.b2_b[:, :, :, :2], b_o) < RTOL
    rec.close()
    rec0 = Recursion(*objects_from(p, np.zeros(0, np.int32), 6), device=0)
    rec0.recur_b()                                            # no sites on this rank: a no-op, like the reference's empty loop
    rec0.close()


def test_unchanged_lattice_keeps_the_cached_regions():
    """Every SCF iteration of the reference hands over the same lattice%nn (the Fortran shim calls rsrec_set_lattice before every
    driver call): an identical table must not throw away the cached regions; a changed one must take effect."""
    from helpers import objects_from, supercell_problem
    from rslmtoasa_amd.recursion import Recursion
    p = supercell_problem((11, 11, 11))
    sites = np.array([1, 500, 900, 1331], dtype=np.int32)
    rec = Recursion(*objects_from(p, sites, 10))
    rec.recur_b()
    t_first = rec.timing()["host_ms"]
    a0 = rec.a_b.copy()
    rec.update_lattice()                   # same tables again
    rec.update_hamiltonian()
    rec.recur_b()
    assert np.array_equal(a0, rec.a_b)
    assert rec.timing()["host_ms"] < max(0.2 * t_first, 0.5), (t_first, rec.timing()["host_ms"])    # no breadth-first search this time
    nn = rec.lattice.nn.copy()
    nn[0, 2] = 0                           # atom 1 loses one neighbour: a different operator
    rec.lattice.nn = nn
    rec.update_lattice()
    rec.update_hamiltonian()
    rec.recur_b()
    assert np.abs(rec.a_b[:, :, :, 0] - a0[:, :, :, 0]).max() > 1e-6      # (first visible in A_3: psi_2 vanishes on the seed atom)
    rec.close()


@pytest.mark.parametrize("nmax", [0, 3])
def test_scalar_recursion_ignores_the_spin_orbit_block(nmax, oracle_lib):
    """The scalar hop reads ee(:,:,1,ih) / hall(:,:,1,i) alone (recursion.f90:3336, :3372): a non-zero `lsham` (never built in the
    reference's own nsp = 1 runs, so none of its fixtures has one) must not enter, although the block recursion's on-site table carries
    it.  Found by tools/fuzz_recursion.py; random ragged lattice, several atom types, with and without impurity atoms."""
    from helpers import rel_err_rows
    from test_gpu_spmm_random import random_problem
    rng = np.random.default_rng(4242 + nmax)
    p = random_problem(rng, 150, 9, 2, nmax, False, True)
    p["nsp"] = 1
    assert np.abs(p["lsham"]).max() > 0.0
    irec = np.array([1, 75, 150], np.int32)
    lld = 6
    rec = Recursion(*objects_from(p, irec, lld, nsp=1, llsp=lld), device=0)
    rec.recur()
    a_o, b_o = oracle_lib.Oracle(p).scalar_lanczos(irec, lld, lld)
    assert rel_err_rows(rec.a[:, :, :3, 0], a_o) < RTOL and rel_err_rows(rec.b2[:, :, :3, 0], b_o) < RTOL
    rec.close()


@pytest.mark.parametrize("hoh", [False, True])
@pytest.mark.parametrize("nsites", [2, 11])
def test_per_atom_blocks_grouped_over_chains(nsites, hoh, oracle_lib):
    """Atoms with their own operator blocks (`hall`, an impurity region) are one-tile groups; option s5_octet forms their groups over 8
    chains of the batch once every chain's region covers the lattice (k_spmm5<., false, true>).  Nine such atoms with DIFFERENT blocks in a
    128-atom periodic cell, a full and a partial octet of chains, +- hoh: block Lanczos and Chebyshev moments against the oracle, and the
    same numbers, bit for bit, as without the option (the grouping changes no summation order)."""
    rng = np.random.default_rng(99)
    p = dict(supercell_problem((4, 4, 8), hoh=hoh))
    nmax = 9
    scale = 1.0 + 0.05 * rng.standard_normal(nmax)
    p["nmax"] = nmax
    p["hall"] = np.asfortranarray(p["ee"][:, :, :, :1] * scale)
    if hoh:
        p["hallo"] = np.asfortranarray(p["eeo"][:, :, :, :1] * scale)
    irec = rng.choice(128, nsites, replace=False).astype(np.int32) + 1
    lld = 9
    rec = Recursion(*objects_from(p, irec, lld, emin=-3.0, emax=1.8), device=0)
    rec.set_option("s5_octet", 0)
    rec.recur_b(); rec.chebyshev_recur()
    a0, b0, m0 = rec.a_b.copy(), rec.b2_b.copy(), rec.mu_n.copy()
    assert rec.timing()["octet_launches"] == 0
    rec.set_option("s5_octet", 1)
    rec.recur_b()
    assert rec.timing()["octet_launches"] > 0
    rec.chebyshev_recur()
    assert rec.timing()["octet_launches"] > 0
    assert np.array_equal(rec.a_b, a0) and np.array_equal(rec.b2_b, b0) and np.array_equal(rec.mu_n, m0)
    o = oracle_lib.Oracle(p)
    a_o, b_o = o.block_lanczos(irec, lld)
    mu_o, div = o.chebyshev(irec, lld, *chebyshev_scaling(-3.0, 1.8))
    assert div == 0 and rel_err(rec.a_b, a_o) < RTOL and rel_err(rec.b2_b, b_o) < RTOL and rel_err(rec.mu_n, mu_o) < RTOL
    rec.close()
