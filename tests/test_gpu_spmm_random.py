"""H|psi> of the matrix-core SpMM (k_spmm5, through rsrec_apply_operator) against a plain numpy restatement of the reference's
ham_vec_matmul / ham_hoh_vec_matmul (recursion.f90:913 / :785) on RANDOM ragged lattices: up to the kernel's maximum of 31 neighbour
slots, atoms with missing neighbours (0 entries), several atom types, per-atom (impurity) blocks, spin-diagonal and spin-mixing
blocks, with and without hoh.  The schedule of the kernel (a stream of 9-orbital entries cut into steps of 4/4/2 orbitals with a
27-step period) sees every tail length and entry count here; the fixtures of the other tests only a few.
"""
import numpy as np
import pytest

from helpers import objects_from
from rslmtoasa_amd.recursion import Recursion

pytestmark = pytest.mark.gpu


def random_problem(rng, kk, nslots, ntype, nmax, hoh, collinear):
    nn = np.zeros((kk, nslots + 1), np.int32, order="F")
    for k in range(kk):
        nr = nslots if k % 7 else int(rng.integers(1, nslots + 1))     # most atoms full, some with fewer slots
        nn[k, 0] = nr
        for nb in range(2, nr + 1):
            nn[k, nb - 1] = 0 if rng.random() < 0.1 else int(rng.integers(1, kk + 1))   # 0 = absent neighbour
    nn[kk - 1, 0] = nslots                                               # at least one atom uses every slot
    iz = rng.integers(1, ntype + 1, kk).astype(np.int32)

    def blocks(n, last):
        a = (rng.standard_normal((18, 18, n, last)) + 1j * rng.standard_normal((18, 18, n, last))) * 0.2
        if collinear:                                                    # hopping blocks of a collinear magnet: no spin-flip part
            a[:9, 9:] = 0
            a[9:, :9] = 0
        return np.asfortranarray(a)

    p = dict(nn=nn, iz=iz, nmax=nmax, hoh=int(hoh), nsp=2, ee=blocks(nslots, ntype),
             lsham=np.asfortranarray((rng.standard_normal((18, 18, ntype)) + 1j * rng.standard_normal((18, 18, ntype))) * 0.1))
    if nmax:
        p["hall"] = blocks(nslots, nmax)
    if hoh:
        p["eeo"] = blocks(nslots, ntype)
        p["enim"] = np.asfortranarray((rng.standard_normal((18, 18, ntype)) + 1j * rng.standard_normal((18, 18, ntype))) * 0.1)
        if nmax:
            p["hallo"] = blocks(nslots, nmax)
    return p


def apply_blocks(p, x, use_o):
    """sum over the slots of block(slot, class of k) @ x[neighbour]: the loops of recursion.f90:935-975 with every atom active."""
    kk = x.shape[2]
    out = np.zeros_like(x)
    for k in range(kk):
        if k < p["nmax"]:
            H = (p["hallo"] if use_o else p["hall"])[:, :, :, k]
        else:
            H = (p["eeo"] if use_o else p["ee"])[:, :, :, p["iz"][k] - 1]
        acc = H[:, :, 0] @ x[:, :, k]
        for nb in range(2, p["nn"][k, 0] + 1):
            n = p["nn"][k, nb - 1]
            if n:
                acc = acc + H[:, :, nb - 1] @ x[:, :, n - 1]
        out[:, :, k] = acc
    return out


def ham_vec_numpy(p, x, a, b):
    ls = np.stack([p["lsham"][:, :, t - 1] for t in p["iz"]], axis=2)
    soc = np.einsum("ijk,jlk->ilk", ls, x)
    if not p["hoh"]:
        out = apply_blocks(p, x, False) + soc
    else:
        en = np.einsum("ijk,jlk->ilk", np.stack([p["enim"][:, :, t - 1] for t in p["iz"]], axis=2), x)
        h = apply_blocks(p, x, False)
        out = h - apply_blocks(p, h, True) + en + soc                    # :905
    return (out - b * x) / a


CASES = [  # kk, nslots, ntype, nmax, hoh, collinear
    (150, 31, 3, 4, False, False),
    (150, 31, 3, 4, True, False),
    (233, 31, 2, 0, False, True),
    (233, 30, 2, 3, True, True),
    (97, 1, 1, 0, False, False),
    (97, 2, 2, 1, True, True),
    (180, 15, 1, 0, False, True),
    (180, 15, 1, 0, True, True),
    (150, 31, 1, 0, True, False),
    (180, 19, 4, 7, False, True),
    (180, 13, 4, 7, True, False),
    (64, 9, 2, 2, True, True),
]


@pytest.mark.parametrize("kk,nslots,ntype,nmax,hoh,collinear", CASES)
def test_whole_vector_product_on_random_ragged_lattice(kk, nslots, ntype, nmax, hoh, collinear):
    rng = np.random.default_rng(1000 * kk + 10 * nslots + ntype + 2 * int(hoh) + int(collinear))
    p = random_problem(rng, kk, nslots, ntype, nmax, hoh, collinear)
    rec = Recursion(*objects_from(p, np.array([1], np.int32), 4), device=0)
    rec.set_option("s5_queue", 2)    # operator stream in LDS: persistent workgroups with a group queue
    rec.set_option("s5_lds", 2)      # several classes of atoms: the class-run form (workgroup rows dealt to the runs of the class-sorted atom list)
    rec.set_option("s5_run_min", 1)  # ... for runs of any size
    x = np.asfortranarray(rng.standard_normal((18, 18, kk)) + 1j * rng.standard_normal((18, 18, kk)))
    a, b = 1.7, -0.3
    want = ham_vec_numpy(p, x, a, b)
    got = rec.ham_hoh_vec_matmul(x, a, b) if hoh else rec.ham_vec_matmul(x, a, b)
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 2e-13 * scale
    if hoh:
        # ham_vec_matmul itself is the PLAIN operator ee + l.s also when hoh is set (recursion.f90:913-977; chebyshev_orbital_mod
        # relies on it, :2950, :2964)
        plain = (apply_blocks(p, x, False) + np.einsum("ijk,jlk->ilk", np.stack([p["lsham"][:, :, t - 1] for t in p["iz"]], axis=2), x) - b * x) / a
        got = rec.ham_vec_matmul(x, a, b)
        assert np.abs(got - plain).max() <= 2e-13 * np.abs(plain).max()
    rec.close()


RECUR_CASES = [  # kk, nslots, ntype, nmax, hoh, collinear
    (120, 31, 3, 4, False, False),
    (120, 22, 2, 3, True, True),
    (90, 7, 2, 0, True, False),
    (140, 14, 1, 0, False, True),
]


@pytest.mark.parametrize("variant", [1, 2, "ci", "runs"])
@pytest.mark.parametrize("kk,nslots,ntype,nmax,hoh,collinear", RECUR_CASES)
def test_recursions_on_random_ragged_lattice(kk, nslots, ntype, nmax, hoh, collinear, variant, oracle_lib):
    """Block Lanczos and Chebyshev moments on the same random lattices (region growth through missing neighbours and impurity atoms)
    against the CPU oracle, for the VALU set, the matrix-core set and the matrix-core set with the large-launch SpMM forced."""
    from helpers import RTOL, rel_err
    rng = np.random.default_rng(77 + kk + nslots)
    p = random_problem(rng, kk, nslots, ntype, nmax, hoh, collinear)
    irec = np.array([1, kk // 2, kk], np.int32)          # an impurity atom (if any), a bulk atom, the last atom
    lld = 6
    rec = Recursion(*objects_from(p, irec, lld, emin=-60.0, emax=60.0), device=0)
    if variant in ("ci", "runs"):
        rec.set_option("kernels", 2)
        rec.set_option("spmm5", 2)
        if variant == "runs":
            # operators with several classes of atoms: once a chain's region covers the lattice every class run of the class-sorted atom
            # list gets a launch with that class's operator stream in LDS (forced here for runs of any size, persistent queue form
            # included), the rest -- impurity atoms, chains still growing -- the global-load launch
            rec.set_option("s5_lds", 2)
            rec.set_option("s5_run_min", 1)
            rec.set_option("s5_queue", 2)
    else:
        rec.set_option("kernels", variant)
        rec.set_option("spmm5", 1)
    o = oracle_lib.Oracle(p)
    rec.recur_b()
    a_o, b_o = o.block_lanczos(irec, lld)
    assert rel_err(rec.a_b, a_o) < RTOL and rel_err(rec.b2_b, b_o) < RTOL
    rec.chebyshev_recur()
    from rslmtoasa_amd.recursion import chebyshev_scaling
    a, b = chebyshev_scaling(-60.0, 60.0)                        # recursion.f90:3078-3079
    mu_o, div = o.chebyshev(irec, lld, a, b)
    assert div == 0
    assert rel_err(rec.mu_n, mu_o) < RTOL
    rec.close()


@pytest.mark.parametrize("waves", [8, 12])
@pytest.mark.parametrize("kk,nslots,hoh,collinear", [(180, 15, False, True), (180, 15, True, True), (150, 31, True, False), (97, 3, False, False), (233, 30, False, False)])
def test_split_tasks_equal_the_nine_tile_wave_bitwise(kk, nslots, hoh, collinear, waves):
    """Option s5_split = 3 (k_spmm5<., true, false, 3>: a wave takes a third of a group's nine tiles; persistent form of one-class operators)
    accumulates every tile in the same k order as the nine-tile wave: block Lanczos and Chebyshev results must be bitwise the same, on
    ragged lattices (padding atoms in the last group, absent neighbours), with and without hoh (two-input second pass) and with
    spin-mixing blocks."""
    rng = np.random.default_rng(4242 + kk + nslots + int(hoh))
    p = random_problem(rng, kk, nslots, 1, 0, hoh, collinear)
    irec = np.array([1, kk // 2, kk], np.int32)
    rec = Recursion(*objects_from(p, irec, 6, emin=-60.0, emax=60.0), device=0)
    for k, v in (("kernels", 2), ("spmm5", 2), ("s5_queue", 2), ("graph", 0)):
        rec.set_option(k, v)
    rec.recur_b()
    a0, b0 = rec.a_b.copy(), rec.b2_b.copy()
    rec.chebyshev_recur()
    m0 = rec.mu_n.copy()
    rec.set_option("s5_split", 3)
    rec.set_option("s5_waves", waves)
    rec.recur_b()
    assert np.isfinite(rec.a_b).all() and np.array_equal(rec.a_b, a0) and np.array_equal(rec.b2_b, b0)
    rec.chebyshev_recur()
    assert np.array_equal(rec.mu_n, m0)
    rec.close()
