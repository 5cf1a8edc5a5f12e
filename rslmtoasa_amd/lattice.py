"""Synthetic periodic lattices in the reference's table format (the recursion's geometry inputs).

The recursion reads only three tables of the reference's ``lattice`` type (lattice.f90:138-239):

* ``nn(kk, nncols)`` int32, 1-based, ``nn[:, 0]`` = neighbour count including the on-site slot,
  ``nn[i, m]`` = atom in neighbour slot *m* (0 = absent) -- slot *m* is the same displacement vector
  for every atom of a type (lattice.f90:2823-2893),
* ``iz(kk)`` type of every atom, ``irec(nrec)`` the recursion seed sites.

The reference builds them with an O(kk^2) search (lattice.f90:3035); for periodic supercells
(``pbc``, lattice.f90:1037-1085) the table is a pure index calculation, done here in O(kk).
"""
import numpy as np

# bcc primitive vectors in units of alat (lattice.f90:739-741)
BCC_PRIMITIVE = np.array([[-0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.5, 0.5, -0.5]])


def bcc_supercell(dims, slot_vec, primitive=BCC_PRIMITIVE):
    """Neighbour table of an n1 x n2 x n3 periodic supercell of a one-atom Bravais lattice.

    ``slot_vec[m]`` is the Cartesian displacement (units of alat) of neighbour slot m, m = 1..nb-1
    (row 0 is the on-site slot); atoms are numbered with the first cell index fastest.
    Returns ``nn`` with shape (kk, nb + 1) in the reference's convention (last column unused = 0).
    """
    n1, n2, n3 = (int(x) for x in dims)
    slot_vec = np.asarray(slot_vec, dtype=np.float64)
    nb = slot_vec.shape[0]
    steps = np.rint(slot_vec @ np.linalg.inv(primitive)).astype(np.int64)  # d = sum_k steps[k] * a_k
    if not np.allclose(steps @ primitive, slot_vec, atol=1e-9):
        raise ValueError("slot vectors are not lattice vectors of the given primitive cell")
    kk = n1 * n2 * n3
    c1, c2, c3 = np.meshgrid(np.arange(n1), np.arange(n2), np.arange(n3), indexing="ij")
    c1, c2, c3 = (c.ravel(order="F") for c in (c1, c2, c3))
    nn = np.zeros((kk, nb + 1), dtype=np.int32, order="F")
    nn[:, 0] = nb
    for m in range(1, nb):
        j = ((c1 + steps[m, 0]) % n1) + n1 * (((c2 + steps[m, 1]) % n2) + n2 * ((c3 + steps[m, 2]) % n3))
        nn[:, m] = j + 1
    return nn


def supercell_positions(dims, primitive=BCC_PRIMITIVE):
    """Cartesian positions (3, kk) of the supercell atoms in units of alat, same atom numbering as bcc_supercell."""
    n1, n2, n3 = (int(x) for x in dims)
    c1, c2, c3 = np.meshgrid(np.arange(n1), np.arange(n2), np.arange(n3), indexing="ij")
    cells = np.stack([c.ravel(order="F") for c in (c1, c2, c3)], axis=1).astype(np.float64)
    return np.asfortranarray((cells @ primitive).T)


def spread_sites(kk, nsites):
    """Seed sites 1 + k*floor(kk/S), k = 0..S-1, 1-based like ``irec``."""
    stride = max(kk // max(nsites, 1), 1)
    return (1 + stride * np.arange(nsites, dtype=np.int64)).astype(np.int32)


def active_region_sizes(nn, seed, nsteps):
    """N_act after each of ``nsteps`` applications of H starting from ``seed`` (1-based):
    the breadth-first growth of ``izero`` in hop_b (recursion.f90:1604-1636)."""
    kk = nn.shape[0]
    nb = int(nn[:, 0].max())
    active = np.zeros(kk + 1, dtype=bool)
    active[seed] = True
    sizes = []
    nbr = nn[:, 1:nb]
    for _ in range(nsteps):
        hit = active[nbr].any(axis=1)  # index 0 (absent) is never active
        active[1:] |= hit
        active[0] = False
        sizes.append(int(active.sum()))
    return sizes
