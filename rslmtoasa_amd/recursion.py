"""Host-side mirror of the reference's ``recursion_mod`` (source/recursion.f90) on top of the C ABI.

Same names, argument meaning and error behaviour as the Fortran type the rest of RS-LMTO-ASA drives
(recursion.f90:41-116): a ``Recursion`` object borrows ``hamiltonian``/``lattice``/``control``/``energy``
objects, exposes ``recur()``, ``recur_b()``, ``chebyshev_recur()``, ``recur_b_ij()``, ``zsqr()`` and fills
``a, b2, a_b, b2_b, mu_n`` with the reference's shapes and index order (arrays are Fortran-ordered numpy
arrays, so ``a_b[l, m, ll, site]`` reads like ``a_b(l+1, m+1, ll+1, site+1)``).

All arithmetic happens in librsrec (HIP kernels); this module only marshals arrays.  The production host
for the reference is the Fortran shim under ``fortran/`` -- this Python mirror exists so the parity tests
and the benchmark read like the reference's own call sites (self.f90:799-806, :829).
"""
import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from . import _lib


@dataclass
class Control:
    """control%lld, %llsp, %nsp, %recur (control.f90:36-164)."""
    lld: int = 21
    llsp: int = 0
    nsp: int = 2
    recur: str = "block"


@dataclass
class Energy:
    """energy%energy_min / %energy_max (energy.f90:175-207)."""
    energy_min: float = -1.0
    energy_max: float = 1.0


@dataclass
class Lattice:
    """The tables of lattice.f90:138-239 the recursion reads."""
    nn: np.ndarray            # (kk, nncols) int32, 1-based, 0 = absent, column 0 = count
    iz: np.ndarray            # (kk,) int32 1-based type
    irec: np.ndarray          # (nrec,) int32 1-based recursion sites
    nmax: int = 0
    ntype: int = 1
    ijpair: np.ndarray = None  # (njij, 2) atom pairs for recur_b_ij (lattice%ijpair)
    cr: np.ndarray = None      # (3, kk) positions, lattice%cr -- optional locality hint for the engine

    @property
    def kk(self):
        return self.nn.shape[0]

    @property
    def nrec(self):
        return len(self.irec)


@dataclass
class Hamiltonian:
    """hamiltonian%ee, %lsham, %eeo, %enim, %hall, %hallo, %hoh (hamiltonian.f90:51-70)."""
    ee: np.ndarray
    lsham: np.ndarray
    eeo: np.ndarray = None
    enim: np.ndarray = None
    hall: np.ndarray = None
    hallo: np.ndarray = None
    hoh: bool = False
    local_axis: bool = False


def site_partition(rank, nprocs, nsites):
    """get_mpi_variables (mpi.f90:32-58): 1-based inclusive (start_atom, end_atom)."""
    s, e = C.c_int(), C.c_int()
    _lib.lib().rsrec_site_partition(rank, nprocs, nsites, C.byref(s), C.byref(e))
    return s.value, e.value


def chebyshev_scaling(energy_min, energy_max):
    """a, b of chebyshev_recur (recursion.f90:3078-3079).  The literal 0.3 there is default REAL(4)."""
    a = (energy_max - energy_min) / float(np.float32(2) - np.float32(0.3))
    b = (energy_max + energy_min) / 2
    return a, b


def _fc(a, dtype):
    return np.asfortranarray(np.asarray(a, dtype=dtype))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Recursion:
    """Drop-in counterpart of ``type(recursion)`` (recursion.f90:41-116)."""

    def __init__(self, hamiltonian, lattice, control, energy=None, device=0, rank=0, nprocs=1):
        self.hamiltonian, self.lattice, self.control, self.en = hamiltonian, lattice, control, energy or Energy()
        self.rank, self.nprocs = rank, nprocs
        self._L = _lib.lib()
        self._h = C.c_void_p()
        rc = self._L.rsrec_create(C.byref(self._h), device)
        if rc != 0:
            raise _lib.RsrecError(rc, "rsrec_create failed (no usable gfx950 device? there is no CPU fallback)")
        self.restore_to_default()
        self.update_lattice()
        self.update_hamiltonian()

    # -- lifetime --------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.rsrec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            buf = C.create_string_buffer(512)
            self._L.rsrec_last_error(self._h, buf, 512)
            # the reference calls g_logger%fatal here (recursion.f90:1942, :2595)
            raise _lib.RsrecError(rc, buf.value.decode(errors="replace"))

    def set_option(self, key, value):
        self._check(self._L.rsrec_set_option(self._h, key.encode(), int(value)))


    def timing(self):
        out = (C.c_double * 12)()
        self._L.rsrec_get_timing(self._h, out, 12)
        keys = ("total_ms", "hop_ms", "hop_launches", "atom_steps", "block_multiplies", "rest_ms", "host_ms", "hop_fuses_a", "hop_mfma_flop", "hop_required_flop",
                "operator_arrays_from_device", "octet_launches")
        return dict(zip(keys, list(out)))

    # -- state (restore_to_default, recursion.f90:3713-3825) --------------------------------------------
    def restore_to_default(self):
        lat, ctl = self.lattice, self.control
        llmax = max(ctl.llsp, ctl.lld)
        nrec = lat.nrec
        njij = 0 if lat.ijpair is None else len(lat.ijpair)
        nsites = nrec if njij == 0 else 4 * njij
        self.a = np.zeros((llmax, 18, nrec, 3), np.float64, order="F")
        self.b2 = np.zeros((llmax, 18, nrec, 3), np.float64, order="F")
        self.a_b = np.zeros((18, 18, ctl.lld, nsites), np.complex128, order="F")
        self.b2_b = np.zeros((18, 18, ctl.lld, nsites), np.complex128, order="F")
        self.mu_n = np.zeros((18, 18, 2 * ctl.lld + 2, nsites), np.complex128, order="F")
        self.mu_ng = np.zeros_like(self.mu_n)

    def update_lattice(self):
        lat = self.lattice
        self._nn = _fc(lat.nn, np.int32)
        self._iz = _fc(lat.iz, np.int32)
        self._check(self._L.rsrec_set_lattice(self._h, lat.kk, self._nn.shape[1], _ptr(self._nn), _ptr(self._iz), int(lat.nmax), int(lat.ntype)))
        if lat.cr is not None:
            cr = _fc(lat.cr, np.float64)
            assert cr.shape == (3, lat.kk)
            self._check(self._L.rsrec_set_positions(self._h, _ptr(cr)))

    def update_hamiltonian(self):
        """Must be called whenever the caller rebuilt the blocks (self.f90:777-797 does before every recur*)."""
        ham = self.hamiltonian
        keep = {}
        for k in ("ee", "lsham", "eeo", "enim", "hall", "hallo"):
            v = getattr(ham, k)
            keep[k] = None if v is None else _fc(v, np.complex128)
        self._ham_keep = keep
        self._check(self._L.rsrec_set_hamiltonian(self._h, keep["ee"].shape[2], int(bool(ham.hoh)), int(self.control.nsp),
                                                  _ptr(keep["ee"]), _ptr(keep["lsham"]), _ptr(keep["eeo"]), _ptr(keep["enim"]),
                                                  _ptr(keep["hall"]), _ptr(keep["hallo"])))

    def _assemble(self, part, hmag, nbr_type, obarm):
        hoh = bool(self.hamiltonian.hoh)
        hm = _fc(hmag, np.complex128)
        assert hm.ndim == 5 and hm.shape[:2] == (9, 9) and hm.shape[3] == 4, "hmag is (9,9,nslots,4,ncls)"
        nslots, ncls = hm.shape[2], hm.shape[4]
        ty = ob = None
        ntype = 0
        if hoh:
            ty, ob = _fc(nbr_type, np.int32), _fc(obarm, np.complex128)
            assert ty.shape == (nslots, ncls) and ob.shape[:2] == (18, 18)
            ntype = ob.shape[2]
        blocks = np.zeros((18, 18, nslots, ncls), np.complex128, order="F")
        blocks_o = np.zeros_like(blocks) if hoh else None
        self._check(self._L.rsrec_assemble_blocks(self._h, part, ncls, nslots, int(hoh), _ptr(hm), _ptr(ty), _ptr(ob), ntype, _ptr(blocks), _ptr(blocks_o)))
        return blocks, blocks_o

    def build_bulkham(self, hmag, nbr_type=None, obarm=None):
        """hamiltonian%build_bulkham after chbar_nc (hamiltonian.f90:1553-1616), on the device: ee (and eeo = ee.obar with hoh) of every atom
        type from the (Hx, Hy, Hz, H0) parts ``hmag`` (9,9,nslots,4,ntype); fills ``self.hamiltonian.ee / .eeo``.  A following
        update_hamiltonian() finds the blocks on the device."""
        self.hamiltonian.ee, eeo = self._assemble(0, hmag, nbr_type, obarm)
        if eeo is not None:
            self.hamiltonian.eeo = eeo

    def build_locham(self, hmag, nbr_type=None, obarm=None):
        """hamiltonian%build_locham (hamiltonian.f90:1618-1667): hall / hallo of the first nmax atoms, as build_bulkham."""
        self.hamiltonian.hall, hallo = self._assemble(1, hmag, nbr_type, obarm)
        if hallo is not None:
            self.hamiltonian.hallo = hallo

    def _my_sites(self):
        start, end = site_partition(self.rank, self.nprocs, self.lattice.nrec)   # recursion.f90:1816
        return start, end, np.ascontiguousarray(self.lattice.irec[start - 1:end], dtype=np.int32)

    # -- drivers ---------------------------------------------------------------------------------------
    def recur_b(self):
        """Block Lanczos for the sites this rank owns (recursion.f90:1807-1866)."""
        lld = self.control.lld
        start, end, seeds = self._my_sites()
        n = len(seeds)
        a_b = np.zeros((18, 18, lld, n), np.complex128, order="F")
        b2_b = np.zeros_like(a_b)
        self._check(self._L.rsrec_block_lanczos(self._h, n, _ptr(seeds), lld, _ptr(a_b), _ptr(b2_b)))
        self.a_b[:, :, :, :n] = a_b                                   # index i - start_atom + 1 (:1847-1848)
        self.b2_b[:, :, :, :n] = b2_b
        d = np.arange(18)
        self.a[:lld, :, :n, 0] = a_b[d, d].real.transpose(1, 0, 2)    # :1850-1851
        self.b2[:lld, :, :n, 0] = b2_b[d, d].real.transpose(1, 0, 2)

    def recur_b_local_axis(self, rot):
        """recur_b with hamiltonian%local_axis = T (recursion.f90:1830-1832): the Hamiltonian object holds the GLOBAL-frame blocks
        (ee_glob, ...), ``rot`` (18,18,nrec) the spin-frame rotation of every site; all sites go in one batched call."""
        lld = self.control.lld
        start, end, seeds = self._my_sites()
        n = len(seeds)
        r = _fc(np.asarray(rot)[:, :, start - 1:end], np.complex128)
        a_b = np.zeros((18, 18, lld, n), np.complex128, order="F")
        b2_b = np.zeros_like(a_b)
        self._check(self._L.rsrec_block_lanczos_local_axis(self._h, n, _ptr(seeds), _ptr(r), lld, _ptr(a_b), _ptr(b2_b)))
        self.a_b[:, :, :, :n] = a_b
        self.b2_b[:, :, :, :n] = b2_b
        d = np.arange(18)
        self.a[:lld, :, :n, 0] = a_b[d, d].real.transpose(1, 0, 2)
        self.b2[:lld, :, :n, 0] = b2_b[d, d].real.transpose(1, 0, 2)

    def pack_diag(self, site_offset, nsites_total, a_img, b2_img):
        """This rank's part of the zero-padded (lld, 18, nsites_total) images of a / b2 that the ranks all-reduce
        (bands.f90:271-274), written from the coefficients resident on the device.  a_img / b2_img: numpy arrays or raw
        (device) addresses, e.g. ``tensor.data_ptr()`` of the CUDA tensor handed to the collective."""
        pa = a_img if isinstance(a_img, int) else a_img.ctypes.data
        pb = b2_img if isinstance(b2_img, int) else b2_img.ctypes.data
        self._check(self._L.rsrec_pack_diag(self._h, int(site_offset), int(nsites_total), C.c_void_p(pa), C.c_void_p(pb)))

    def pack_moments(self, site_offset, nsites_total, mu_img):
        """The Chebyshev counterpart: mu_n(18,18,2 lld + 2,site) of this rank's sites inside a zero image over all sites, written from
        the moments resident on the device.  mu_img: numpy array or raw (device) address."""
        pm = mu_img if isinstance(mu_img, int) else mu_img.ctypes.data
        self._check(self._L.rsrec_pack_moments(self._h, int(site_offset), int(nsites_total), C.c_void_p(pm)))

    # -- library-level communicator (RCCL bound by librsrec itself: no MPI, no torch) -----------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        if _lib.lib().rsrec_comm_unique_id(buf) != 0:
            raise _lib.RsrecError(2, "rsrec_comm_unique_id failed (RCCL not available?)")
        return buf.raw

    def comm_init(self, rank, nranks, ident=None, path=None, timeout_s=60.0):
        """Collective: every rank passes the same 128-byte id (from comm_unique_id on one rank), or a `path` through which rank 0
        publishes it."""
        if path is not None:
            self._check(self._L.rsrec_comm_init_file(self._h, int(rank), int(nranks), os.fsencode(path), float(timeout_s)))
        else:
            self._check(self._L.rsrec_comm_init(self._h, int(rank), int(nranks), ident))

    def comm_size(self):
        r, n = C.c_int(), C.c_int()
        self._check(self._L.rsrec_comm_size(self._h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def allreduce_sum(self, buf, n=None):
        """In-place sum over the ranks (bands.f90:271-274): numpy float64 array, or a raw device address with `n` doubles."""
        if isinstance(buf, int):
            self._check(self._L.rsrec_allreduce_sum(self._h, C.c_void_p(buf), int(n)))
        else:
            # the images this mirror reduces are float64 (diagonals, LDOS) or complex128 (a_b, mu_n): a complex array counts as 2 doubles per element
            if buf.dtype not in (np.float64, np.complex128):
                raise TypeError("allreduce_sum: float64 or complex128 arrays only, got %s" % buf.dtype)
            if not (buf.flags["C_CONTIGUOUS"] or buf.flags["F_CONTIGUOUS"]):
                raise ValueError("allreduce_sum: the array must be contiguous (it is reduced in place)")
            self._check(self._L.rsrec_allreduce_sum(self._h, C.c_void_p(buf.ctypes.data), buf.nbytes // 8))

    def recur_b_ij(self):
        """Four chains per atom pair, seeds (psi_i +- psi_j)/sqrt2 and (psi_i +- i psi_j)/sqrt2 (recursion.f90:1655-1800)."""
        lld = self.control.lld
        pairs = np.asarray(self.lattice.ijpair, dtype=np.int32)
        njij = len(pairs)
        start, end = site_partition(self.rank, self.nprocs, njij)
        slots, sa, sc = self._pair_seeds(pairs[start - 1:end], skip_diagonal_repeats=True)
        nch = len(slots)
        a_b = np.zeros((18, 18, lld, nch), np.complex128, order="F")
        b2_b = np.zeros_like(a_b)
        self._check(self._L.rsrec_block_lanczos_seeded(self._h, nch, 2, _ptr(sa), _ptr(sc), lld, _ptr(a_b), _ptr(b2_b)))
        self.a_b[:, :, :, slots] = a_b
        self.b2_b[:, :, :, slots] = b2_b

    @staticmethod
    def _pair_seeds(pairs, skip_diagonal_repeats):
        """(asign, bsign) of the four chains of every pair (recursion.f90:1679-1707 / :2403-2448); slot = ij_loc*4 - 4 + reci."""
        s2 = 1.0 / np.sqrt(2.0)          # one_over_sqrt_two (math.f90)
        slots, seeds, coefs = [], [], []
        for ij_loc, (i, j) in enumerate(pairs):
            for reci in range(4):
                a, b = s2, (s2, -s2, 1j * s2, -1j * s2)[reci]
                if i == j and skip_diagonal_repeats:
                    if reci > 0:
                        continue                                      # recur_b_ij :1705-1706 `cycle`: slots 2..4 stay zero
                    a = b = 1.0                                       # :1702-1704
                seeds.append((i, j)); coefs.append((a, b))            # assigned in order: for i == j the second value stays
                slots.append(ij_loc * 4 + reci)
        return slots, np.ascontiguousarray(seeds, dtype=np.int32), np.ascontiguousarray(coefs, dtype=np.complex128)

    def chebyshev_recur_ij(self):
        """Chebyshev moments of the four chains per pair (recursion.f90:2376-2487); no i == j special case there."""
        lld = self.control.lld
        a, b = chebyshev_scaling(self.en.energy_min, self.en.energy_max)
        pairs = np.asarray(self.lattice.ijpair, dtype=np.int32)
        start, end = site_partition(self.rank, self.nprocs, len(pairs))
        slots, sa, sc = self._pair_seeds(pairs[start - 1:end], skip_diagonal_repeats=False)
        nch = len(slots)
        mu = np.zeros((18, 18, 2 * lld + 2, nch), np.complex128, order="F")
        self._check(self._L.rsrec_chebyshev_seeded(self._h, nch, 2, _ptr(sa), _ptr(sc), lld, a, b, _ptr(mu)))
        self.mu_n[:, :, :, slots] = mu

    def zsqr(self):
        """b2_b <- sqrt(b2_b) in place (recursion.f90:1980-2023)."""
        self._check(self._L.rsrec_zsqr(self._h, self.b2_b.shape[2] * self.b2_b.shape[3], _ptr(self.b2_b)))

    def chebyshev_recur(self):
        """Chebyshev moments for the sites this rank owns (recursion.f90:3057-3130)."""
        lld = self.control.lld
        a, b = chebyshev_scaling(self.en.energy_min, self.en.energy_max)
        start, end, seeds = self._my_sites()
        n = len(seeds)
        mu = np.zeros((18, 18, 2 * lld + 2, n), np.complex128, order="F")
        self._check(self._L.rsrec_chebyshev(self._h, n, _ptr(seeds), lld, a, b, _ptr(mu)))
        self.mu_n[:, :, :, :n] = mu

    def compute_moments_stochastic(self, v_a, v_b, cond_ll, vo_a=None, vo_b=None, seeds=None, coefs=None, atlist=None):
        """Kubo-Bastin double moments mu_nm_stochastic(18,18,cond_ll,cond_ll,nvec) (recursion.f90:979-1234).
        ``cond_calctype='per_type'``: pass ``atlist`` (lattice%atlist, one seed atom per type).  Random vectors: pass ``seeds``
        (nvec, nseed) atoms and ``coefs`` (nvec, nseed) complex (the caller owns the random numbers)."""
        a, b = chebyshev_scaling(self.en.energy_min, self.en.energy_max)
        if seeds is None:
            seeds = np.asarray(atlist, dtype=np.int32).reshape(-1, 1)
            coefs = np.ones(seeds.shape, np.complex128)
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        coefs = np.ascontiguousarray(coefs, dtype=np.complex128)
        nvec, nseed = seeds.shape
        mu = np.zeros((18, 18, cond_ll, cond_ll, nvec), np.complex128, order="F")
        keep = [None if v is None else _fc(v, np.complex128) for v in (v_a, vo_a, v_b, vo_b)]
        self._check(self._L.rsrec_kubo_moments(self._h, nvec, nseed, _ptr(seeds), _ptr(coefs), int(cond_ll), a, b,
                                               _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), _ptr(mu)))
        self.mu_nm_stochastic = mu
        return mu

    def ham_vec_matmul(self, psi_in, a, b):
        """psi_out = (H psi_in - b psi_in)/a on a whole vector psi(18,18,kk) with the PLAIN operator ee + l.s, whatever hoh says
        (recursion.f90:913-977)."""
        x = _fc(psi_in, np.complex128)
        out = np.zeros_like(x)
        self._check(self._L.rsrec_apply_operator(self._h, 2, None, None, _ptr(x), _ptr(out), float(a), float(b)))
        return out

    def ham_hoh_vec_matmul(self, psi_in, a, b):
        """The same with H = h - h o h + e_nu + l.s (recursion.f90:785-911); needs hamiltonian%hoh."""
        x = _fc(psi_in, np.complex128)
        out = np.zeros_like(x)
        self._check(self._L.rsrec_apply_operator(self._h, 0, None, None, _ptr(x), _ptr(out), float(a), float(b)))
        return out

    def chebyshev_orbital_mod(self, cr, alat, seeds=None, per_seed=False):
        """The moments of chebyshev_orbital_mod (recursion.f90:2834-3049): mu_n_orb(18,18,lld) = (1/kk) sum over all atoms as seeds
        (`seeds` = a subset: the plain sum over it, not divided), device-resident.  per_seed: also every seed's contribution."""
        lld = self.control.lld
        a, b = chebyshev_scaling(self.en.energy_min, self.en.energy_max)
        kk = self.lattice.kk
        sd = np.arange(1, kk + 1, dtype=np.int32) if seeds is None else np.ascontiguousarray(seeds, dtype=np.int32)
        crf = _fc(cr, np.float64)
        assert crf.shape == (3, kk)
        mu = np.zeros((18, 18, lld), np.complex128, order="F")
        ms = np.zeros((18, 18, lld, len(sd)), np.complex128, order="F") if per_seed else None
        self._check(self._L.rsrec_orbital_moments(self._h, len(sd), _ptr(sd), lld, a, b, _ptr(crf), float(alat), _ptr(mu), _ptr(ms)))
        if seeds is None:
            mu = mu / float(kk)                                   # :3006
        return (mu, ms) if per_seed else mu

    def velo_vec_matmul(self, v_op, psi_in, vo_op=None):
        """psi_out = V psi_in (recursion.f90:587; :656 with hoh)."""
        x = _fc(psi_in, np.complex128)
        out = np.zeros_like(x)
        v = _fc(v_op, np.complex128)
        vo = None if vo_op is None else _fc(vo_op, np.complex128)
        self._check(self._L.rsrec_apply_operator(self._h, 1, _ptr(v), _ptr(vo), _ptr(x), _ptr(out), 1.0, 0.0))
        return out

    def recur(self):
        """Scalar Haydock recursion, 18 orbital chains per site (recursion.f90:3485-3532)."""
        lld = self.control.lld
        llmax = self.a.shape[0]
        start, end, seeds = self._my_sites()
        n = len(seeds)
        a = np.zeros((llmax, 18, n), np.float64, order="F")
        b2 = np.zeros_like(a)
        self._check(self._L.rsrec_scalar_lanczos(self._h, n, _ptr(seeds), lld, llmax, _ptr(a), _ptr(b2)))
        self.a[:, :, :n, 0] = a
        self.b2[:, :, :n, 0] = b2
