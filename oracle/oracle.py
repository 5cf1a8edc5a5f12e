"""TEST INFRASTRUCTURE ONLY: ctypes front-end of the CPU oracle (oracle/rsrec_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It is the checker for the HIP path and is never used to produce a product result.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "librsrec_oracle.so")


class _Problem(C.Structure):
    _fields_ = [("kk", C.c_int), ("nncols", C.c_int), ("nmax", C.c_int), ("ntype", C.c_int), ("nslots", C.c_int),
                ("hoh", C.c_int), ("nsp", C.c_int),
                ("nn", C.c_void_p), ("iz", C.c_void_p),
                ("ee", C.c_void_p), ("lsham", C.c_void_p), ("eeo", C.c_void_p), ("enim", C.c_void_p),
                ("hall", C.c_void_p), ("hallo", C.c_void_p)]


def build(force=False):
    src = os.path.join(_HERE, "rsrec_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "librsrec_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _f(a, dtype):
    return np.asfortranarray(np.asarray(a, dtype=dtype))


class DiagonalizationError(RuntimeError):
    """The reference's fatal exit of crecal_b (recursion.f90:1942): zheev fails on the NaN matrix a Krylov breakdown leaves behind.
    .a_b / .b2_b hold what was computed (NaN from the failing level on for the chains that broke down)."""
    def __init__(self, msg, a_b, b2_b):
        super().__init__(msg)
        self.a_b, self.b2_b = a_b, b2_b


class Oracle:
    """Holds one problem (lattice tables + Hamiltonian blocks) for the CPU oracle."""

    def __init__(self, p):
        self.keep = {}
        nn = self.keep["nn"] = _f(p["nn"], np.int32)
        self.kk, self.nncols = nn.shape
        ee = self.keep["ee"] = _f(p["ee"], np.complex128)
        self.nslots, self.ntype = ee.shape[2], ee.shape[3]
        self.nmax = int(p.get("nmax", 0))
        self.hoh = int(p.get("hoh", 0))
        self.nsp = int(p.get("nsp", 2))
        self.keep["iz"] = _f(p["iz"], np.int32)
        zt = np.zeros((18, 18, self.ntype), np.complex128, order="F")
        self.keep["lsham"] = _f(p.get("lsham", zt), np.complex128)
        self.keep["enim"] = _f(p.get("enim", zt), np.complex128)
        self.keep["eeo"] = _f(p.get("eeo", np.zeros_like(ee)), np.complex128)
        if self.nmax > 0:
            self.keep["hall"] = _f(p["hall"], np.complex128)
            self.keep["hallo"] = _f(p.get("hallo", np.zeros_like(self.keep["hall"])), np.complex128)
        ptr = lambda k: self.keep[k].ctypes.data if k in self.keep else None
        self.P = _Problem(self.kk, self.nncols, self.nmax, self.ntype, self.nslots, self.hoh, self.nsp,
                          ptr("nn"), ptr("iz"), ptr("ee"), ptr("lsham"), ptr("eeo"), ptr("enim"), ptr("hall"), ptr("hallo"))

    def block_lanczos(self, seeds, lld, fatal_ok=False):
        """recur_b.  A chain whose Krylov space is exhausted makes the reference call g_logger%fatal('Diagonalization error')
        (recursion.f90:1942): DiagonalizationError here, unless fatal_ok (then that chain's coefficients are NaN from the failing level on)."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        n = len(seeds)
        a_b = np.zeros((18, 18, lld, n), np.complex128, order="F")
        b2_b = np.zeros_like(a_b)
        rc = lib().orc_block_lanczos(C.byref(self.P), n, seeds.ctypes.data_as(C.c_void_p), lld,
                                     a_b.ctypes.data_as(C.c_void_p), b2_b.ctypes.data_as(C.c_void_p))
        if rc == 2 and not fatal_ok:
            raise DiagonalizationError("Diagonalization error (recursion.f90:1942)", a_b, b2_b)
        assert rc in (0, 2)
        return a_b, b2_b

    def block_lanczos_seeded(self, seeds, coefs, lld, fatal_ok=False):
        """seeds (nchains, nseed) int, coefs (nchains, nseed) complex"""
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        coefs = np.ascontiguousarray(coefs, dtype=np.complex128)
        n, ns = seeds.shape
        a_b = np.zeros((18, 18, lld, n), np.complex128, order="F")
        b2_b = np.zeros_like(a_b)
        rc = lib().orc_block_lanczos_seeded(C.byref(self.P), n, ns, seeds.ctypes.data_as(C.c_void_p), coefs.ctypes.data_as(C.c_void_p), lld,
                                            a_b.ctypes.data_as(C.c_void_p), b2_b.ctypes.data_as(C.c_void_p))
        if rc == 2 and not fatal_ok:
            raise DiagonalizationError("Diagonalization error (recursion.f90:1942)", a_b, b2_b)
        assert rc in (0, 2)
        return a_b, b2_b

    def chebyshev_seeded(self, seeds, coefs, lld, a, b):
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        coefs = np.ascontiguousarray(coefs, dtype=np.complex128)
        n, ns = seeds.shape
        mu = np.zeros((18, 18, 2 * lld + 2, n), np.complex128, order="F")
        rc = lib().orc_chebyshev_seeded(C.byref(self.P), n, ns, seeds.ctypes.data_as(C.c_void_p), coefs.ctypes.data_as(C.c_void_p), lld,
                                        C.c_double(a), C.c_double(b), mu.ctypes.data_as(C.c_void_p))
        return mu, rc

    def chebyshev(self, seeds, lld, a, b):
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        n = len(seeds)
        mu = np.zeros((18, 18, 2 * lld + 2, n), np.complex128, order="F")
        rc = lib().orc_chebyshev(C.byref(self.P), n, seeds.ctypes.data_as(C.c_void_p), lld, C.c_double(a), C.c_double(b),
                                 mu.ctypes.data_as(C.c_void_p))
        return mu, rc

    def kubo_moments(self, seeds, coefs, cond_ll, a, b, v_a, v_b, vo_a=None, vo_b=None):
        """compute_moments_stochastic (recursion.f90:979): seeds (nvec, nseed) int, coefs (nvec, nseed) complex -> mu (18,18,cond_ll,cond_ll,nvec)."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        coefs = np.ascontiguousarray(coefs, dtype=np.complex128)
        nvec, ns = seeds.shape
        mu = np.zeros((18, 18, cond_ll, cond_ll, nvec), np.complex128, order="F")
        keep = [None if v is None else _f(v, np.complex128) for v in (v_a, vo_a, v_b, vo_b)]
        p = lambda x: None if x is None else x.ctypes.data_as(C.c_void_p)
        L = lib()
        L.orc_kubo_moments.restype = C.c_int
        L.orc_kubo_moments.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 5
        rc = L.orc_kubo_moments(C.byref(self.P), nvec, ns, p(seeds), p(coefs), int(cond_ll), float(a), float(b), p(keep[0]), p(keep[1]), p(keep[2]), p(keep[3]), p(mu))
        assert rc == 0
        return mu

    def orbital_moments(self, seeds, lld, a, b, cr, alat, per_seed=False):
        """chebyshev_orbital_mod (recursion.f90:2834), moment part: sum over `seeds` of sum_k left_k^H T_{n-1}(H~) r  -> (18,18,lld)
        [per_seed: and the per-seed contributions (18,18,lld,nseeds) and the sums (3,nseeds) of left_vec, left_vec1, left_vec2 the reference prints]."""
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        crf = _f(cr, np.float64)
        mu = np.zeros((18, 18, lld), np.complex128, order="F")
        ms = np.zeros((18, 18, lld, len(seeds)), np.complex128, order="F") if per_seed else None
        sums = np.zeros((3, len(seeds)), np.complex128, order="F") if per_seed else None
        L = lib()
        L.orc_orbital_moments.restype = C.c_int
        L.orc_orbital_moments.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        rc = L.orc_orbital_moments(C.byref(self.P), len(seeds), seeds.ctypes.data_as(C.c_void_p), int(lld), float(a), float(b), crf.ctypes.data_as(C.c_void_p),
                                   float(alat), mu.ctypes.data_as(C.c_void_p), None if ms is None else ms.ctypes.data_as(C.c_void_p),
                                   None if sums is None else sums.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return (mu, ms, sums) if per_seed else mu

    def scalar_lanczos(self, seeds, lld, llmax=None):
        seeds = np.ascontiguousarray(seeds, dtype=np.int32)
        n = len(seeds)
        llmax = llmax or lld
        a = np.zeros((llmax, 18, n), np.float64, order="F")
        b2 = np.zeros_like(a)
        rc = lib().orc_scalar_lanczos(C.byref(self.P), n, seeds.ctypes.data_as(C.c_void_p), lld, llmax,
                                      a.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return a, b2


def zsqr(b2_b):
    out = np.array(b2_b, dtype=np.complex128, order="F", copy=True)
    nmat = out.size // 324
    lib().orc_zsqr(nmat, out.ctypes.data_as(C.c_void_p))
    return out


def block_green(a_b, b_sqrt, ene, a_inf, b_inf, eta=0.0 + 0.0j, sym_term=False):
    """green%bgreen (green.f90:1191) for ONE site: a_b, b_sqrt (18,18,lld) -> g0 (18,18,len(ene))."""
    a_b = np.asfortranarray(a_b, dtype=np.complex128); b_sqrt = np.asfortranarray(b_sqrt, dtype=np.complex128)
    ene = np.ascontiguousarray(ene, dtype=np.float64)
    a_inf = np.asfortranarray(a_inf, dtype=np.float64); b_inf = np.asfortranarray(b_inf, dtype=np.float64)
    g0 = np.zeros((18, 18, len(ene)), dtype=np.complex128, order="F")
    L = lib()
    L.orc_block_green.restype = C.c_int
    L.orc_block_green.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_int] + [C.c_void_p] * 5
    rc = L.orc_block_green(a_b.shape[2], len(ene), ene.ctypes.data_as(C.c_void_p), float(np.real(eta)), float(np.imag(eta)), int(bool(sym_term)),
                           a_inf.ctypes.data_as(C.c_void_p), b_inf.ctypes.data_as(C.c_void_p), a_b.ctypes.data_as(C.c_void_p),
                           b_sqrt.ctypes.data_as(C.c_void_p), g0.ctypes.data_as(C.c_void_p))
    assert rc == 0, "singular matrix in the continued fraction"
    return g0


def terminator(a_b, b_sqrt):
    """recursion%get_terminf (recursion.f90:2092): a_b, b_sqrt (18,18,lld,nsites) -> a_inf, b_inf (18,18,nsites), a_inf0, b_inf0 (nsites)."""
    a_b = np.asfortranarray(a_b, dtype=np.complex128); b_sqrt = np.asfortranarray(b_sqrt, dtype=np.complex128)
    lld, n = a_b.shape[2], a_b.shape[3]
    a_inf = np.zeros((18, 18, n), order="F"); b_inf = np.zeros_like(a_inf)
    a0, b0 = np.zeros(n), np.zeros(n)
    L = lib()
    L.orc_terminator.restype = C.c_int
    L.orc_terminator.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 6
    rc = L.orc_terminator(n, lld, a_b.ctypes.data_as(C.c_void_p), b_sqrt.ctypes.data_as(C.c_void_p), a_inf.ctypes.data_as(C.c_void_p),
                          b_inf.ctypes.data_as(C.c_void_p), a0.ctypes.data_as(C.c_void_p), b0.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return a_inf, b_inf, a0, b0


def scalar_density(a, b2, ene, dw_l, cshi, lld=None):
    """dos%density (density_of_states.f90:248-363; bprldos :370-404): a, b2 (llmax,18,nsites[,nmdir]) of the scalar recursion ->
    tdens (18, len(ene), nsites, nmdir); dw_l, cshi (18, nsites)."""
    a = np.asfortranarray(a, dtype=np.float64); b2 = np.asfortranarray(b2, dtype=np.float64)
    if a.ndim == 3:
        a = a[:, :, :, None]; b2 = b2[:, :, :, None]
    a = np.asfortranarray(a); b2 = np.asfortranarray(b2)
    llmax, _, n, nmd = a.shape
    ene = np.ascontiguousarray(ene, dtype=np.float64)
    dw_l = np.asfortranarray(dw_l, dtype=np.float64); cshi = np.asfortranarray(cshi, dtype=np.float64)
    out = np.zeros((18, len(ene), n, nmd), order="F")
    L = lib()
    L.orc_scalar_density.restype = C.c_int
    L.orc_scalar_density.argtypes = [C.c_int] * 4 + [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 4
    rc = L.orc_scalar_density(n, nmd, llmax, int(lld or llmax), a.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), len(ene), ene.ctypes.data_as(C.c_void_p),
                              dw_l.ctypes.data_as(C.c_void_p), cshi.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def sgreen_from_density(tdens):
    """green%sgreen for control%nmdir = 1 (green.f90:676-683): g0(j,j,ie,ia) = -i pi doso(j,ie); tdens (18, nen, nsites[, 1]) -> g0 (18,18,nen,nsites)."""
    t = tdens[:, :, :, 0] if tdens.ndim == 4 else tdens
    g0 = np.zeros((18, 18) + t.shape[1:], np.complex128, order="F")
    for j in range(18):
        g0[j, j] = -1j * t[j] * np.pi
    return g0


def chebyshev_green(mu_n, ene, emin, emax):
    """green%chebyshev_green (green.f90:1030) for ONE site: mu_n (18,18,2lld+2) -> g0 (18,18,len(ene))."""
    mu_n = np.asfortranarray(mu_n, dtype=np.complex128)
    ene = np.ascontiguousarray(ene, dtype=np.float64)
    g0 = np.zeros((18, 18, len(ene)), dtype=np.complex128, order="F")
    L = lib()
    L.orc_chebyshev_green.restype = C.c_int
    L.orc_chebyshev_green.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    L.orc_chebyshev_green((mu_n.shape[2] - 2) // 2, len(ene), ene.ctypes.data_as(C.c_void_p), float(emin), float(emax),
                          mu_n.ctypes.data_as(C.c_void_p), g0.ctypes.data_as(C.c_void_p))
    return g0


def site_partition(rank, nprocs, nsites):
    s, e = C.c_int(), C.c_int()
    lib().orc_site_partition(rank, nprocs, nsites, C.byref(s), C.byref(e))
    return s.value, e.value


def assemble_blocks(hmag, nbr_type=None, obarm=None):
    """build_bulkham / build_locham after chbar_nc (hamiltonian.f90:1565-1570, :1599 / :1631-1636, :1654), restated with numpy (checker only):
    blocks = [[H0 + Hz, Hx - i Hy], [Hx + i Hy, H0 - Hz]] from hmag (9,9,nslots,4,ncls) = (Hx, Hy, Hz, H0); blocks_o = blocks . obarm(type
    behind the slot), zero for empty slots.  Pinned by tests/golden/*_hmag.npz against the reference's own ee / eeo / hall / hallo."""
    hm = np.asarray(hmag)
    nsl, ncls = hm.shape[2], hm.shape[4]
    b = np.zeros((18, 18, nsl, ncls), np.complex128, order="F")
    b[:9, :9] = hm[:, :, :, 3] + hm[:, :, :, 2]
    b[9:, 9:] = hm[:, :, :, 3] - hm[:, :, :, 2]
    b[:9, 9:] = hm[:, :, :, 0] - 1j * hm[:, :, :, 1]
    b[9:, :9] = hm[:, :, :, 0] + 1j * hm[:, :, :, 1]
    if obarm is None:
        return b, None
    bo = np.zeros_like(b)
    for c in range(ncls):
        for m in range(nsl):
            t = int(nbr_type[m, c])
            if t > 0:
                bo[:, :, m, c] = b[:, :, m, c] @ obarm[:, :, t - 1]
    return b, bo
