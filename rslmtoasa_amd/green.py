"""Host-side mirror of the reference's ``green`` type for the block-recursion Green function (green.f90:42-72, :588-621, :1191-1339).

Only the stage that sits directly on the recursion's output is here: ``block_green`` turns the block coefficients of the
sites this rank owns into ``g0(18,18,nE,site)`` on the GPU (``rsrec_block_green``).  The terminator ``a_inf, b_inf`` comes
from ``recursion%get_terminf`` (recursion.f90:2092), which stays on the CPU in the reference's own code and
is therefore an argument here.
"""
import numpy as np


def _ptr(a):
    import ctypes as C
    return a.ctypes.data_as(C.c_void_p)


class Green:
    def __init__(self, recursion, ene, sym_term=False):
        self.recursion = recursion
        self.ene = np.ascontiguousarray(ene, dtype=np.float64)      # energy%ene(1:channels_ldos+10), energy.f90:205-207
        self.sym_term = bool(sym_term)                              # control%sym_term
        self.g0 = None                                              # green%g0: allocated once, overwritten by every call (green.f90:150-170)

    def _g0_buffer(self, n):
        shape = (18, 18, len(self.ene), n)
        if self.g0 is None or self.g0.shape != shape:
            self.g0 = np.zeros(shape, dtype=np.complex128, order="F")
        return self.g0

    def block_green(self, a_inf, b_inf, eta=0.0 + 0.0j, nsites=None):
        """green%block_green: requires ``recursion.zsqr()`` to have been called (self.f90:829), like the reference."""
        rec = self.recursion
        n = rec.a_b.shape[3] if nsites is None else nsites
        lld = rec.a_b.shape[2]
        a_b = np.asfortranarray(rec.a_b[:, :, :, :n])
        b_s = np.asfortranarray(rec.b2_b[:, :, :, :n])
        a_inf = np.asfortranarray(a_inf, dtype=np.float64)
        b_inf = np.asfortranarray(b_inf, dtype=np.float64)
        assert a_inf.shape == (18, 18, n) and b_inf.shape == (18, 18, n)
        g0 = self._g0_buffer(n)
        rec._check(rec._L.rsrec_block_green(rec._h, n, lld, len(self.ene), _ptr(self.ene), float(np.real(eta)), float(np.imag(eta)),
                                            int(self.sym_term), _ptr(a_inf), _ptr(b_inf), _ptr(a_b), _ptr(b_s), _ptr(g0)))
        return g0

    def density(self, dw_l=None, cshi=None, nsites=None, nmdir=1):
        """dos%density for every site (and direction) of the last scalar recursion (density_of_states.f90:248-363, bprldos :370-404):
        ``recursion.a, recursion.b2 (llmax,18,site[,mdir])`` -> ``tdens (18, nE, site, mdir)`` on the GPU.  ``dw_l, cshi (18, site)`` are the
        potential parameters the reference reads (potential.f90:392-393: 1 and 0 unless set)."""
        rec = self.recursion
        a = np.asarray(rec.a); b2 = np.asarray(rec.b2)
        if a.ndim == 3:
            a = a[:, :, :, None]; b2 = b2[:, :, :, None]
        n = a.shape[2] if nsites is None else nsites
        a = np.asfortranarray(a[:, :, :n, :nmdir], dtype=np.float64); b2 = np.asfortranarray(b2[:, :, :n, :nmdir], dtype=np.float64)   # control%nmdir directions
        llmax, nmd = a.shape[0], a.shape[3]
        dw = np.ones((18, n), order="F") if dw_l is None else np.asfortranarray(dw_l, dtype=np.float64)
        cs = np.zeros((18, n), order="F") if cshi is None else np.asfortranarray(cshi, dtype=np.float64)
        assert dw.shape == (18, n) and cs.shape == (18, n)
        tdens = np.zeros((18, len(self.ene), n, nmd), order="F")
        rec._check(rec._L.rsrec_scalar_density(rec._h, n, nmd, llmax, int(rec.control.lld), _ptr(a), _ptr(b2), len(self.ene), _ptr(self.ene), _ptr(dw), _ptr(cs), _ptr(tdens)))
        self.tdens = tdens
        return tdens

    def sgreen(self, dw_l=None, cshi=None, nsites=None):
        """green%sgreen (green.f90:628-705) for control%nmdir = 1: g0(j,j,ie,ia) = -i pi doso(j,ie) with doso = dos%density on the GPU."""
        t = self.density(dw_l, cshi, nsites)
        assert t.shape[3] == 1, "sgreen: the nmdir = 3 combination (green.f90:684-699) stays the reference's host loop over density's output"
        g0 = self._g0_buffer(t.shape[2])
        g0[...] = 0.0
        for j in range(18):
            g0[j, j] = -1j * t[j, :, :, 0] * np.pi
        return g0

    def terminator(self, nsites=None):
        """recursion%get_terminf (recursion.f90:2092) on the GPU for the coefficients held by the recursion object (b2_b after zsqr):
        returns a_inf, b_inf (18,18,n) and the mean diagonals a_inf0, b_inf0 (n)."""
        rec = self.recursion
        n = rec.a_b.shape[3] if nsites is None else nsites
        lld = rec.a_b.shape[2]
        a_b = np.asfortranarray(rec.a_b[:, :, :, :n])
        b_s = np.asfortranarray(rec.b2_b[:, :, :, :n])
        a_inf = np.zeros((18, 18, n), np.float64, order="F")
        b_inf = np.zeros_like(a_inf)
        a0, b0 = np.zeros(n), np.zeros(n)
        rec._check(rec._L.rsrec_terminator(rec._h, n, lld, _ptr(a_b), _ptr(b_s), _ptr(a_inf), _ptr(b_inf), _ptr(a0), _ptr(b0)))
        return a_inf, b_inf, a0, b0

    def block_ldos(self, eta=0.0 + 0.0j, site_offset=0, nsites_total=None, out=None):
        """The LDOS stage for the sites of the last ``recur_b`` call, entirely on the device from the coefficients that call left
        there: zsqr -> get_terminf -> bgreen -> the reduction of bands%calculate_fermi (bands.f90:258-268).  Returns a dict with the
        zero-padded images ``dtot(nen)``, ``dosia(nsites_total, nen)``, ``dosial(nsites_total, 18, nen)`` and the terminators used.
        ``out`` = (dtot, dosia, dosial) raw DEVICE addresses (e.g. ``tensor.data_ptr()``): the images are written there in place."""
        import ctypes as C
        rec = self.recursion
        start, end = rec._my_sites()[:2]
        n = end - start + 1
        ntot = n + site_offset if nsites_total is None else nsites_total
        nen = len(self.ene)
        a_inf = np.zeros((18, 18, n), np.float64, order="F")
        b_inf = np.zeros_like(a_inf)
        if out is None:
            dtot = np.zeros(nen)
            dosia = np.zeros((ntot, nen), order="F")
            dosial = np.zeros((ntot, 18, nen), order="F")
            ptrs = (_ptr(dtot), _ptr(dosia), _ptr(dosial))
        else:
            dtot = dosia = dosial = None
            ptrs = tuple(C.c_void_p(int(p)) for p in out)
        rec._check(rec._L.rsrec_block_ldos(rec._h, nen, _ptr(self.ene), float(np.real(eta)), float(np.imag(eta)), int(self.sym_term),
                                           int(site_offset), int(ntot), ptrs[0], ptrs[1], ptrs[2], _ptr(a_inf), _ptr(b_inf)))
        return dict(dtot=dtot, dosia=dosia, dosial=dosial, a_inf=a_inf, b_inf=b_inf)

    def chebyshev_green(self, nsites=None):
        """green%chebyshev_green (green.f90:1030-1108): g0 from the Chebyshev moments ``recursion.mu_n``."""
        rec = self.recursion
        n = rec.mu_n.shape[3] if nsites is None else nsites
        lld = (rec.mu_n.shape[2] - 2) // 2
        mu = np.asfortranarray(rec.mu_n[:, :, :, :n])
        g0 = self._g0_buffer(n)
        rec._check(rec._L.rsrec_chebyshev_green(rec._h, n, lld, len(self.ene), _ptr(self.ene), float(rec.en.energy_min), float(rec.en.energy_max),
                                                _ptr(mu), _ptr(g0)))
        return g0

    def ldos(self):
        """Orbital-resolved local density of states, -Im g0_jj / pi (density_of_states.f90:248-260)."""
        d = np.arange(18)
        return -self.g0[d, d].imag / np.pi
