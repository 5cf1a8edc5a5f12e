"""TEST INFRASTRUCTURE ONLY (oracle side).

Readers/writers for the stream files exchanged with the compiled-reference drivers
(oracle/dump_fixture.f90, oracle/ref_kernel.f90) and for the committed golden fixtures
(tests/golden/*.npz).  Nothing under rslmtoasa_amd/ imports this module.

Array conventions follow the reference (Fortran, column-major):
  nn(kk, nncols) int32, 1-based, 0 = absent, column 1 = neighbour count incl. on-site   lattice.f90:1854
  iz(kk) int32 type map, irec(nrec) seed atoms                                          lattice.f90:138-239
  ee/eeo(18,18,nslots,ntype), hall/hallo(18,18,nslots,nmax), lsham/enim(18,18,ntype)    hamiltonian.f90:290-301
  a_b/b2_b(18,18,lld,nrec), mu_n(18,18,2*lld+2,nrec), a/b2(llmax,18,nrec)              recursion.f90:3760-3790
numpy arrays here are kept in *Fortran index order* (order='F'), i.e. arr[l, m, ll, site].
"""
import struct
import numpy as np

MAGIC = 0x52534658
GREEN_MAGIC = 0x47524e31
ETA_GREEN_MAGIC = 0x47524e33
CHEB_GREEN_MAGIC = 0x47524e32
DENSITY_MAGIC = 0x44454e31
KIND_BLOCK, KIND_CHEB, KIND_SCALAR, KIND_BLOCK_IJ, KIND_CHEB_IJ = 0, 1, 2, 3, 4


def _rd(f, dtype, shape):
    n = int(np.prod(shape)) if len(shape) else 1
    a = np.fromfile(f, dtype=dtype, count=n)
    if a.size != n:
        raise IOError("short read")
    return a.reshape(shape, order="F")


def read_fixture_bin(path):
    """Parse fixture.bin written by dump_fixture.f90 -> dict of numpy arrays."""
    d = {}
    with open(path, "rb") as f:
        magic, version = struct.unpack("<ii", f.read(8))
        assert magic == MAGIC and version == 1
        hdr = struct.unpack("<11i", f.read(44))
        kk, nncols, nmax, ntype, nrec, lld, nsp, hoh, kind, nslots, llmax = hdr
        emin, emax, acheb, bcheb = struct.unpack("<4d", f.read(32))
        d.update(kk=kk, nncols=nncols, nmax=nmax, ntype=ntype, nrec=nrec, lld=lld, nsp=nsp, hoh=hoh,
                 kind=kind, nslots=nslots, llmax=llmax, emin=emin, emax=emax, acheb=acheb, bcheb=bcheb)
        d["iz"] = _rd(f, np.int32, (kk,))
        d["nn"] = _rd(f, np.int32, (kk, nncols))
        d["irec"] = _rd(f, np.int32, (nrec,))
        d["cr"] = _rd(f, np.float64, (3, kk))
        d["ee"] = _rd(f, np.complex128, (18, 18, nslots, ntype))
        d["lsham"] = _rd(f, np.complex128, (18, 18, ntype))
        d["eeo"] = _rd(f, np.complex128, (18, 18, nslots, ntype))
        d["enim"] = _rd(f, np.complex128, (18, 18, ntype))
        if nmax > 0:
            d["hall"] = _rd(f, np.complex128, (18, 18, nslots, nmax))
            d["hallo"] = _rd(f, np.complex128, (18, 18, nslots, nmax))
        _read_outputs(f, d, kind, lld, nrec, llmax)
        tail = f.read(12)
        if tail:
            # Green-function stage appended by dump_fixture.f90 for block recursions (self.f90:820-831 run_dos):
            # energies, terminator, sqrt(B^2) (zsqr), g0 = block_green (green.f90:588-621, bgreen :1191-1339)
            gmagic, nen, sym = struct.unpack("<iii", tail)
            if gmagic == DENSITY_MAGIC:        # scalar recursion: green%sgreen / dos%density (green.f90:628-705, density_of_states.f90:248-404); sym = nmdir
                nmd = sym
                dn = dict(nen=nen, nmdir=nmd, ene=_rd(f, np.float64, (nen,)))
                pot = _rd(f, np.float64, (18, 2, nrec))                  # per site: dw_l(1:18), cshi(1:18)
                dn["dw_l"], dn["cshi"] = np.asfortranarray(pot[:, 0, :]), np.asfortranarray(pot[:, 1, :])
                dn["a"] = _rd(f, np.float64, (llmax, 18, nrec, nmd))
                dn["b2"] = _rd(f, np.float64, (llmax, 18, nrec, nmd))
                dn["tdens"] = _rd(f, np.float64, (18, nen, nrec, nmd))   # written (site fastest, then direction): same order
                dn["g0"] = _rd(f, np.complex128, (18, 18, nen, nrec))
                d["density"] = dn
                assert f.read(1) == b"", "trailing bytes in fixture"
                return d
            if gmagic == CHEB_GREEN_MAGIC:     # chebyshev_green (green.f90:1030-1108): energies, g0
                d["green"] = dict(nen=nen, ene=_rd(f, np.float64, (nen,)), g0=_rd(f, np.complex128, (18, 18, nen, nrec)))
                assert f.read(1) == b"", "trailing bytes in fixture"
                return d
            assert gmagic == GREEN_MAGIC
            d["green"] = dict(nen=nen, sym_term=sym, ene=_rd(f, np.float64, (nen,)), a_inf=_rd(f, np.float64, (18, 18, nrec)),
                              b_inf=_rd(f, np.float64, (18, 18, nrec)), b_sqrt=_rd(f, np.complex128, (18, 18, lld, nrec)),
                              g0=_rd(f, np.complex128, (18, 18, nen, nrec)))
            tail = f.read(12)
            if tail:
                # green%block_green_eta (green.f90:544-579): (energy point, eta) pairs and g at that point
                emagic, neta, _ = struct.unpack("<iii", tail)
                assert emagic == ETA_GREEN_MAGIC
                pts, etas, gs = [], [], []
                for _ in range(neta):
                    pts.append(struct.unpack("<i", f.read(4))[0])
                    etas.append(complex(*struct.unpack("<dd", f.read(16))))
                    gs.append(_rd(f, np.complex128, (18, 18, nrec)))
                d["green"].update(eta_points=np.array(pts, np.int32), eta=np.array(etas, np.complex128), g_eta=np.stack(gs, axis=2))   # (18,18,neta,nrec)
            assert f.read(1) == b"", "trailing bytes in fixture"
    return d


def _read_outputs(f, d, kind, lld, nrec, llmax):
    if kind in (KIND_BLOCK_IJ, KIND_CHEB_IJ):
        nrec = 4 * (nrec // 2)        # nrec counts the 2*npairs atoms of the pair list; outputs have 4 chains per pair
        kind -= 3
    if kind == KIND_BLOCK:
        d["a_b"] = _rd(f, np.complex128, (18, 18, lld, nrec))
        d["b2_b"] = _rd(f, np.complex128, (18, 18, lld, nrec))
    elif kind == KIND_CHEB:
        d["mu_n"] = _rd(f, np.complex128, (18, 18, 2 * lld + 2, nrec))
    else:
        d["a"] = _rd(f, np.float64, (llmax, 18, nrec))
        d["b2"] = _rd(f, np.float64, (llmax, 18, nrec))


def write_kernel_in(path, p):
    """Write kernel_in.bin for ref_kernel.f90 from a problem dict (same keys as a fixture)."""
    kk, nncols = p["nn"].shape
    nslots, ntype = p["ee"].shape[2], p["ee"].shape[3]
    nmax = int(p.get("nmax", 0))
    nrec = len(p["irec"])
    z = lambda shape: np.zeros(shape, dtype=np.complex128, order="F")
    with open(path, "wb") as f:
        f.write(struct.pack("<ii", MAGIC, 1))
        f.write(struct.pack("<11i", kk, nncols, nmax, ntype, nrec, int(p["lld"]), int(p["nsp"]), int(p["hoh"]),
                            int(p["kind"]), nslots, int(p["lld"])))
        f.write(struct.pack("<2d", float(p.get("emin", -1.0)), float(p.get("emax", 1.0))))
        for name, dt in (("iz", np.int32), ("nn", np.int32), ("irec", np.int32)):
            np.asarray(p[name], dtype=dt).ravel(order="F").tofile(f)
        for name, shape in (("ee", (18, 18, nslots, ntype)), ("lsham", (18, 18, ntype)),
                            ("eeo", (18, 18, nslots, ntype)), ("enim", (18, 18, ntype))):
            np.asarray(p.get(name, z(shape)), dtype=np.complex128).ravel(order="F").tofile(f)
        if nmax > 0:
            for name in ("hall", "hallo"):
                np.asarray(p.get(name, z((18, 18, nslots, nmax))), dtype=np.complex128).ravel(order="F").tofile(f)


def read_kernel_out(path, lld, nrec):
    d = {}
    with open(path, "rb") as f:
        magic, version, kind = struct.unpack("<iii", f.read(12))
        assert magic == MAGIC
        _read_outputs(f, d, kind, lld, nrec, lld)
    return d


INPUT_KEYS = ("iz", "nn", "irec", "ee", "lsham", "eeo", "enim", "hall", "hallo")
SCALAR_KEYS = ("kk", "nmax", "ntype", "nrec", "lld", "nsp", "hoh", "kind", "nslots", "emin", "emax", "acheb", "bcheb")
OUTPUT_KEYS = ("a_b", "b2_b", "mu_n", "a", "b2")


def save_golden(path, d, extra=None, drop=("cr",)):
    out = {k: np.asarray(v) for k, v in d.items() if k not in drop and k != "nncols" and k != "llmax" and k != "green" and k != "density"}
    if not d.get("hoh"):
        # eeo/enim are not read by the non-hoh path: do not store megabytes of unused blocks
        out.pop("eeo", None); out.pop("enim", None); out.pop("hallo", None)
    if extra:
        out.update(extra)
    np.savez_compressed(path, **out)


def load_golden(path):
    with np.load(path, allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    for k in SCALAR_KEYS:
        if k in d:
            d[k] = d[k].item()
    return d
