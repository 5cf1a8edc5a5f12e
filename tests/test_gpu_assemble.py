"""build_bulkham / build_locham after chbar_nc (hamiltonian.f90:1553-1667) on the device: rsrec_assemble_blocks (second half).

Inputs: the four 9x9 parts (Hx, Hy, Hz, H0) the compiled reference held in `hmag` for every class atom, the type behind every neighbour
slot and `obarm` (tests/golden/<case>_hmag.npz, oracle/make_fixtures.py run_hmag_case).  Expected: the ee / eeo / hall / hallo of the same
reference run (<case>.npz).  ee is a sum / difference of two numbers (bit-exact); eeo = ee.obar is an 18-term complex dot product whose
summation order differs from the reference's zgemm: 1e-14 of the largest element."""
import numpy as np
import pytest

from helpers import RTOL, load_golden, rel_err
from rslmtoasa_amd.recursion import Control, Hamiltonian, Lattice, Recursion

pytestmark = pytest.mark.gpu
CASES = ["bccFe_nsp2_block_hoh", "bccFe_nsp4_block", "fccCu001_block_hoh", "B2FeCo_block_hoh"]


def make_rec(z, ham):
    lat = Lattice(nn=z["nn"], iz=z["iz"], irec=np.asarray(z["irec"], np.int32), nmax=int(z["nmax"]), ntype=int(z["ntype"]))
    return Recursion(ham, lat, Control(lld=int(z["lld"]), nsp=int(z["nsp"])), device=0)


@pytest.mark.parametrize("name", CASES)
def test_device_assembled_blocks_match_reference_and_are_reused(name):
    z, hm = load_golden(name), load_golden(name + "_hmag")
    hoh, nmax = bool(z["hoh"]), int(z["nmax"])
    keys = ["ee"] + (["eeo"] if hoh else []) + ((["hall"] + (["hallo"] if hoh else [])) if nmax else [])
    # reference run: blocks from the reference's host arrays
    ham_ref = Hamiltonian(ee=z["ee"], lsham=z["lsham"], eeo=z.get("eeo"), enim=z.get("enim"), hall=z.get("hall") if nmax else None,
                          hallo=z.get("hallo") if nmax and hoh else None, hoh=hoh)
    rec = make_rec(z, ham_ref)
    assert rec.timing()["operator_arrays_from_device"] == 0
    rec.recur_b()
    a_ref, b_ref = rec.a_b.copy(), rec.b2_b.copy()
    # the same engine, blocks assembled on the device
    ob = hm["obarm"] if hoh else None
    rec.build_bulkham(hm["hmag_type"], hm["nbr_type_type"] if hoh else None, ob)
    if nmax:
        rec.build_locham(hm["hmag_atom"], hm["nbr_type_atom"] if hoh else None, ob)
    H = rec.hamiltonian
    assert np.array_equal(H.ee, z["ee"])
    if nmax:
        assert np.array_equal(H.hall, z["hall"])
    if hoh:
        assert np.abs(H.eeo - z["eeo"]).max() <= 1e-14 * np.abs(z["eeo"]).max()
        if nmax:
            assert np.abs(H.hallo - z["hallo"]).max() <= 1e-14 * np.abs(z["hallo"]).max()
    rec.update_hamiltonian()
    assert rec.timing()["operator_arrays_from_device"] == len(keys)          # nothing was uploaded again
    rec.a_b[:] = 0
    rec.b2_b[:] = 0
    rec.recur_b()
    assert rel_err(rec.a_b, a_ref) <= RTOL and rel_err(rec.b2_b, b_ref) <= RTOL
    assert rel_err(rec.a_b, z["a_b"]) <= RTOL and rel_err(rec.b2_b, z["b2_b"]) <= RTOL
    # an array edited on the host after the assembly is NOT taken from the device
    H.ee = H.ee.copy(order="F")
    H.ee[0, 0, 0, 0] += 1e-3
    rec.update_hamiltonian()
    assert rec.timing()["operator_arrays_from_device"] == len(keys) - 1
    rec.recur_b()
    assert rel_err(rec.a_b, a_ref) > 1e-6
    rec.close()


def test_assemble_blocks_refuses_bad_arguments():
    z, hm = load_golden("fccCu001_block_hoh"), load_golden("fccCu001_block_hoh_hmag")
    ham = Hamiltonian(ee=z["ee"], lsham=z["lsham"], eeo=z["eeo"], enim=z["enim"], hoh=True)
    rec = make_rec(z, ham)
    from rslmtoasa_amd._lib import RsrecError
    bad = hm["nbr_type_type"].copy()
    bad[3, 1] = 7                                      # a type the obarm table does not have
    with pytest.raises(RsrecError):
        rec.build_bulkham(hm["hmag_type"], bad, hm["obarm"])
    rec.update_hamiltonian()                           # the engine is still usable, from the host arrays
    assert rec.timing()["operator_arrays_from_device"] == 0
    rec.recur_b()
    assert rel_err(rec.a_b, z["a_b"]) <= RTOL
    rec.close()
