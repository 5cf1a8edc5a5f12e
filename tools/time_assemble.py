#!/usr/bin/env python3
"""build_bulkham / build_locham on the device (rsrec_assemble_blocks) against the host route, per SCF iteration:
   host:   [reference builds ee / eeo / hall / hallo on the host]  ->  rsrec_set_hamiltonian uploads them
   device: rsrec_assemble_blocks (upload hmag, kernel, download the blocks for the host's other readers)  ->  rsrec_set_hamiltonian finds them there
The host's own assembly time (numpy restatement here, 2 small zgemm per block in the reference) is printed for scale only."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import load_golden
from oracle import oracle
from rslmtoasa_amd.recursion import Control, Hamiltonian, Lattice, Recursion


def med(f, n=20):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts[2:]))


if __name__ == "__main__":
    for name in ("bccFe_nsp2_block_hoh", "fccCu001_block_hoh", "B2FeCo_block_hoh"):
        z, hm = load_golden(name), load_golden(name + "_hmag")
        nmax = int(z["nmax"])
        ham = Hamiltonian(ee=z["ee"], lsham=z["lsham"], eeo=z["eeo"], enim=z["enim"], hall=z["hall"] if nmax else None, hallo=z["hallo"] if nmax else None, hoh=True)
        lat = Lattice(nn=z["nn"], iz=z["iz"], irec=np.asarray(z["irec"], np.int32), nmax=nmax, ntype=int(z["ntype"]))
        rec = Recursion(ham, lat, Control(lld=12, nsp=2), device=0)
        t_up = med(rec.update_hamiltonian)

        def assemble():
            rec.build_bulkham(hm["hmag_type"], hm["nbr_type_type"], hm["obarm"])
            if nmax:
                rec.build_locham(hm["hmag_atom"], hm["nbr_type_atom"], hm["obarm"])
        t_asm = med(assemble)
        t_set = med(rec.update_hamiltonian)
        assert rec.timing()["operator_arrays_from_device"] == (4 if nmax else 2)

        def host():
            oracle.assemble_blocks(hm["hmag_type"], hm["nbr_type_type"], hm["obarm"])
            if nmax:
                oracle.assemble_blocks(hm["hmag_atom"], hm["nbr_type_atom"], hm["obarm"])
        t_host = med(host, 8)
        nblk = (int(z["ntype"]) + nmax) * z["ee"].shape[2]
        print("%-22s %3d blocks: set_hamiltonian with upload %.3f ms | device assembly %.3f ms + set_hamiltonian from device copies %.3f ms | (numpy host assembly %.2f ms)"
              % (name, nblk, t_up, t_asm, t_set, t_host))
        rec.close()
