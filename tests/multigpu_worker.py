"""One rank of the two-process hardware test of the path's one exchange (tests/test_gpu_multigpu.py): this process owns GPU `rank`,
runs recur_b on its share of the sites (get_mpi_variables rule, mpi.f90:32-58), writes its part of the zero-padded a / b2 image on the
device (rsrec_pack_diag) and sums the images of all ranks with the LIBRARY's communicator (rsrec_comm_init_file -> RCCL over xGMI;
no MPI, no torch.distributed) -- bands.f90:271-274.  Writes the reduced image to <out>/img_<rank>.npy."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, nranks, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    import torch                                          # torch first (two HIP runtimes over one ROCr)
    torch.cuda.set_device(rank)
    from helpers import objects_from, supercell_problem
    from rslmtoasa_amd.recursion import Recursion, site_partition
    p = supercell_problem((4, 4, 8))
    sites = np.array([1, 9, 40, 77, 100], dtype=np.int32)
    lld = 8
    rec = Recursion(*objects_from(p, sites, lld), device=rank, rank=rank, nprocs=nranks)
    rec.comm_init(rank, nranks, path=os.path.join(out, "comm.id"), timeout_s=120.0)
    assert rec.comm_size() == (rank, nranks)
    rec.recur_b()
    start, end = site_partition(rank, nranks, len(sites))
    img = torch.zeros((2, len(sites), 18, lld), dtype=torch.float64, device="cuda:%d" % rank)
    rec.pack_diag(start - 1, len(sites), img[0].data_ptr(), img[1].data_ptr())
    torch.cuda.synchronize()
    rec.allreduce_sum(img.data_ptr(), img.numel())        # device image reduced where it lies
    host = np.arange(16.0) + rank                         # a host array through the staging path
    rec.allreduce_sum(host)
    assert np.array_equal(host, nranks * np.arange(16.0) + sum(range(nranks)))
    np.save(os.path.join(out, "img_%d.npy" % rank), img.cpu().numpy())
    rec.close()


if __name__ == "__main__":
    main()
