"""Multi-GPU glue of the recursion path: site partition + ONE packed all-reduce.

The recursion has no exchange inside its loop: every rank owns whole sites (recursion.f90:1816-1820, partition rule
of get_mpi_variables, mpi.f90:32-58).  The only collective on the path is the gather of per-site results, which the
reference performs as MPI_ALLREDUCE(MPI_IN_PLACE, ..., MPI_SUM) on zero-padded arrays (bands.f90:271-274 issues three
back-to-back calls; here the arrays are packed into one buffer -> one RCCL all-reduce over xGMI, latency-bound).
"""
import numpy as np

from .recursion import site_partition


def pack_local(arrays, start, end, nsites):
    """Zero-padded global image of per-site arrays: each `a` has the site index LAST and holds sites start..end (1-based)."""
    chunks = []
    for a in arrays:
        a = np.asarray(a)
        g = np.zeros(a.shape[:-1] + (nsites,), dtype=a.dtype)
        g[..., start - 1:end] = a[..., :end - start + 1]
        chunks.append(np.ascontiguousarray(g).view(np.float64).ravel())
    return np.concatenate(chunks)


def unpack_global(buf, arrays, nsites):
    out, off = [], 0
    for a in arrays:
        a = np.asarray(a)
        shape = a.shape[:-1] + (nsites,)
        n = int(np.prod(shape)) * (2 if np.iscomplexobj(a) else 1)
        out.append(buf[off:off + n].view(a.dtype).reshape(shape))
        off += n
    return out


def allgather_sites(arrays, rank, nprocs, nsites, dist=None, device=None):
    """All ranks obtain the per-site arrays of all `nsites` sites.  `dist` = torch.distributed (backend nccl = RCCL on the
    GPUs, gloo in the CPU tests); with dist=None (single process) the arrays are returned as they are."""
    start, end = site_partition(rank, nprocs, nsites)
    if dist is None or nprocs == 1:
        return [np.asarray(a)[..., :nsites] for a in arrays]
    import torch
    buf = torch.from_numpy(pack_local(arrays, start, end, nsites))
    if device is not None:
        buf = buf.to(device)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)          # sum of zero-padded images == all-gather (bands.f90:271)
    return unpack_global(buf.cpu().numpy(), arrays, nsites)
