"""world_size-2 test of the N>1 path on CPU (gloo): site partition by the get_mpi_variables rule + the one packed
all-reduce that gathers per-site coefficients (rslmtoasa_amd/parallel.py).  The per-rank chains are computed with the
CPU oracle here (no GPU in this container); on the GPU box the same glue runs under RCCL in bench.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, sites, lld, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import supercell_problem
    from oracle import oracle
    from rslmtoasa_amd.parallel import allgather_sites
    from rslmtoasa_amd.recursion import site_partition
    p = supercell_problem((4, 4, 8))
    start, end = site_partition(rank, world, len(sites))
    o = oracle.Oracle(p)
    a_b, b2_b = o.block_lanczos(sites[start - 1:end], lld)
    ga, gb = allgather_sites([a_b, b2_b], rank, world, len(sites), dist=dist)
    if rank == 0:
        q.put((ga, gb))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_site_partition_and_gather():
    from helpers import supercell_problem
    from oracle import oracle
    sites = np.array([1, 9, 40, 77, 100], dtype=np.int32)   # 5 sites over 2 ranks: 3 + 2 (remainder to the lowest rank)
    lld = 6
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sites, lld, q)) for r in range(2)]
    for p in procs:
        p.start()
    ga, gb = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    a_ref, b_ref = oracle.Oracle(supercell_problem((4, 4, 8))).block_lanczos(sites, lld)
    assert np.array_equal(ga, a_ref) and np.array_equal(gb, b_ref)


def test_pack_unpack_roundtrip():
    from rslmtoasa_amd.parallel import pack_local, unpack_global
    rng = np.random.default_rng(0)
    a = rng.standard_normal((3, 4, 2)) + 1j * rng.standard_normal((3, 4, 2))
    b = rng.standard_normal((5, 2))
    buf = pack_local([a, b], 2, 3, 6)
    ga, gb = unpack_global(buf, [a, b], 6)
    assert np.array_equal(ga[..., 1:3], a) and np.all(ga[..., [0, 3, 4, 5]] == 0)
    assert np.array_equal(gb[..., 1:3], b)
