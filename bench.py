#!/usr/bin/env python3
"""bench.py -- headline benchmark of the recursion hot path (BASELINE.json).

Workload (configs[1] of BASELINE.json): bcc Fe periodic supercell 22^3 = 10 648 atoms, spin-polarised 18x18
complex blocks (physical Fe stencil dumped from the reference's tests/scf/cases/bulk/bccFe), block-Lanczos
recursion with LL = 50.  One "step" = one full `recur_b` pass (recursion.f90:1807) over a batch of S = 64
recursion sites per GPU: 49 recursion levels of H|psi>, A_n, B_n^2, 18x18 eigen-solve and vector update for
every site.  Lattice tables and Hamiltonian blocks are resident in HBM before the timed region; the timed
region contains everything `recur_b` does per call (region search, kernels, coefficients back to the host)
and, for N > 1, the one packed RCCL all-reduce that gathers the per-site diagonal coefficients exactly like
the reference's MPI_ALLREDUCE-as-allgather (bands.f90:271-274).

Sites are independent: with N GPUs every rank owns S sites (weak scaling), no collective inside the loop.

usage: python bench.py --gpus N --steps K --warmup W   (N>1: launched by torch.distributed.run, one rank per GPU)
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

# the host side of a rank is one thread driving one GPU: keep the BLAS / OpenMP pools of numpy and torch small (one rank per GPU,
# up to 8 ranks per node; idle pools of one thread per visible CPU only add scheduler load)
for _k in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ.setdefault(_k, "8")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from rslmtoasa_amd.lattice import bcc_supercell, spread_sites, supercell_positions  # noqa: E402
from rslmtoasa_amd.recursion import Control, Energy, Hamiltonian, Lattice, Recursion  # noqa: E402

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (public spec); rate measured here: profiles/ubench_f64_r01.txt
HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOP_PER_BLOCK_MULT = 46656.0   # 18x18x18 complex MACs x 8 flop
# Memory-side bytes per H|psi> launch of the DEFAULT workload (22^3 atoms, 64 sites, LL=50), from separate rocprofv3 --pmc passes
# (profiles/r01_final_rocprof_summary.txt): 2 x FETCH_SIZE (gfx950 reads 1/2, MI355X_MICROARCH.md) + WRITE_SIZE = 29.2 GB + 3.0 GB.
# The counters sit on the L2's fabric side, so Infinity-Cache hits of the neighbour gathers are included.
HOP_TRAFFIC_BYTES_PER_LAUNCH = 32.2e9
CPU_SAMPLE_SITES = 10
BYTES_PER_ATOM_STEP = 51840.0   # 10 blocks of 5184 B per active atom per level, H_B = 0 (stencil operator)


def load_stencil():
    with np.load(os.path.join(ROOT, "tests", "golden", "bccFe_nsp2_block.npz"), allow_pickle=False) as z:
        return z["ee"], z["lsham"], z["slot_vec"]


def cpu_baseline(nn, ee, lsham, lld, threads):
    """CPU leg on the host cores of this node, bounded sample = ONE site of the same workload (same lattice, LL).

    Preferred: the compiled reference itself (oracle/_ref/ref_kernel.x, built in the build container from the
    reference sources; it is the reference's own recur_b/crecal_b/hop_b with MKL + OpenMP).  Fallback: the C
    restatement in oracle/ ("port")."""
    flop = None
    from rslmtoasa_amd.lattice import active_region_sizes
    sizes = [1] + active_region_sizes(nn, 1, lld - 1)
    # exact algorithmic work of one chain (matches SURVEY.md 8d: 384.7 GFLOP for this config)
    nb = int(nn[0, 0])
    mults = sum(nb * s for s in sizes[:-1])
    atom_steps = sum(sizes[1:])
    flop = FLOP_PER_BLOCK_MULT * (mults + 5 * atom_steps)
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_kernel.x")
    if os.path.exists(exe):
        try:
            sys.path.insert(0, ROOT)
            from oracle import fixture_io as fio
            scratch = tempfile.mkdtemp(prefix="rsrec_cpu_")
            kk = nn.shape[0]
            nsample = CPU_SAMPLE_SITES          # bounded sample: ~10-30 s of CPU work on 16 cores
            p = dict(nn=nn, iz=np.ones(kk, np.int32), irec=spread_sites(kk, nsample), lld=lld, nsp=2, hoh=0, kind=0, ee=ee, lsham=lsham)
            fio.write_kernel_in(os.path.join(scratch, "kernel_in.bin"), p)
            env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_STACKSIZE="1G")
            r = subprocess.run(["bash", "-c", "ulimit -s unlimited; exec " + exe], cwd=scratch, env=env, capture_output=True, text=True, timeout=600)
            t = None
            for line in r.stdout.splitlines():
                if "recursion wall time" in line:
                    t = float(line.split()[-2])
            if r.returncode == 0 and t:
                return {"value": nsample * flop / t * 1e-9, "unit": "GFLOP/s", "cores": threads, "kind": "reference", "seconds": t,
                        "sites_per_s": nsample / t,
                        "sample": "%d sites of the same %d-atom cell, LL=%d (%.1f GFLOP), compiled reference recur_b via oracle/_ref/ref_kernel.x (amdflang -O2 + MKL, OpenMP)" % (nsample, kk, lld, nsample * flop * 1e-9)}
        except Exception as e:  # noqa
            print("cpu_baseline(reference) failed: %r" % (e,), file=sys.stderr)
    from oracle import oracle
    os.environ["OMP_NUM_THREADS"] = str(threads)
    o = oracle.Oracle(dict(nn=nn, iz=np.ones(nn.shape[0], np.int32), ee=ee, lsham=lsham, hoh=0, nsp=2))
    t0 = time.time()
    o.block_lanczos(np.array([1], np.int32), lld)
    t = time.time() - t0
    return {"value": flop / t * 1e-9, "unit": "GFLOP/s", "cores": oracle.lib().orc_num_threads(), "kind": "port", "seconds": t,
            "sample": "1 site of the same 10648-atom cell, LL=%d (%.1f GFLOP), C restatement oracle/rsrec_oracle.c (OpenMP)" % (lld, flop * 1e-9)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sites", type=int, default=64, help="recursion sites per GPU per step")
    ap.add_argument("--cells", type=int, default=22, help="n for the n^3 periodic bcc supercell")
    ap.add_argument("--lld", type=int, default=50)
    ap.add_argument("--kernels", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--spmm4", type=int, default=-1, help="SpMM kernel: 0 = 16x16x4 MFMA, 1 = 4x4x4 MFMA one wave per group, 4 = 4x4x4 cooperative; -1 = library default")
    ap.add_argument("--no-positions", action="store_true", help="do not pass atom positions (locality hint)")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (development sweeps), may be repeated")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-green", action="store_true", help="skip the (untimed, separately reported) Green-function stage")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # one process per GPU.  BENCH_REHEARSAL=1 (development only): several ranks share the GPUs that exist and talk over gloo,
    # to exercise the N > 1 code path on a one-GPU box (RCCL refuses two ranks on one device)
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    device_index = local_rank % max(torch.cuda.device_count(), 1) if rehearsal else local_rank
    coll_device = "cpu" if rehearsal else "cuda"
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(device_index)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))   # "nccl" is RCCL on ROCm

    ee, lsham, slot_vec = load_stencil()
    n = args.cells
    nn = bcc_supercell((n, n, n), slot_vec)
    kk = nn.shape[0]
    nsites_total = args.sites * world
    irec = spread_sites(kk, nsites_total)
    lat = Lattice(nn=nn, iz=np.ones(kk, np.int32), irec=irec, nmax=0, ntype=1, cr=None if args.no_positions else supercell_positions((n, n, n)))
    ham = Hamiltonian(ee=ee, lsham=lsham, hoh=False)
    ctl = Control(lld=args.lld, nsp=2, recur="block")
    rec = Recursion(ham, lat, ctl, Energy(), device=device_index, rank=rank, nprocs=world)   # uploads tables: resident before timing
    if args.kernels:
        rec.set_option("kernels", args.kernels)
    if args.batch:
        rec.set_option("batch", args.batch)
    if args.spmm4 >= 0:
        rec.set_option("spmm4", args.spmm4)
    for kv in args.opt:
        k, v = kv.split("=")
        rec.set_option(k, int(v))

    from rslmtoasa_amd.parallel import allgather_sites

    def step():
        rec.recur_b()
        if world > 1:
            # the path's one exchange: zero-padded all-reduce == all-gather of the per-site results (bands.f90:271-274),
            # here the diagonal coefficients a(ll,l,site), b2(ll,l,site) that feed the LDOS continued fraction
            start, end = rec._my_sites()[:2]
            nloc = end - start + 1
            allgather_sites([rec.a[:args.lld, :, :nloc, 0], rec.b2[:args.lld, :, :nloc, 0]], rank, world, nsites_total, dist=dist, device=coll_device)

    for _ in range(args.warmup):
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    tm_acc = {"hop_ms": 0.0, "hop_launches": 0.0, "atom_steps": 0.0, "block_multiplies": 0.0, "total_ms": 0.0, "host_ms": 0.0}
    for _ in range(args.steps):
        step()
        tm = rec.timing()
        for k in tm_acc:
            tm_acc[k] += tm[k]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    if rank == 0:
        default_workload = (args.cells == 22 and args.sites == 64 and args.lld == 50 and not args.kernels and not args.batch
                            and args.spmm4 < 0 and not args.no_positions and not args.opt)
        # algorithmic work (reference semantics: only blocks whose source atom is inside the active region are multiplied)
        flop_rank = FLOP_PER_BLOCK_MULT * (tm_acc["block_multiplies"] + 5.0 * tm_acc["atom_steps"])
        flop_total = flop_rank * world
        bytes_total = BYTES_PER_ATOM_STEP * tm_acc["atom_steps"] * world
        # dominant kernel = H|psi>: the block SpMM, 46656 flop per block multiply (+ 46656 per atom-step when the kernel
        # also forms the A_n partial: VALU kernels and the fused MFMA variant)
        hop_flop = FLOP_PER_BLOCK_MULT * (tm_acc["block_multiplies"] + (tm_acc["atom_steps"] if tm.get("hop_fuses_a", 1.0) else 0.0))
        hop_s = tm_acc["hop_ms"] * 1e-3
        achieved = hop_flop / hop_s * 1e-12 if hop_s > 0 else 0.0
        out = {
            "metric": "block-recursion throughput (H|psi> + A_n + B_n recursion levels, recursion.f90 recur_b)",
            "value": flop_total / elapsed * 1e-9,
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic periodic bcc lattice; physical Fe spd stencil (18x18 complex blocks) dumped from the reference's bulk/bccFe case",
            "config": {"workload": "bcc Fe %d^3 = %d atoms, nsp=2 18x18 blocks, block Lanczos LL=%d, %d sites per GPU per step" % (n, kk, args.lld, args.sites),
                       "sites_per_gpu": args.sites, "atoms": kk, "lld": args.lld, "parallelism": "site-partition x%d (get_mpi_variables rule), no collective in the loop" % world},
            "sites_per_s": nsites_total * args.steps / elapsed,
            "atom_steps_per_s": tm_acc["atom_steps"] * world / elapsed,
            "gbytes_per_s": bytes_total / elapsed * 1e-9,
            "device_ms_per_step": tm_acc["total_ms"] / args.steps,
            "host_ms_per_step": tm_acc["host_ms"] / args.steps,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                         "traffic": HOP_TRAFFIC_BYTES_PER_LAUNCH if default_workload else None, "kernel": ("k_spmm5" if default_workload else "k_spmm5 / k_spmm4") + " (H|psi> block SpMM, FP64 MFMA 4x4x4)" if not tm.get("hop_fuses_a", 1.0) else "H|psi> (hop)", "launches": tm_acc["hop_launches"],
                         "avg_launch_ms": tm_acc["hop_ms"] / max(tm_acc["hop_launches"], 1),
                         "hbm_view": {"achieved": bytes_total / elapsed * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_total / elapsed * 1e-9 / HBM_PEAK_GBS,
                                      "note": "whole recursion level, algorithmic 51840 B per atom-step"}},
        }
        if world == 1 and not args.no_green:
            # Not part of `value`: the stage behind the recursion (zsqr + green%bgreen) for the same sites, once, so
            # that "sites/s from Hamiltonian to g0" can be quoted.  Terminator (get_terminf, a CPU routine in the reference too)
            # taken from the bulk bcc Fe fixture of the same Hamiltonian; 2510 energies = the reference's default mesh.
            try:
                from rslmtoasa_amd.green import Green
                gz = np.load(os.path.join(ROOT, "tests", "golden", "bccFe_nsp2_block_green.npz"), allow_pickle=False)
                nloc = args.sites
                ene = float(gz["ene_full_first"]) + float(gz["ene_full_step"]) * np.arange(int(gz["nen_full"]))
                a_inf = np.repeat(gz["a_inf"][:, :, :1], nloc, axis=2); b_inf = np.repeat(gz["b_inf"][:, :, :1], nloc, axis=2)
                gr = Green(rec, ene)
                for _ in range(2):                                   # second call = steady state of an SCF loop (buffers exist)
                    t0 = time.perf_counter()
                    rec.zsqr()
                    gr.block_green(a_inf, b_inf, nsites=nloc)
                    tg = time.perf_counter() - t0
                tmg = rec.timing()
                out["green"] = {"wall_ms": tg * 1e3, "kernel_ms": tmg["hop_ms"], "energies": len(ene),
                                "sites_per_s_recursion_plus_green": nloc / (elapsed / args.steps + tg),
                                "note": "zsqr + rsrec_block_green (green.f90:1191 bgreen) for the sites of one step, incl. the g0 download; not in `value`"}
            except Exception as e:  # noqa
                print("green stage skipped: %r" % (e,), file=sys.stderr)
        if world == 1 and not args.no_cpu:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            out["cpu_baseline"] = cpu_baseline(nn, ee, lsham, args.lld, threads)
        print(json.dumps(out))
    rec.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
