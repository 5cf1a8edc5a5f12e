#!/usr/bin/env python3
"""bench.py -- headline benchmark of the recursion hot path (BASELINE.json).

Default workload (since round 4): the cell BASELINE.json's north-star target is quoted on -- bcc Fe periodic supercell 46^3 = 97 336
atoms (configs[2], the 10^5-atom cell; one GPU holds it: 4 vectors x 64 chains x 505 MB), spin-polarised 18x18
complex blocks (physical Fe stencil dumped from the reference's tests/scf/cases/bulk/bccFe), block-Lanczos
recursion with LL = 50.  `--cells 22` is configs[1] (10 648 atoms; the default of rounds 1-3).  Parity at 46^3: the reference cannot
build that cell here, so its levels 1-10 are compared with the reference's on the 22^3 cell (identical until the regions meet their
periodic images) and the levels beyond are property-checked (tests/test_gpu_parity.py), not parity-checked.
One "step" = one full `recur_b` pass (recursion.f90:1807) over a batch of S = 64
recursion sites per GPU: 49 recursion levels of H|psi>, A_n, B_n^2, 18x18 eigen-solve and vector update for
every site.  Lattice tables and Hamiltonian blocks are resident in HBM before the timed region; the timed
region contains everything `recur_b` does per call (region search, kernels, coefficients back to the host)
and, for N > 1, the one RCCL all-reduce that gathers the per-site diagonal coefficients exactly like the
reference's MPI_ALLREDUCE-as-allgather (bands.f90:271-274) -- packed and reduced on the device.

Other workloads of BASELINE.json (same metric, named in config.workload):
  --cells 22              configs[1]: 22^3 = 10 648 atoms, 64 sites per GPU
  --hoh                   H = h - h o h + e_nu + l.s (hop_b_hoh, recursion.f90:1411)
  --recur chebyshev       configs[3] shape: Chebyshev moments (chebyshev_recur, recursion.f90:3057)
  --spin-mixing           the same stencil in a spin frame tilted by 60 degrees: every hopping block has spin-flip entries
                          (non-collinear operator; nothing is skipped as a structural zero)
  --workload fccCu001     configs[3]: the reference's surface case (tests/scf/cases/surface/fccCu001: fcc Cu(001) slab cluster, 9 318 atoms,
                          THREE atom types, 19 neighbour slots), Chebyshev recursion LL=50, 64 sites; lattice tables and blocks from the
                          committed fixture tests/golden/fccCu001_cheb.npz (the reference holds no Fe(001) case)
  --workload kubo         stochastic Kubo-Bastin double moments (compute_moments_stochastic, recursion.f90:979): periodic fcc Pt
                          supercell (--cells, default 20 -> 8 000 atoms) with the blocks and velocity operators of the reference's
                          conductivity case (tests/golden/fccPt_kubo[_hoh].npz), --cond-ll moment orders (default 50); a step = the
                          cond_ll x cond_ll moment blocks of one vector: 3 cond_ll whole-lattice SpMMs + the moment GEMMs; single GPU
  --workload B2FeCo       configs[4]: the reference's impurity case (tests/scf/cases/impurity/B2FeCo: 4 152 atoms, nmax = 15 atoms with
                          per-atom `hall` blocks + 3 bulk types), block Lanczos with hoh, LL=50; sites = the 15 impurity-region atoms
                          (per-site LDOS of all inequivalent atoms) + 49 host atoms; fixture tests/golden/B2FeCo_block_hoh.npz

Sites are independent: with N GPUs every rank owns S sites (weak scaling), no collective inside the loop.

usage: python bench.py --gpus N --steps K --warmup W
  N > 1 without WORLD_SIZE in the environment: this process only launches `python -m torch.distributed.run --nproc-per-node N`
  on itself (it never touches the GPU) and relays the ranks' output; under torch.distributed.run WORLD_SIZE must equal N.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (public spec); rate measured here: profiles/ubench_f64_r01.txt
HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FLOP_PER_BLOCK_MULT = 46656.0   # 18x18x18 complex MACs x 8 flop
CPU_SAMPLE_SITES = 10
BYTES_PER_ATOM_STEP = {"block": 51840.0, "chebyshev": 15552.0}   # 10 resp. 3 blocks of 5184 B per active atom per level, H_B = 0
POST_FLOP_BLOCKS = {"block": 5.0, "chebyshev": 2.0}              # (nb + 5) resp. (nb + 2) x 46 656 flop per atom-step


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--sites", type=int, default=64, help="recursion sites per GPU per step")
    ap.add_argument("--cells", type=int, default=None, help="n for the n^3 periodic supercell (default: 46 = the 10^5-atom north-star cell; kubo: 20)")
    ap.add_argument("--lld", type=int, default=50)
    ap.add_argument("--recur", choices=("block", "chebyshev"), default=None, help="default: block (chebyshev for --workload fccCu001)")
    ap.add_argument("--cond-ll", type=int, default=50, help="--workload kubo: moment orders per side")
    ap.add_argument("--vectors", type=int, default=1, help="--workload kubo: vectors per step (the library advances up to 8 together as the chains of one launch)")
    ap.add_argument("--workload", choices=("bcc", "fccCu001", "B2FeCo", "kubo"), default="bcc", help="bcc: synthetic periodic bcc Fe supercell (--cells); others: lattices of the reference's own cases")
    ap.add_argument("--hoh", action="store_true")
    ap.add_argument("--spin-mixing", action="store_true", help="stencil rotated into a tilted spin frame: spin-flip entries in every block")
    ap.add_argument("--kernels", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--no-positions", action="store_true", help="do not pass atom positions (locality hint)")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (development sweeps), may be repeated")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-green", action="store_true", help="skip the (untimed, separately reported) LDOS stage")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--master-port", type=int, default=0)
    args = ap.parse_args()
    if args.recur is None:
        args.recur = "chebyshev" if args.workload == "fccCu001" else "block"
    if args.workload == "B2FeCo":
        args.hoh = True
    args.cells_given = args.cells is not None
    if args.cells is None:
        args.cells = 20 if args.workload == "kubo" else 46
    return args


def launch_ranks(args):
    """--gpus N > 1 outside torch.distributed.run: start N ranks as a child job.  This parent never imports torch or loads
    librsrec (a process that has touched the GPU must not spawn/exec the job, and needs no device itself)."""
    port = args.master_port or (29500 + os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env)
    sys.exit(r.returncode)


def load_stencil(hoh):
    import numpy as np
    name = "bccFe_nsp2_block_hoh.npz" if hoh else "bccFe_nsp2_block.npz"
    with np.load(os.path.join(ROOT, "tests", "golden", "bccFe_nsp2_block.npz"), allow_pickle=False) as z:
        slot_vec = z["slot_vec"]
    with np.load(os.path.join(ROOT, "tests", "golden", name), allow_pickle=False) as z:
        d = {k: z[k] for k in ("ee", "lsham") + (("eeo", "enim") if hoh else ())}
    d["slot_vec"] = slot_vec
    return d


def tilt_spin_frame(st, theta):
    """H' = U^H H U with U = exp(-i theta sigma_y / 2) (x) 1_9 applied to every block: the same physical operator written in a
    spin frame tilted by theta about y.  Hopping blocks of a collinear magnet, diagonal in spin in the global frame
    (hamiltonian.f90:1553-1617), acquire up/down entries -- what a non-collinear (nsp = 3/4, rotated moments) run feeds hop_b."""
    import numpy as np
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    U = np.kron(np.array([[c, -s], [s, c]]), np.eye(9)).astype(np.complex128)
    out = dict(st)
    for k in ("ee", "eeo"):
        if k in st:
            out[k] = np.einsum("ab,bcst,cd->adst", U.conj().T, st[k], U)
    for k in ("lsham", "enim"):
        if k in st:
            out[k] = np.einsum("ab,bct,cd->adt", U.conj().T, st[k], U)
    return out


def algorithmic_work(nn, seed, napply, hoh):
    """Block multiplies and post-hop atom-steps of ONE chain from `seed` (1-based) in the reference's count (hop_b :1576-1625: one zgemm
    per (atom, slot) whose source atom is inside the region; hop_b_hoh: two passes + the two on-site products): any lattice table."""
    import numpy as np
    kk = nn.shape[0]
    nb = int(nn[:, 0].max())
    nbr = nn[:, 1:nb]
    fan = 1 + np.bincount(nbr[nbr > 0].ravel(), minlength=kk + 1)[1:]          # on-site + every atom that lists n as a neighbour
    active = np.zeros(kk + 1, dtype=bool)
    active[seed] = True
    mults = atom_steps = 0.0
    for _ in range(napply):
        for p in range(2 if hoh else 1):
            mults += float(fan[active[1:]].sum()) + (2.0 * float(active[1:].sum()) if (hoh and p == 0) else 0.0)
            hit = active[nbr].any(axis=1)
            active[1:] |= hit
            active[0] = False
        atom_steps += float(active[1:].sum())
    return mults, atom_steps


def build_workload(args, world):
    """The recursion problem of a bench run: tables, blocks, the sites of all ranks, and the names the JSON line carries."""
    import numpy as np
    from rslmtoasa_amd.lattice import bcc_supercell, spread_sites, supercell_positions
    nsites_total = args.sites * world
    W = {"emin": -3.0, "emax": 1.8}       # Chebyshev window of the reference's Chebyshev cases (tests/golden/*_cheb.npz)
    if args.workload == "bcc":
        st = load_stencil(args.hoh)
        if args.spin_mixing:
            st = tilt_spin_frame(st, np.pi / 3)
        n = args.cells
        nn = bcc_supercell((n, n, n), st["slot_vec"])
        kk = nn.shape[0]
        variant = ("hoh " if args.hoh else "") + ("spin-mixing " if args.spin_mixing else "")
        W.update(nn=nn, iz=np.ones(kk, np.int32), nmax=0, ntype=1, ee=st["ee"], lsham=st["lsham"], eeo=st.get("eeo"), enim=st.get("enim"), hall=None, hallo=None,
                 cr=None if args.no_positions else supercell_positions((n, n, n)), irec=spread_sites(kk, nsites_total),
                 name="bcc Fe %d^3 = %d atoms, nsp=2 18x18 blocks, %s%s LL=%d, %d sites per GPU per step"
                      % (n, kk, variant, "block Lanczos" if args.recur == "block" else "Chebyshev", args.lld, args.sites),
                 key="%s%s%s_c%d_s%d_l%d" % (args.recur, "_hoh" if args.hoh else "", "_mix" if args.spin_mixing else "", n, args.sites, args.lld),
                 data="synthetic periodic bcc lattice; physical Fe spd stencil (18x18 complex blocks) dumped from the reference's bulk/bccFe case")
        if kk > 20000:
            W["parity_note"] = ("the reference cannot build a cell of this size in the build container: levels 1-10 are compared with the compiled reference's "
                                "on the 22^3 cell (identical until the regions meet their images), the levels beyond are property-checked "
                                "(translation invariance between sites, Hermiticity of A_n and B_n^2; tests/test_gpu_parity.py), not parity-checked")
        return W
    if args.spin_mixing or args.cells_given:
        print("bench.py: --cells / --spin-mixing belong to --workload bcc", file=sys.stderr)
        sys.exit(2)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import load_golden_with_inputs
    if args.workload == "fccCu001":
        g = load_golden_with_inputs("fccCu001_block_hoh" if args.hoh else "fccCu001_cheb")
        kk = int(g["kk"])
        irec = spread_sites(kk, nsites_total)
        what = "fcc Cu(001) surface cluster of the reference's surface/fccCu001 case: %d atoms, 3 atom types, %d neighbour slots" % (kk, int(g["nn"][:, 0].max()))
    else:
        g = load_golden_with_inputs("B2FeCo_block_hoh")
        kk, nmax = int(g["kk"]), int(g["nmax"])
        host = nmax + spread_sites(kk - nmax, max(nsites_total - nmax, 1))
        irec = np.concatenate([np.arange(1, nmax + 1), host])[:nsites_total].astype(np.int32)
        what = "B2 FeCo impurity cluster of the reference's impurity/B2FeCo case: %d atoms, %d of them with per-atom hall blocks, 3 bulk types" % (kk, nmax)
    W.update(nn=g["nn"], iz=g["iz"], nmax=int(g.get("nmax", 0)), ntype=g["ee"].shape[3], ee=g["ee"], lsham=g["lsham"], eeo=g.get("eeo") if args.hoh else None,
             enim=g.get("enim") if args.hoh else None, hall=g.get("hall"), hallo=g.get("hallo") if args.hoh else None,
             cr=None if args.no_positions else cluster_positions(args.workload, g), irec=irec,
             name="%s; nsp=2 18x18 blocks, %s%s LL=%d, %d sites per GPU per step" % (what, "hoh " if args.hoh else "", "block Lanczos" if args.recur == "block" else "Chebyshev", args.lld, args.sites),
             key="%s%s_%s_s%d_l%d" % (args.recur, "_hoh" if args.hoh else "", args.workload, args.sites, args.lld),
             data="lattice tables and Hamiltonian blocks of the reference's own %s case (committed fixture tests/golden/, dumped from the compiled reference)" % args.workload)
    return W


def cluster_positions(workload, g):
    """lattice%cr of the reference run behind the fixture (tests/golden/<workload>_cr.npz): the locality hint the Fortran drop-in passes too."""
    import numpy as np
    with np.load(os.path.join(ROOT, "tests", "golden", workload + "_cr.npz"), allow_pickle=False) as z:
        cr = z["cr"]
    assert cr.shape == (3, int(g["kk"]))
    return cr


FCC_PRIMITIVE = [[0.0, 0.5, 0.5], [0.5, 0.0, 0.5], [0.5, 0.5, 0.0]]      # lattice.f90 bravais, fcc, units of alat


def main_kubo(args):
    """--workload kubo: rsrec_kubo_moments on a periodic fcc Pt supercell (one GPU; the vectors of a real run shard over ranks like sites)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if args.gpus != 1 or os.environ.get("WORLD_SIZE", "1") != "1":
        print("bench.py: --workload kubo is a single-GPU line", file=sys.stderr)
        sys.exit(2)
    from helpers import load_golden
    from rslmtoasa_amd.lattice import bcc_supercell
    from rslmtoasa_amd.recursion import Control, Energy, Hamiltonian, Lattice, Recursion
    import rslmtoasa_amd.recursion as R
    import torch
    torch.cuda.set_device(0)
    z = load_golden("fccPt_kubo_hoh" if args.hoh else "fccPt_kubo")
    n, cond_ll, nvec = args.cells, args.cond_ll, max(1, args.vectors)
    nn = bcc_supercell((n, n, n), z["slot_vec"], primitive=np.array(FCC_PRIMITIVE))
    kk, nb = nn.shape[0], int(nn[0, 0])
    a, b = float(z["acheb"]), float(z["bcheb"])
    ham = Hamiltonian(ee=z["ee"], lsham=z["lsham"], eeo=z.get("eeo"), enim=z.get("enim"), hoh=args.hoh)
    lat = Lattice(nn=nn, iz=np.ones(kk, np.int32), irec=np.array([1], np.int32), nmax=0, ntype=1)
    rec = Recursion(ham, lat, Control(lld=cond_ll, nsp=2), Energy(), device=0)
    for kv in args.opt:
        k, v = kv.split("=")
        rec.set_option(k, int(v))
    R.chebyshev_scaling = lambda emin, emax: (a, b)          # the fixture's window (the mirror would derive it from energy_min / energy_max)
    vo = dict(vo_a=z.get("vo_a"), vo_b=z.get("vo_b")) if args.hoh else {}

    def step():
        return rec.compute_moments_stochastic(z["v_a"], z["v_b"], cond_ll, atlist=np.arange(1, nvec + 1, dtype=np.int32), **vo)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = {"hop_ms": 0.0, "rest_ms": 0.0, "total_ms": 0.0, "hop_launches": 0.0, "block_multiplies": 0.0, "hop_required_flop": 0.0, "hop_mfma_flop": 0.0}
    for _ in range(args.steps):
        mu = step()
        tm = rec.timing()
        for k in acc:
            acc[k] += tm[k]
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert np.isfinite(mu).all()
    gemm_flop = 8.0 * (cond_ll * 18.0) ** 2 * kk * 18.0 * args.steps * nvec          # L^H R over the (atom, row) index: complex MACs x 8
    spmm_alg = FLOP_PER_BLOCK_MULT * acc["block_multiplies"]
    hop_s, gemm_s = acc["hop_ms"] * 1e-3, acc["rest_ms"] * 1e-3
    achieved = acc["hop_required_flop"] / hop_s * 1e-12
    out = {
        "metric": "Kubo double-moment throughput (whole-lattice H / velocity SpMMs + moment contraction, recursion.f90 compute_moments_stochastic)",
        "value": (spmm_alg + gemm_flop) / elapsed * 1e-9, "unit": "GFLOP/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic periodic fcc lattice; Pt spd blocks and velocity operators (v_a, v_b%s) dumped from the reference's conductivity/fccPt case" % (", vo_a, vo_b" if args.hoh else ""),
        "config": {"workload": "fcc Pt %d^3 = %d atoms, nsp=2 18x18 blocks, %sstochastic Kubo moments cond_ll=%d (%d x %d blocks of 18x18), %d vector%s per step" % (n, kk, "hoh " if args.hoh else "", cond_ll, cond_ll, cond_ll, nvec, "" if nvec == 1 else "s"),
                   "workload_key": "kubo%s_c%d_l%d%s" % ("_hoh" if args.hoh else "", n, cond_ll, "" if nvec == 1 else "_v%d" % nvec), "atoms": kk, "cond_ll": cond_ll, "vectors_per_step": nvec, "neighbour_slots": nb,
                   "parallelism": "single GPU (the vectors of a run shard over ranks like recursion sites)", "collective": None},
        "vectors_per_s": args.steps * nvec / elapsed, "ms_per_vector": elapsed / (args.steps * nvec) * 1e3,
        "device_ms_per_step": acc["total_ms"] / args.steps,
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS, "frac_kernel": achieved / FP64_PEAK_TFLOPS,
                     "frac_step": (acc["hop_required_flop"] + gemm_flop) / elapsed * 1e-12 / FP64_PEAK_TFLOPS,
                     "frac_algorithmic": spmm_alg / hop_s * 1e-12 / FP64_PEAK_TFLOPS, "frac_step_algorithmic": (spmm_alg + gemm_flop) / elapsed * 1e-12 / FP64_PEAK_TFLOPS,
                     "traffic": None, "kernel": "k_spmm5 (whole-lattice block SpMM: H with the fused Chebyshev step, v_a, v_b)", "launches": acc["hop_launches"],
                     "avg_launch_ms": acc["hop_ms"] / max(acc["hop_launches"], 1), "share_of_device_time": acc["hop_ms"] / max(acc["total_ms"], 1e-9),
                     "flops_counted": "required by the block structure (frac, frac_kernel, frac_step); *_algorithmic = 46656 per block multiply",
                     "executed": {"achieved": acc["hop_mfma_flop"] / hop_s * 1e-12, "frac": acc["hop_mfma_flop"] / hop_s * 1e-12 / FP64_PEAK_TFLOPS, "unit": "TFLOP/s"},
                     "gemm": {"kernel": "k_kubo_gram + k_kubo_gram_reduce (own FP64 MFMA contraction L^H R on the vectors in place, %d x %d x %d per 64 right vectors)" % (cond_ll * 18, min(cond_ll, 64) * 18, kk * 18), "ms_per_step": acc["rest_ms"] / args.steps,
                              "achieved": gemm_flop / gemm_s * 1e-12 if gemm_s > 0 else None, "frac": gemm_flop / gemm_s * 1e-12 / FP64_PEAK_TFLOPS if gemm_s > 0 else None,
                              "share_of_device_time": acc["rest_ms"] / max(acc["total_ms"], 1e-9)}},
    }
    if not args.no_cpu:
        from rslmtoasa_amd._proc import under_profiler
        if under_profiler():
            print("cpu_baseline skipped: running under a profiler", file=sys.stderr)
        else:
            # bounded sample: the same lattice and operators at a small moment order on the C restatement (the compiled reference's
            # compute_moments_stochastic builds its own lattice from a case directory and cannot be fed this table)
            from oracle import oracle
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            os.environ["OMP_NUM_THREADS"] = str(threads)
            c_ll = min(cond_ll, 8)
            prob = dict(nn=nn, iz=np.ones(kk, np.int32), ee=z["ee"], lsham=z["lsham"], hoh=int(args.hoh), nsp=2, nmax=0)
            if args.hoh:
                prob.update(eeo=z["eeo"], enim=z["enim"])
            o = oracle.Oracle(prob)
            tc = time.time()
            o.kubo_moments(np.array([[1]], np.int32), np.ones((1, 1), np.complex128), c_ll, a, b, z["v_a"], z["v_b"], z.get("vo_a") if args.hoh else None, z.get("vo_b") if args.hoh else None)
            tc = time.time() - tc
            flop_c = FLOP_PER_BLOCK_MULT * acc["block_multiplies"] / args.steps / max(acc["hop_launches"] / args.steps, 1) * (3 * c_ll - 1) * (2 if args.hoh else 1) + 8.0 * (c_ll * 18.0) ** 2 * kk * 18.0
            out["cpu_baseline"] = {"value": flop_c / tc * 1e-9, "unit": "GFLOP/s", "cores": oracle.lib().orc_num_threads(), "cpu_quota": cpu_quota(), "host_cpus": os.cpu_count(), "kind": "port", "seconds": tc,
                                   "sample": "the same %d-atom lattice and operators at cond_ll=%d (%.1f GFLOP), C restatement oracle/rsrec_oracle.c (OpenMP)" % (kk, c_ll, flop_c * 1e-9)}
    print(json.dumps(out), flush=True)
    rec.close()


def cpu_quota():
    """CPU bandwidth limit of this container (cgroup v2 cpu.max / v1 cfs quota), in CPUs; None if unlimited/unknown."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(p)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:
        return None


def cpu_baseline(W, lld, threads, recur, hoh):
    """CPU leg on the host cores of this node, bounded sample = a few sites of the same workload (same lattice, LL).

    Preferred: the compiled reference itself (oracle/_ref/ref_kernel.x, built in the build container from the
    reference sources: the reference's own recur_b / chebyshev_recur with MKL + OpenMP; SURVEY 8c planned the C port
    for this leg, the compiled reference is the stronger baseline and the task statement allows it as kind "reference").
    Fallback: the C restatement in oracle/ ("port")."""
    import numpy as np
    from rslmtoasa_amd._proc import run_with_unlimited_stack, under_profiler
    nn, emin, emax = W["nn"], W["emin"], W["emax"]
    napply = lld - 1 if recur == "block" else lld + 1
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_kernel.x")
    if under_profiler():
        print("cpu_baseline skipped: running under a profiler (its preload must not reach a child process)", file=sys.stderr)
        return None
    kk = nn.shape[0]
    what = "recur_b" if recur == "block" else "chebyshev_recur"
    prob = dict(nn=nn, iz=W["iz"], nmax=W["nmax"], ee=W["ee"], lsham=W["lsham"], hoh=int(hoh), nsp=2)
    for k in ("eeo", "enim", "hall", "hallo"):
        if W.get(k) is not None:
            prob[k] = W[k]
    # bounded sample: sites of the run's own list, ~4 TFLOP = 10-30 s of CPU work on 16 cores
    work = [algorithmic_work(nn, int(s), napply, hoh) for s in W["irec"][:CPU_SAMPLE_SITES]]
    flops = [FLOP_PER_BLOCK_MULT * (m + POST_FLOP_BLOCKS[recur] * a) for m, a in work]
    nsample = 1
    while nsample < len(flops) and sum(flops[:nsample + 1]) <= 4.0e12:
        nsample += 1
    flop = sum(flops[:nsample])
    if os.path.exists(exe):
        try:
            from oracle import fixture_io as fio
            scratch = tempfile.mkdtemp(prefix="rsrec_cpu_")
            p = dict(prob, irec=np.asarray(W["irec"][:nsample], np.int32), lld=lld, kind=0 if recur == "block" else 1, emin=emin, emax=emax)
            fio.write_kernel_in(os.path.join(scratch, "kernel_in.bin"), p)
            r = run_with_unlimited_stack([exe], cwd=scratch, env={"OMP_NUM_THREADS": str(threads)}, timeout=900)   # scrubbed environment, no shell hop
            t = None
            for line in r.stdout.splitlines():
                if "recursion wall time" in line:
                    t = float(line.split()[-2])
            if r.returncode == 0 and t:
                return {"value": flop / t * 1e-9, "unit": "GFLOP/s", "cores": threads, "cpu_quota": cpu_quota(), "host_cpus": os.cpu_count(),
                        "kind": "reference", "seconds": t, "sites_per_s": nsample / t,
                        "sample": "%d sites of the same %d-atom lattice, LL=%d (%.1f GFLOP), compiled reference %s via oracle/_ref/ref_kernel.x (amdflang -O2 + MKL, OpenMP)" % (nsample, kk, lld, flop * 1e-9, what)}
            print("cpu_baseline(reference) failed rc=%d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:]), file=sys.stderr)
        except Exception as e:  # noqa
            print("cpu_baseline(reference) failed: %r" % (e,), file=sys.stderr)
    from oracle import oracle
    os.environ["OMP_NUM_THREADS"] = str(threads)
    o = oracle.Oracle(prob)
    t0 = time.time()
    if recur == "block":
        o.block_lanczos(np.asarray(W["irec"][:1], np.int32), lld)
    else:
        from rslmtoasa_amd.recursion import chebyshev_scaling
        o.chebyshev(np.asarray(W["irec"][:1], np.int32), lld, *chebyshev_scaling(emin, emax))
    t = time.time() - t0
    return {"value": flops[0] / t * 1e-9, "unit": "GFLOP/s", "cores": oracle.lib().orc_num_threads(), "cpu_quota": cpu_quota(), "host_cpus": os.cpu_count(),
            "kind": "port", "seconds": t, "sites_per_s": 1.0 / t,
            "sample": "1 site of the same %d-atom lattice, LL=%d (%.1f GFLOP), C restatement oracle/rsrec_oracle.c (OpenMP)" % (kk, lld, flops[0] * 1e-9)}


def ldos_stage(rec, gz, ene, nloc, step_s):
    """The LDOS stage for the sites of one step, on the device from the resident coefficients: zsqr + get_terminf + bgreen + the
    reduction of bands.f90:258-268 (rsrec_block_ldos).  Second call = steady state of an SCF loop (buffers exist)."""
    from rslmtoasa_amd.green import Green
    gr = Green(rec, ene)
    for _ in range(2):
        t0 = time.perf_counter()
        r = gr.block_ldos()
        tg = time.perf_counter() - t0
    tmg = rec.timing()
    assert (r["dosial"] == r["dosial"]).all()
    return {"wall_ms": tg * 1e3, "green_kernel_ms": tmg["hop_ms"], "zsqr_terminator_reduction_ms": tmg["rest_ms"], "energies": len(ene),
            "sites_per_s_recursion_plus_ldos": nloc / (step_s + tg),
            "note": "rsrec_block_ldos: zsqr + get_terminf + bgreen + LDOS reduction for the sites of one step, all on the device; "
                    "dtot/dosia/dosial (18 doubles per site and energy) come back; not in `value`"}


def profiled_traffic(workload_key, kernel):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC summaries (profiles/traffic.json, written by
    tools/profile_bench.sh + tools/summarize_pmc.py from separate rocprofv3 --pmc passes; FETCH_SIZE doubled as the guide's gfx950
    correction prescribes).  Returns (bytes, source) or (None, None) when no profile of this workload is committed."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            tab = json.load(f)
        e = tab.get(workload_key, {}).get(kernel)
        if e:
            # a figure measured on other kernel sources is not this run's traffic: refuse it instead of printing a stale number
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from summarize_pmc_sha import kernel_sha
            if e.get("kernel_sha") != kernel_sha():
                return None, "stale: profiles/traffic.json[%s] was measured on other kernel sources (%s); re-run tools/profile_bench.sh" % (workload_key, e.get("kernel_sha", "no hash"))
            return float(e["bytes_per_launch"]), e.get("source")
    except Exception:
        pass
    return None, None


def main():
    args = parse_args()
    if args.workload == "kubo":
        return main_kubo(args)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        launch_ranks(args)                      # never returns
    world = int(env_world or "1")
    if world != args.gpus:
        print("bench.py: WORLD_SIZE=%d but --gpus %d: refusing to report a wrong n_gpus" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # the host side of a rank is one thread driving one GPU: keep the BLAS / OpenMP pools of numpy and torch small
    for k in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ.setdefault(k, "8")
    import numpy as np
    sys.path.insert(0, ROOT)
    from rslmtoasa_amd.lattice import bcc_supercell, spread_sites, supercell_positions
    from rslmtoasa_amd.recursion import Control, Energy, Hamiltonian, Lattice, Recursion, site_partition
    import torch
    dist = None
    # one process per GPU.  BENCH_REHEARSAL=1 (development / the 1-GPU test box only): the ranks share the GPUs that exist and
    # talk over gloo (RCCL refuses two ranks on one device); the collective then runs on a host copy of the device image
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    ndev = torch.cuda.device_count()
    if not rehearsal and world > ndev:
        print("bench.py: --gpus %d but only %d device(s) visible" % (world, ndev), file=sys.stderr)
        sys.exit(2)
    device_index = local_rank % max(ndev, 1) if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    backend = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device_index))   # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus
        backend = dist.get_backend()

    W = build_workload(args, world)
    nn = W["nn"]
    kk = nn.shape[0]
    nsites_total = args.sites * world
    if len(W["irec"]) < nsites_total:
        print("bench.py: %d sites asked for, the lattice has %d" % (nsites_total, len(W["irec"])), file=sys.stderr)
        sys.exit(2)
    lat = Lattice(nn=nn, iz=W["iz"], irec=W["irec"], nmax=W["nmax"], ntype=W["ntype"], cr=W["cr"])
    ham = Hamiltonian(ee=W["ee"], lsham=W["lsham"], eeo=W["eeo"], enim=W["enim"], hall=W["hall"], hallo=W["hallo"], hoh=args.hoh)
    ctl = Control(lld=args.lld, nsp=2, recur="block" if args.recur == "block" else "chebyshev")
    rec = Recursion(ham, lat, ctl, Energy(energy_min=W["emin"], energy_max=W["emax"]), device=device_index, rank=rank, nprocs=world)   # uploads tables: resident before timing
    rec.set_option("graph", 0)      # a measurement run brackets the H|psi> kernels with HIP events; the captured-graph replay of small batches has none
    if args.kernels:
        rec.set_option("kernels", args.kernels)
    if args.batch:
        rec.set_option("batch", args.batch)
    for kv in args.opt:
        k, v = kv.split("=")
        rec.set_option(k, int(v))

    start, end = site_partition(rank, world, nsites_total)
    img = None
    if world > 1:
        # device image of the per-site results of ALL ranks (zero-padded; all-reduce(sum) == all-gather, bands.f90:271-274)
        img = torch.zeros((2, nsites_total, 18, args.lld) if args.recur == "block" else (nsites_total, 2 * args.lld + 2, 18, 18, 2),
                          dtype=torch.float64, device="cuda")

    def step():
        if args.recur == "block":
            rec.recur_b()
        else:
            rec.chebyshev_recur()
        if world > 1:
            # the path's one exchange, on the device: the library writes this rank's part of the image straight into the tensor
            # the collective reduces (no host round trip); gloo rehearsal reduces a host copy
            torch.cuda.synchronize()          # the previous collective (torch's stream) has released the image
            if args.recur == "block":
                rec.pack_diag(start - 1, nsites_total, img[0].data_ptr(), img[1].data_ptr())
            else:
                rec.pack_moments(start - 1, nsites_total, img.data_ptr())        # mu_n of this rank's sites into the zero image, on the device
            if rehearsal:
                hostimg = img.cpu()
                dist.all_reduce(hostimg, op=dist.ReduceOp.SUM)
                img.copy_(hostimg)
            else:
                dist.all_reduce(img, op=dist.ReduceOp.SUM)

    for _ in range(args.warmup):
        step()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    t0 = time.perf_counter()
    tm_acc = {"hop_ms": 0.0, "hop_launches": 0.0, "atom_steps": 0.0, "block_multiplies": 0.0, "total_ms": 0.0, "host_ms": 0.0, "hop_mfma_flop": 0.0,
              "hop_required_flop": 0.0}
    for _ in range(args.steps):
        step()
        tm = rec.timing()
        for k in tm_acc:
            tm_acc[k] += tm[k]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if args.recur == "block":     # the gathered image holds every rank's sites: check this rank's and a remote rank's entries
            a_img = img[0].cpu().numpy()
            mine = rec.a[:args.lld, :, :end - start + 1, 0].transpose(2, 1, 0)
            assert np.array_equal(a_img[start - 1:end], mine), "gathered image does not reproduce this rank's coefficients"
            other = (start - 1 + args.sites) % nsites_total
            assert np.abs(a_img[other]).max() > 0, "gathered image has no data from the other ranks"

    if rank == 0:
        wl_key = W["key"]
        tuned = bool(args.kernels or args.batch or args.no_positions or args.opt)
        # algorithmic work (reference semantics: only blocks whose source atom is inside the active region are multiplied)
        flop_rank = FLOP_PER_BLOCK_MULT * (tm_acc["block_multiplies"] + POST_FLOP_BLOCKS[args.recur] * tm_acc["atom_steps"])
        flop_total = flop_rank * world
        bytes_total = BYTES_PER_ATOM_STEP[args.recur] * tm_acc["atom_steps"] * world
        # dominant kernel = H|psi>: the block SpMM.  Two flop counts:
        #   algorithmic: 46656 per block multiply = the reference's zgemm on full 18x18 blocks (`value` and the CPU baseline use it)
        #   required:    what the operator's block structure needs -- 23328 per spin-diagonal block (the hopping blocks of a collinear
        #                magnet, hamiltonian.f90:1553-1617), 46656 per spin-mixing block (rsrec_get_timing out[9]).  Roofline fractions
        #                are quoted in REQUIRED flops: a kernel cannot exceed the pipe's peak in them.
        # (+ 46656 per atom-step when the kernel also forms the A_n partial: VALU kernels)
        fuses = tm.get("hop_fuses_a", 1.0) and args.recur == "block"
        hop_flop = FLOP_PER_BLOCK_MULT * (tm_acc["block_multiplies"] + (tm_acc["atom_steps"] if fuses else 0.0))
        hop_req = tm_acc["hop_required_flop"] + (FLOP_PER_BLOCK_MULT * tm_acc["atom_steps"] if fuses else 0.0)
        hop_s = tm_acc["hop_ms"] * 1e-3
        achieved_alg = hop_flop / hop_s * 1e-12 if hop_s > 0 else 0.0
        achieved = hop_req / hop_s * 1e-12 if hop_s > 0 else 0.0
        executed = tm_acc["hop_mfma_flop"] / hop_s * 1e-12 if hop_s > 0 else 0.0   # matrix flops the kernel issued (tile padding included)
        step_tflops = flop_total / world / elapsed * 1e-12          # per-GPU whole-level rate, algorithmic count
        req_rank = hop_req + FLOP_PER_BLOCK_MULT * POST_FLOP_BLOCKS[args.recur] * tm_acc["atom_steps"]     # whole level, required count (post-hop blocks are dense)
        step_req_tflops = req_rank / elapsed * 1e-12
        kernel = "k_spmm5"
        traffic, traffic_src = (None, None) if tuned else profiled_traffic(wl_key, kernel)
        out = {
            "metric": ("block-recursion throughput (H|psi> + A_n + B_n recursion levels, recursion.f90 recur_b)" if args.recur == "block" else
                       "Chebyshev-recursion throughput (H|psi> + moment levels, recursion.f90 chebyshev_recur)"),
            "value": flop_total / elapsed * 1e-9,
            "unit": "GFLOP/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": W["data"],
            "config": {"workload": W["name"],
                       "workload_key": wl_key, "sites_per_gpu": args.sites, "atoms": kk, "lld": args.lld, "parity": W.get("parity_note", "coefficients vs the compiled reference at 1e-10 (tests/)"),
                       "parallelism": "site-partition x%d (get_mpi_variables rule), no collective in the loop" % world,
                       "collective": None if world == 1 else "%s all-reduce of the zero-padded per-site image on the %s" % (backend, "host (rehearsal)" if rehearsal else "device")},
            "sites_per_s": nsites_total * args.steps / elapsed,
            "atom_steps_per_s": tm_acc["atom_steps"] * world / elapsed,
            "gbytes_per_s": bytes_total / elapsed * 1e-9,
            "device_ms_per_step": tm_acc["total_ms"] / args.steps,
            "host_ms_per_step": tm_acc["host_ms"] / args.steps,
            # roofline of the dominant kernel (frac = frac_kernel) and of the whole recursion level (frac_step): the headline `value`
            # divided by the same peak -- the whole-level number is the one to compare runs by
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_PEAK_TFLOPS,
                         "frac_kernel": achieved / FP64_PEAK_TFLOPS, "frac_step": step_req_tflops / FP64_PEAK_TFLOPS,
                         "frac_algorithmic": achieved_alg / FP64_PEAK_TFLOPS, "frac_step_algorithmic": step_tflops / FP64_PEAK_TFLOPS,
                         "achieved_algorithmic": achieved_alg, "required_per_algorithmic": hop_req / hop_flop if hop_flop > 0 else None,
                         "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": kernel + " (H|psi> block SpMM, FP64 MFMA)" if not fuses else "H|psi> (hop)", "launches": tm_acc["hop_launches"],
                         "avg_launch_ms": tm_acc["hop_ms"] / max(tm_acc["hop_launches"], 1),
                         "flops_counted": ("required by the operator's block structure: 23328 per spin-diagonal block multiply, 46656 per spin-mixing one "
                                           "(frac, frac_kernel, frac_step); *_algorithmic = the reference's zgemm count, 46656 per block multiply (SURVEY 8d), "
                                           "which includes multiplications by structural zeros and can therefore exceed the peak"),
                         # what the matrix pipe actually did: MFMA flops issued by the kernel (18 -> 20 row padding of the tiles included)
                         "executed": {"achieved": executed, "frac": executed / FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                      "executed_per_algorithmic": tm_acc["hop_mfma_flop"] / hop_flop if hop_flop > 0 else None},
                         "hbm_view": {"achieved": bytes_total / world / elapsed * 1e-9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_total / world / elapsed * 1e-9 / HBM_PEAK_GBS,
                                      "note": "whole recursion level per GPU, algorithmic %d B per atom-step" % BYTES_PER_ATOM_STEP[args.recur]}},
        }
        if world == 1 and not args.no_green and args.recur == "block":
            # Not part of `value`: the LDOS stage behind the recursion for the same sites, entirely on the device from the resident
            # coefficients (zsqr + get_terminf + bgreen + the -Im g0_jj/pi reduction of bands.f90:227-268), so that a true
            # "sites/s from Hamiltonian to LDOS" can be quoted.  2510 energies = the reference's default mesh.
            try:
                gz = np.load(os.path.join(ROOT, "tests", "golden", "bccFe_nsp2_block_green.npz"), allow_pickle=False)
                ene = float(gz["ene_full_first"]) + float(gz["ene_full_step"]) * np.arange(int(gz["nen_full"]))
                out["ldos"] = ldos_stage(rec, gz, ene, args.sites, elapsed / args.steps)
            except Exception as e:  # noqa
                print("LDOS stage skipped: %r" % (e,), file=sys.stderr)
        if world == 1 and not args.no_cpu:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            cb = cpu_baseline(W, args.lld, threads, args.recur, args.hoh)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    rec.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
