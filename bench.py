#!/usr/bin/env python3
"""bench.py -- factorize+solve time of the nested-dissection elimination on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A *step* is one numeric factorization (``factor(A, nd, nd_loc; swlevel=0)``) plus one ``ldiv!`` with a
single right-hand side, with the sparsity pattern analysed and every input (values of A, b) already
resident in HBM when the timed region starts.  Metric (BASELINE.json): factorize+solve time in seconds
(lower is better) on a synthetic 3D problem; N > 1 is STRONG scaling of the same problem (subtrees of the
elimination tree per rank, Schur complements sent point-to-point at the joins, RCCL over xGMI).

One JSON line on rank 0.  ``roofline`` describes the dominant kernel (the FP64 MFMA GEMM): achieved =
its executed flops / its HIP-event time over one extra profiled step; ``traffic`` = measured HBM bytes per launch from a
committed PMC pass of THIS round whose launch count equals the run's (else null, with the reason).  ``cpu_baseline``
times the NumPy oracle (a port of the reference's algorithm, redundant LUs included) and SuperLU on a bounded sample,
in the sample's own seconds -- nothing is extrapolated -- next to this library's time on the same sample.
``factor_oneshot_s`` is the one-shot ``factor`` from host arrays (analysis + numeric), what the Julia shim binds.
``metric_workload`` (N = 1) repeats the measurement on BASELINE.json's own workload class: complex 3-D Helmholtz with
the fronts compressed at tolerance 1e-4 (the largest such problem that fits one GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_DATASHEET = 78.6  # TFLOP/s, AMD MI355X datasheet (FP64 matrix); the guides list no f64 row


def _prepare(hs, name):
    import numpy as np  # noqa: F401

    A, b, nd = hs.problems.make_problem(name, rhs="randn")
    nd, nd_loc = hs.symfact(nd)
    perm = hs.postorder(nd)
    Ap = A[perm - 1][:, perm - 1].tocsc()
    nd = hs.permuted(nd, hs.invperm(perm))
    return Ap, b[perm - 1], nd, nd_loc


def cpu_baseline(sample="auto", dev=None):
    """The CPU baseline of SURVEY.md 8(d): the C restatement of the reference's algorithm (oracle/hs_oracle_c.c: every `\\` a fresh LU, explicit
    L and R, serial recursion -- parallel only inside the BLAS, as Julia + OpenBLAS would run it) timed on the host cores on a bounded sample of
    the headline workload's class, in the sample's own seconds; the NumPy restatement and SuperLU on the small sample and the GPU path on the
    SAME sample beside it.  Nothing is extrapolated.  sample = "auto": Poisson 64^3 when the 32^3 calibration run predicts <= 45 s, else 32^3."""
    import numpy as np
    import scipy.sparse.linalg as spla

    import hsamd
    from oracle import hs_oracle as O
    from oracle import hs_oracle_c as OC

    hs = hsamd.load()
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1

    def problem(name):
        A, b, nd = hs.problems.make_problem(name, rhs="randn")
        o = O.parse_elimtree(*hs.serialize_elimtree(nd))
        o, o_loc = O.symfact(o)
        perm = O.postorder(o)
        Ap = A[perm - 1][:, perm - 1].tocsc()
        o = O.permuted(o, O.invperm(perm))
        return Ap, b[perm - 1], o, o_loc

    def run_c(name, P):
        Ap, bp, o, o_loc = P
        fl = O.tree_flops(o) * (4 if np.iscomplexobj(Ap.data) else 1)
        t0 = time.perf_counter()
        x, info = OC.factor_solve(Ap, o, o_loc, bp, kernels="blas")
        dt = time.perf_counter() - t0
        assert np.isfinite(x).all()
        res = float(np.linalg.norm(Ap @ x - bp) / np.linalg.norm(bp))
        return dict(workload=name, n=int(Ap.shape[0]), seconds=dt, factor_s=info["factor_s"], ldiv_s=info["ldiv_s"], minimal_flops=fl, executed_flops=info["factor_flops"],
                    getrf_calls=info["factor_getrf"], GFLOPs_executed=info["factor_flops"] / info["factor_s"] / 1e9, GFLOPs_minimal=fl / dt / 1e9, residual=res), x

    small = "poisson3d_32"
    Ps = problem(small)
    # BLAS threads: the pool's default is one per hardware thread of the HOST (128), a GPU box's share of it is a fraction: pick the count that
    # is fastest on the calibration sample and use it for everything below; `cores` reports that count
    from threadpoolctl import threadpool_limits

    tried = {}
    host_threads = cores
    c_small, x = None, None
    for k in sorted({k for k in (4, 8, 16, 32, 64, host_threads) if k <= host_threads}):
        with threadpool_limits(limits=k):
            ck, xk = run_c(small, Ps)
        tried[k] = round(ck["seconds"], 3)
        if c_small is None or ck["seconds"] < c_small["seconds"]:
            c_small, x, best = ck, xk, k
    cores = best
    limit = threadpool_limits(limits=cores)
    name, P, c = small, Ps, c_small
    want = sample if sample != "auto" else ("poisson3d_64" if c_small["factor_s"] * 64.0 <= 45.0 else small)  # measured ratio of the two factor times: 64
    if want != small:
        # the larger sample's time sits in fronts of order 4,000-8,000: its thread count is the one that factors a 4,096 x 4,096 matrix fastest
        import scipy.linalg as sla

        M = np.random.default_rng(0).standard_normal((4096, 4096)) + 64.0 * np.eye(4096)
        lu_s = {}
        for k in sorted({k for k in (8, 16, 32, 64, 128, host_threads) if k <= host_threads}):
            with threadpool_limits(limits=k):
                t0 = time.perf_counter()
                sla.lu_factor(M, check_finite=False)
                lu_s[k] = round(time.perf_counter() - t0, 4)
        cores = min(lu_s, key=lu_s.get)
        tried["lu_4096_s"] = lu_s
        limit.restore_original_limits()
        limit = threadpool_limits(limits=cores)
        P = problem(want)
        c, x = run_c(want, P)
        name = want
    out = {
        "value": c["seconds"],
        "unit": "s (factor + ldiv! of the sample workload; not extrapolated)",
        "cores": int(cores),
        "kind": "port",
        "sample": f"C restatement of the reference's dense path (oracle/hs_oracle_c.c, dense kernels = SciPy's OpenBLAS on {cores} threads, tree walked serially) on {name} "
                  f"(n={c['n']}, {c['minimal_flops']:.3g} minimal flops, {c['executed_flops']:.3g} executed in {c['getrf_calls']} LUs + products): factor {c['factor_s']:.2f} s + ldiv! {c['ldiv_s']:.2f} s "
                  f"= {c['GFLOPs_executed']:.1f} GFLOP/s on the executed count, {c['GFLOPs_minimal']:.1f} on the minimal count; residual {c['residual']:.1e}",
        "sample_workload": name,
        "sample_minimal_flops": c["minimal_flops"],
        "c_restatement": c,
        "c_restatement_small": c_small,
        "blas_threads_tried_small_s": tried,
    }
    # the NumPy restatement (round 1-2's baseline) and SuperLU, single-threaded, on the small sample (external yardstick)
    Ap, bp, o, o_loc = Ps
    t0 = time.perf_counter()
    F = O.factor(Ap, o, o_loc, swlevel=0)
    xo = O.ldiv(F, bp)
    out["numpy_port_small_s"] = time.perf_counter() - t0
    del F
    t0 = time.perf_counter()
    lu = spla.splu(Ap)
    xs = lu.solve(bp)
    out["splu_small_s"] = time.perf_counter() - t0
    out["splu_cores"] = 1
    out["small_sample"] = small
    out["oracle_vs_splu_relerr"] = float(np.linalg.norm(xo - xs) / np.linalg.norm(xs))
    del lu
    limit.restore_original_limits()
    # this library on the same sample: factor + ldiv! with the matrix resident
    if dev is not None:
        import torch

        from hierarchicalsolvers_jl_amd import dist as hsdist

        Ap2, bp2, nd2, nd_loc2 = _prepare(hs, name)
        S = hsdist.StagedSolver(Ap2, nd2, nd_loc2, device=dev, swlevel=0)
        bd0 = torch.from_numpy(np.ascontiguousarray(bp2)).to(dev)
        bd = torch.empty_like(bd0)
        for rep in range(2):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            S.numeric()
            bd.copy_(bd0)
            S.solve(bd)
            torch.cuda.synchronize(dev)
            tg = time.perf_counter() - t0
        out["gpu_same_sample_s"] = tg
        out["gpu_same_sample_relerr_vs_c_restatement"] = float(np.linalg.norm(bd.cpu().numpy() - x) / np.linalg.norm(x))
        out["gpu_speedup_on_sample"] = c["seconds"] / tg
        S.backend.L.hs_free(S.backend._h)
        S.backend._h = None
    return out


def pmc_traffic(workload, launches, round_tag="r03"):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of THIS round (tools/pmc_bench.sh; counters cannot be read
    from inside the timed process).  Refused -- null plus the reason -- when the file is missing or was measured on a different launch count."""
    pmc = os.path.join(ROOT, "profiles", "%s_%s_gemm_pmc_traffic.json" % (round_tag, workload))
    if not os.path.exists(pmc):
        return None, "no PMC pass of this round for this workload under profiles/ (%s)" % os.path.basename(pmc), None
    with open(pmc) as f:
        pj = json.load(f)
    if int(pj.get("launches", -1)) != int(launches):
        return None, "profiles/%s was measured on %s launches per factorization, this run has %d: stale, not reported" % (os.path.basename(pmc), pj.get("launches"), launches), None
    src = "profiles/" + os.path.basename(pmc) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes with --kernel-trace only, FETCH_SIZE x2 for gfx950; %d launches)" % launches
    return pj["traffic_bytes_per_launch"], None, src


def metric_workload(hs, hsdist, dev, name, swlevel, tol, steps):
    """BASELINE.json's own workload class on one GPU: complex 3-D Helmholtz, fronts of the top levels compressed at `tol`."""
    import numpy as np
    import torch

    Ap, bp, nd, nd_loc = _prepare(hs, name)
    fopts = dict(swlevel=swlevel, swsize=8, atol=tol, rtol=tol)
    S = hsdist.StagedSolver(Ap, nd, nd_loc, device=dev, **fopts)
    bd0 = torch.from_numpy(np.ascontiguousarray(bp)).to(dev)
    bd = torch.empty_like(bd0)

    def step():
        S.numeric()
        bd.copy_(bd0)
        S.solve(bd)

    step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    per = (time.perf_counter() - t0) / steps
    st = S.stats()
    x = bd.cpu().numpy()
    out = {
        "workload": name, "n": int(Ap.shape[0]), "dtype": "c128" if np.iscomplexobj(Ap.data) else "f64",
        "compression": f"swlevel={swlevel} swsize=8 atol=rtol={tol:g}", "steps": steps, "value": per, "unit": "s",
        "factor_s": st["t_total"], "residual_plain_ldiv": float(np.linalg.norm(Ap @ x - bp) / np.linalg.norm(bp)),
        "maxrank": int(S.backend.L.hs_maxrank(S.backend._h)), "bytes_factors_GiB": st["bytes_factors"] / 2**30,
        "dense_flops_minimal_count": st["flops_factor"],
    }
    out["flow"] = S.backend.flow_info()
    # the scenario's acceptance number: right-preconditioned GMRES(30) to 1e-8 with this factorization (test/rungmres.jl:47-48)
    try:
        import importlib

        hsg = importlib.import_module("hierarchicalsolvers_jl_amd.gmres")  # (the package attribute `gmres` is the function)

        t0 = time.perf_counter()
        xg, hist = hsg.gmres_device(Ap, bd0, S, reltol=1e-8, restart=30, maxiter=30)
        torch.cuda.synchronize(dev)
        out["gmres"] = {"reltol": 1e-8, "restart": 30, "iterations": len(hist) - 1, "seconds": time.perf_counter() - t0,
                        "final_relres": float(hist[-1] / hist[0]) if hist[0] > 0 else 0.0}
    except Exception as e:  # the bench line must not die on the extra
        out["gmres"] = {"error": repr(e)}
    S.backend.L.hs_free(S.backend._h)
    S.backend._h = None
    del S
    torch.cuda.empty_cache()
    # ---- MFMA roofline of THIS workload (BASELINE.json's metric pairs the time with the MFMA utilisation): one more factorization with a HIP
    # event pair around every launch of the two MFMA kernels -- `gemm_op_kernel<cplx>` (trailing / Schur updates of the fronts, hs_stats) and
    # `gemm_probs_kernel<cplx>` (every grouped product of the compressions, hs_probs_stats; its flops are counted by the kernel itself)
    try:
        import ctypes as C

        L = hs._lib.lib()
        L.hs_probs_stats_mode(2)
        Sp = hsdist.StagedSolver(Ap, nd, nd_loc, device=dev, profile=True, **fopts)
        Sp.numeric()
        torch.cuda.synchronize(dev)
        sp = Sp.stats()
        o3 = (C.c_double * 3)()
        L.hs_probs_stats(o3)
        L.hs_probs_stats_mode(0)
        f_op, t_op, n_op = float(sp["gemm_flops"]), float(sp["t_mfma_kernel"]), int(sp["mfma_kernel_launches"])  # (complex launches are already counted as 8 M N K)
        f_pr, n_pr, t_pr = float(o3[0]), int(o3[1]), float(o3[2])
        tot_t = max(t_op + t_pr, 1e-30)
        ach = (f_op + f_pr) / tot_t / 1e12
        out["roofline"] = {
            "bound": "mfma", "kernel": "gemm_op_kernel + gemm_probs_kernel (v_mfma_f64_16x16x4_f64; complex: 4 real MFMAs per k-step, flops = 8 M N K)",
            "achieved": ach, "peak": FP64_MFMA_PEAK_DATASHEET, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_DATASHEET, "traffic": None,
            "gemm_op_kernel": {"flops": f_op, "seconds": t_op, "launches": n_op, "TFLOPs": f_op / max(t_op, 1e-30) / 1e12, "frac": f_op / max(t_op, 1e-30) / 1e12 / FP64_MFMA_PEAK_DATASHEET},
            "gemm_probs_kernel": {"flops": f_pr, "seconds": t_pr, "launches": n_pr, "TFLOPs": f_pr / max(t_pr, 1e-30) / 1e12, "frac": f_pr / max(t_pr, 1e-30) / 1e12 / FP64_MFMA_PEAK_DATASHEET,
                                  "note": "launch durations summed over the concurrent compression streams: they may overlap each other and the fronts' kernels"},
            "factor_s_profiled": float(sp["t_total"]),
            "share_of_factor_time_in_mfma_kernels": min(tot_t / max(float(sp["t_total"]), 1e-30), 1.0),
            "executed_over_dense_minimal_flops": (f_op + f_pr) / max(float(st["flops_factor"]), 1.0),
            "mfma_util_pct_of_factor_time": 100.0 * (f_op + f_pr) / max(float(sp["t_total"]), 1e-30) / 1e12 / FP64_MFMA_PEAK_DATASHEET,
        }
        Sp.backend.L.hs_free(Sp.backend._h)
        Sp.backend._h = None
    except Exception as e:  # the bench line must not die on the extra
        out["roofline"] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("HS_BENCH_WORKLOAD", "poisson3d_128"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="auto", help="workload of the CPU baseline (auto: Poisson 64^3 if the 32^3 calibration predicts <= 45 s, else 32^3)")
    ap.add_argument("--no-oneshot", action="store_true", help="skip the one-shot factor from host arrays (analysis + numeric)")
    ap.add_argument("--metric-workload", default="helmholtz3d_112", help="BASELINE.json's workload class measured beside the headline at N = 1 ('' = skip)")
    ap.add_argument("--metric-swlevel", type=int, default=4)
    ap.add_argument("--metric-tol", type=float, default=1e-4)
    ap.add_argument("--metric-steps", type=int, default=2)
    ap.add_argument("--no-profile", action="store_true", help="skip the extra profiled step that feeds `roofline`")
    ap.add_argument("--swlevel", type=int, default=0, help="compress fronts at tree levels <= swlevel (<0: from the leaves); 0 = exact")
    ap.add_argument("--swsize", type=int, default=8)
    ap.add_argument("--tol", type=float, default=1e-6, help="atol = rtol of the compressed fronts")
    ap.add_argument("--split", type=int, default=0, help="slice width (multiple of 256) in which compressed fronts eliminate their interior block; 0 = off")
    ap.add_argument("--hss-min", type=int, default=0, help="fronts of the compressed levels with at least this many interior DOFs (multiple of 1024) keep D = Aii as an HSS matrix; 0 = dense LU of D")
    ap.add_argument("--hss-dexp", type=int, default=None, help="orders of magnitude by which the HSS form of D is tighter than --tol (default 2)")
    ap.add_argument("--mf", nargs="?", const=1, default=0, type=int, help="matrix-free compressed branch: S travels between the compressed fronts as HSS matrices (the reference's data flow); 1 = interior blocks of those fronts dense (unless --hss-min), 2 = one HSS matrix, 3 = the reference's 2x2 block factorization over HSS blocks")
    ap.add_argument("--dist-top", type=int, default=-1, help="N > 1: fronts above the rank cut eliminated by their whole group of ranks (csrc/hs_dist.h, RCCL inside the library); "
                    "-1 = on for exact runs when the library's communicator passes its self-test on every rank, 0 = off (subtree-per-rank only: the group's first rank eliminates them)")
    ap.add_argument("--dist-min-gbps", type=float, default=25.0, help="--dist-top -1: minimum measured point-to-point rate (ring shift of 128 MiB through the library's communicator, slowest rank); below ~25 GB/s the block-column "
                    "messages of a 32,768 front (4.3 GB of L parts on the critical path of 2 ranks) cost more than the second rank's share of the trailing updates gains (DESIGN.md section 6)")
    ap.add_argument("--leafsize", type=int, default=32, help="SolverOptions.leafsize (HSS leaves; the device uses at least 128)")
    args = ap.parse_args()

    import numpy as np
    import torch

    import hsamd

    hs = hsamd.load()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    dev = torch.device(f"cuda:{local_rank % max(torch.cuda.device_count(), 1)}")
    torch.cuda.set_device(dev)
    comm = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("HS_BENCH_BACKEND", "nccl")  # "gloo" only to rehearse N > 1 on a single GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    from hierarchicalsolvers_jl_amd import dist as hsdist  # registered by hsamd.load()

    # ---- problem + symbolic layer on the host (outside the timed region; test/rungmres.jl:15-19) ----------
    t0 = time.perf_counter()
    A, b, nd = hs.problems.make_problem(args.workload, rhs="randn")
    nd, nd_loc = hs.symfact(nd)
    perm = hs.postorder(nd)
    Ap = A[perm - 1][:, perm - 1].tocsc()
    nd = hs.permuted(nd, hs.invperm(perm))
    bp = b[perm - 1]
    t_host = time.perf_counter() - t0
    is_c = np.iscomplexobj(Ap.data)

    fopts = dict(swlevel=args.swlevel, swsize=args.swsize, atol=args.tol, rtol=args.tol, split_size=args.split, hss_min=args.hss_min, hss_dexp=args.hss_dexp, mf=args.mf, leafsize=args.leafsize) if args.swlevel != 0 else dict(swlevel=0)
    libcomm, dist_note, p2p_gbps, p2p_all = None, None, 0.0, None
    if world > 1 and args.dist_top != 0 and args.swlevel == 0:
        # the library's own communicator (RCCL over xGMI under the nccl process group): used only if its ring self-test passes on EVERY rank
        # AND moves at least --dist-min-gbps per link (a fan-out slower than that would cost more than the idle ranks gain)
        ok, bw = 1, 0.0
        try:
            libcomm = hsdist.LibComm(rank, world, dev)
            libcomm.selftest(1 << 22)
            bw = libcomm.bandwidth(1 << 27, 4)
        except Exception as e:  # noqa: BLE001
            ok, dist_note = 0, f"rank {rank}: {e!r}"
        on_dev = torch.distributed.get_backend() == "nccl"
        flag = torch.tensor([float(ok), bw], dtype=torch.float64, device=dev if on_dev else "cpu")
        bw_all = [torch.zeros(1, dtype=torch.float64, device=flag.device) for _ in range(world)]
        torch.distributed.all_gather(bw_all, flag[1:2].clone())
        p2p_all = [float(t.item()) for t in bw_all]  # every rank's measured rate: a slow link shows up by its rank
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        p2p_gbps = float(flag[1].item())
        if int(flag[0].item()) == 1 and (p2p_gbps >= args.dist_min_gbps or libcomm.kind() == "host" or args.dist_top == 1):
            fopts["dist_top"] = True
        else:
            dist_note = dist_note or ("the communicator self-test failed on another rank" if int(flag[0].item()) != 1 else
                                      f"point-to-point rate {p2p_gbps:.1f} GB/s below --dist-min-gbps {args.dist_min_gbps:g}")
            libcomm = None
    t0 = time.perf_counter()
    try:
        S = hsdist.StagedSolver(Ap, nd, nd_loc, rank=rank, nranks=world, device=dev, libcomm=libcomm, **fopts)
    except Exception as e:  # noqa: BLE001 -- the plan refuses the group fronts (e.g. a branch of the tree ends above the rank cut): same decision on every rank
        if not fopts.get("dist_top"):
            raise
        dist_note = f"dist_top refused by the plan: {e}"
        fopts.pop("dist_top")
        libcomm = None
        S = hsdist.StagedSolver(Ap, nd, nd_loc, rank=rank, nranks=world, device=dev, **fopts)
    torch.cuda.synchronize(dev)
    t_analyze = time.perf_counter() - t0  # hs_analyze: pattern upload, descriptors, hipMalloc of the factor arena (once per pattern)
    b_dev0 = torch.from_numpy(np.ascontiguousarray(bp)).to(dev)
    b_dev = torch.empty_like(b_dev0)

    def step():
        S.numeric()
        b_dev.copy_(b_dev0)
        S.solve(b_dev)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    per_step = dt / max(args.steps, 1)
    st = S.stats()
    # N > 1, host-driven joins: one more, INSTRUMENTED factorization (the host waits for the device after every stage; not part of `value`):
    # what each rank spent computing, and sending / receiving -- which includes waiting for the peer -- so a first run on real links is diagnosable
    rank_times = None
    if world > 1 and not fopts.get("dist_top"):
        tr = []
        S.numeric(trace=tr)
        mine = {"rank": rank, "compute_s": sum(t for w, t in tr if w.startswith("level")), "transfer_and_wait_s": sum(t for w, t in tr if w.startswith(("send", "recv"))),
                "stages": [(w, round(t, 6)) for w, t in tr], "hss_bytes_moved": getattr(S.backend, "hss_bytes_moved", 0)}
        rank_times = [None] * world
        torch.distributed.all_gather_object(rank_times, mine)
    # ldiv! alone (part of every timed step; timed separately here for the HBM roofline of the solve kernels)
    t_ldiv = 0.0
    if world == 1:
        sync()
        t0 = time.perf_counter()
        for _ in range(3):
            b_dev.copy_(b_dev0)
            S.solve(b_dev)
        sync()
        t_ldiv = (time.perf_counter() - t0) / 3
    # ... and once more level by level (hs_solve_*_levels with events between the levels): GB/s of the sweeps of every tree level against the
    # bytes they must read, (ni^2 + 2 ni nb) * sizeof(T) per front
    solve_levels = None
    if world == 1 and not fopts.get("dist_top"):
        try:
            nl = S.plan.nlevels
            st0 = S.stats()
            lev_bytes = {}
            import ctypes as _C

            for node in range(int(st0["nnodes"])):
                ni, nb, lv = _C.c_int64(), _C.c_int64(), _C.c_int64()
                hs._lib.check(S.backend.L.hs_node_info(S.backend._h, node, _C.byref(ni), _C.byref(nb), _C.byref(lv)))
                lev_bytes[lv.value] = lev_bytes.get(lv.value, 0) + (ni.value * ni.value + 2 * ni.value * nb.value) * (16 if is_c else 8)
            b_dev.copy_(b_dev0)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(2 * nl + 1)]
            evs[0].record()
            for i, lv in enumerate(range(nl, 0, -1)):
                S.backend.fwd(b_dev, lv, lv)
                evs[i + 1].record()
            for i, lv in enumerate(range(1, nl + 1)):
                S.backend.bwd(b_dev, lv, lv)
                evs[nl + i + 1].record()
            sync()
            fw = {lv: evs[i].elapsed_time(evs[i + 1]) for i, lv in enumerate(range(nl, 0, -1))}
            bw = {lv: evs[nl + i].elapsed_time(evs[nl + i + 1]) for i, lv in enumerate(range(1, nl + 1))}
            solve_levels = [{"level": lv, "ms": round(fw[lv] + bw[lv], 3), "GBps": round(lev_bytes.get(lv, 0) / ((fw[lv] + bw[lv]) * 1e-3) / 1e9, 0) if fw[lv] + bw[lv] > 0 else None}
                            for lv in range(1, nl + 1)]
        except Exception as e:  # diagnostics only: never fail the bench line for it
            solve_levels = "unavailable: %r" % (e,)

    # correctness of what was timed: residual of the last solve (never skipped)
    x = b_dev.cpu().numpy()
    res = float(np.linalg.norm(Ap @ x - bp) / np.linalg.norm(bp))

    # ---- roofline of the dominant kernel: one more step with per-launch HIP events --------------------------
    roofline = None
    if not args.no_profile:
        S.backend.L.hs_free(S.backend._h)
        S.backend._h = None
        del S
        torch.cuda.empty_cache()
        S = hsdist.StagedSolver(Ap, nd, nd_loc, rank=rank, nranks=world, device=dev, profile=True, libcomm=libcomm, **fopts)
        S.numeric()
        sp_ = S.stats()
        if sp_["t_mfma_kernel"] > 0:
            # every launch of the kernel `gemm_op_kernel` (trailing and Schur updates; the 32-row TRSM base cases run the same
            # tile code under the name trsm_inv_kernel): `launches` and `avg_launch_ms` are what rocprofv3 --stats shows for it
            t_k, n_k = sp_["t_mfma_kernel"], int(sp_["mfma_kernel_launches"])
            ach = sp_["gemm_flops"] / t_k / 1e12
            roofline = {
                "bound": "mfma",
                "kernel": "gemm_op_kernel<%s> (v_mfma_f64_16x16x4_f64)" % ("cplx" if is_c else "double"),
                "achieved": ach,
                "peak": FP64_MFMA_PEAK_DATASHEET,
                "unit": "TFLOP/s",
                "frac": ach / FP64_MFMA_PEAK_DATASHEET,
                "traffic": None,
                "launches": n_k,
                "avg_launch_ms": t_k * 1e3 / max(n_k, 1),
                "kernel_time_s": t_k,
                "kernel_flops": sp_["gemm_flops"],
                "flops_per_launch": sp_["gemm_flops"] / max(n_k, 1),
                "share_of_factor_time": t_k / max(sp_["t_total"], 1e-30),
                "phases_s": {"gemm_updates": sp_["t_gemm"], "panel": sp_["t_panel"], "laswp+trsm": sp_["t_trsm"], "assemble": sp_["t_assemble"],
                             "factor_total_profiled": sp_["t_total"]},
            }
            # HBM bytes per launch: algorithmic (A and B read once, C read and written once, summed over the launches) and measured
            # (committed PMC passes of this round, refused when their launch count is not this run's)
            roofline["algorithmic_bytes_per_launch"] = sp_["gemm_bytes"] / max(n_k, 1)
            roofline["flop_per_algorithmic_byte"] = sp_["gemm_flops"] / max(sp_["gemm_bytes"], 1.0)
            if world == 1 and args.swlevel == 0:
                tr, why, src = pmc_traffic(args.workload, n_k)
                roofline["traffic"] = tr
                if tr is None:
                    roofline["traffic_note"] = why
                else:
                    roofline["traffic_source"] = src
                    roofline["traffic_over_algorithmic"] = tr / max(roofline["algorithmic_bytes_per_launch"], 1.0)
            else:
                roofline["traffic_note"] = "PMC passes are collected for the single-GPU exact run only"
            peak_meas = S.backend.L.hsk_mfma_f64_peak(2, 100000)
            roofline["peak_measured_issue_rate"] = peak_meas
            roofline["frac_of_measured"] = ach / peak_meas if peak_meas > 0 else None

    maxrank_main = int(S.backend.L.hs_maxrank(S.backend._h)) if getattr(S.backend, "_h", None) else None
    flow_main = S.backend.flow_info() if getattr(S.backend, "_h", None) else None  # what the library did with the options on rank 0
    oneshot = None
    extra_metric = None
    if world == 1:
        # release the resident factorization (139 GiB at 128^3) before anything else allocates
        S.backend.L.hs_free(S.backend._h)
        S.backend._h = None
        del S
        torch.cuda.empty_cache()
        if not args.no_oneshot:
            # what the Julia shim binds: factor(A, nd, nd_loc) from HOST arrays = analysis + arena allocation + upload + numeric, then ldiv!
            # from a host vector (PCIe-inclusive; never `value`)
            t0 = time.perf_counter()
            F1 = hs.factor(Ap, nd, nd_loc, **fopts)
            t_f1 = time.perf_counter() - t0
            t0 = time.perf_counter()
            x1 = hs.ldiv(F1, bp)
            t_l1 = time.perf_counter() - t0
            oneshot = {"factor_oneshot_s": t_f1, "ldiv_host_s": t_l1, "residual": float(np.linalg.norm(Ap @ x1 - bp) / np.linalg.norm(bp))}
            F1.free()
            del F1
        if args.metric_workload and args.swlevel == 0 and args.workload != args.metric_workload:
            try:
                extra_metric = metric_workload(hs, hsdist, dev, args.metric_workload, args.metric_swlevel, args.metric_tol, args.metric_steps)
            except Exception as e:  # never lose the headline line to the extra
                extra_metric = {"workload": args.metric_workload, "error": repr(e)}

    if rank == 0:
        flops = st["flops_factor"] if world == 1 else None
        out = {
            "metric": "factorize+solve time, synthetic 3D problem, " + ("exact (swlevel=0)" if args.swlevel == 0 else "compressed (low-rank off-diagonal blocks)") + " nested-dissection elimination",
            "value": per_step,
            "unit": "s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": per_step * 1e3,
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "c128" if is_c else "f64",
            "data": "synthetic",
            "config": {"workload": args.workload, "n": int(Ap.shape[0]), "nnz": int(Ap.nnz), "tree_nodes": int(st["nnodes"]),
                       "tree_depth": int(st["nlevels"]), "max_front": [int(st["max_ni"]), int(st["max_nb"])], "nrhs": 1,
                       "compression": "none (swlevel=0)" if args.swlevel == 0 else f"{'matrix-free HSS hand-over' if args.mf else 'low-rank off-diagonal blocks'}, swlevel={args.swlevel} swsize={args.swsize} atol=rtol={args.tol:g} split={args.split} hss_min={args.hss_min} mf={int(args.mf)}",
                       "partition": f"subtree-per-rank x{world}" + ((" + fronts above the cut over their rank groups (1-D block-cyclic block columns, " + libcomm.kind() + ")") if fopts.get("dist_top") else
                                                                      ((" + fronts above the cut on the first rank of their group; the joins ship " +
                                                                        ("packed HSS generators (hs_schur_pack / hs_schur_unpack)" if (flow_main or {}).get("mf") else "dense Schur complements")) if world > 1 else ""))},
            "flow": flow_main,
            "factor_s": st["t_total"],
            "residual": res,
            "maxrank": maxrank_main,
            "bytes_factors_GiB": st["bytes_factors"] / 2**30,
            "host_symbolic_s": t_host,
            "analyze_s": t_analyze,
        }
        if dist_note:
            out["dist_top_note"] = dist_note
        if world > 1 and args.dist_top != 0 and args.swlevel == 0:
            out["comm_p2p_GBps"] = p2p_gbps  # measured through the library's communicator before the run (slowest rank)
            out["comm_p2p_GBps_per_rank"] = p2p_all
        if rank_times:
            out["rank_times"] = rank_times
        if flops:
            out["factor_tflops_minimal_count"] = flops / st["t_total"] / 1e12
        if world == 1 and t_ldiv > 0 and st["bytes_solve"] > 0:
            # ldiv! is HBM-bound: every stored factor entry is read once per right-hand side
            gbs = st["bytes_solve"] / t_ldiv / 1e9
            out["solve"] = {"seconds": t_ldiv, "algorithmic_bytes": st["bytes_solve"], "achieved_GBps": gbs, "frac_of_spec_8000GBps": gbs / 8000.0,
                            "frac_of_measured_copy_bw_6290GBps": gbs / 6290.0, "levels": solve_levels}
        if roofline:
            out["roofline"] = roofline
            out["mfma_util_pct"] = 100.0 * roofline["frac"]  # BASELINE.json's metric pairs the time with the MFMA utilisation (FP64-matrix peak 78.6 TF)
        if oneshot:
            out["factor_oneshot_s"] = oneshot["factor_oneshot_s"]  # analysis + numeric from host arrays (hs_factor_*), excluded from `value`
            out["oneshot"] = oneshot
        if extra_metric:
            out["metric_workload"] = extra_metric
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, dev)
        print(json.dumps(out))
    if world > 1:
        # release the factorization before the communicator it borrows, and that before the process group / the HIP runtime go away
        try:
            if getattr(S.backend, "_h", None):
                S.backend.L.hs_free(S.backend._h)
                S.backend._h = None
        except NameError:
            pass
        torch.cuda.synchronize(dev)
        torch.distributed.barrier()
        if libcomm is not None:
            libcomm.close()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
