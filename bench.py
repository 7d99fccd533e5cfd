#!/usr/bin/env python3
"""bench.py -- factorize+solve time of the nested-dissection elimination on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]

A *step* is one numeric factorization (``factor(A, nd, nd_loc; swlevel=0)``) plus one ``ldiv!`` with a
single right-hand side, with the sparsity pattern analysed and every input (values of A, b) already
resident in HBM when the timed region starts.  Metric (BASELINE.json): factorize+solve time in seconds
(lower is better) on a synthetic 3D problem; N > 1 is STRONG scaling of the same problem (subtrees of the
elimination tree per rank, Schur complements sent point-to-point at the joins, RCCL over xGMI).

One JSON line on rank 0.  ``roofline`` describes the dominant kernel (the FP64 MFMA GEMM): achieved =
its executed flops / its HIP-event time over the timed region's last step.  ``cpu_baseline`` times the
NumPy oracle (a port of the reference's algorithm, redundant LUs included) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_DATASHEET = 78.6  # TFLOP/s, AMD MI355X datasheet (FP64 matrix); the guides list no f64 row


def cpu_baseline(flops_full, sample="poisson3d_32", complex_=False):
    """Oracle (port of the reference's algorithm) timed on the host cores on a bounded sample."""
    import numpy as np

    import hsamd
    from oracle import hs_oracle as O

    hs = hsamd.load()
    try:
        from threadpoolctl import threadpool_info

        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    A, b, nd = hs.problems.make_problem(sample)
    o = O.parse_elimtree(*hs.serialize_elimtree(nd))
    o, o_loc = O.symfact(o)
    perm = O.postorder(o)
    Ap = A[perm - 1][:, perm - 1].tocsc()
    o = O.permuted(o, O.invperm(perm))
    fl = O.tree_flops(o) * (4 if complex_ else 1)
    t0 = time.perf_counter()
    F = O.factor(Ap, o, o_loc, swlevel=0)
    x = O.ldiv(F, b[perm - 1])
    dt = time.perf_counter() - t0
    assert np.isfinite(x).all()
    return {
        "value": dt * flops_full / fl,
        "unit": "s (extrapolated to the full workload by minimal-flop ratio)",
        "cores": int(cores),
        "kind": "port",
        "sample": f"oracle factor+ldiv on {sample} (n={A.shape[0]}, {fl:.3g} minimal flops) took {dt:.2f} s = {fl / dt / 1e9:.2f} GFLOP/s",
        "sample_seconds": dt,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("HS_BENCH_WORKLOAD", "poisson3d_128"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="poisson3d_32")
    ap.add_argument("--no-profile", action="store_true", help="skip the extra profiled step that feeds `roofline`")
    ap.add_argument("--swlevel", type=int, default=0, help="compress fronts at tree levels <= swlevel (<0: from the leaves); 0 = exact")
    ap.add_argument("--swsize", type=int, default=8)
    ap.add_argument("--tol", type=float, default=1e-6, help="atol = rtol of the compressed fronts")
    ap.add_argument("--split", type=int, default=0, help="slice width (multiple of 256) in which compressed fronts eliminate their interior block; 0 = off")
    ap.add_argument("--hss-min", type=int, default=0, help="fronts of the compressed levels with at least this many interior DOFs (multiple of 1024) keep D = Aii as an HSS matrix; 0 = dense LU of D")
    ap.add_argument("--hss-dexp", type=int, default=None, help="orders of magnitude by which the HSS form of D is tighter than --tol (default 2)")
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

    fopts = dict(swlevel=args.swlevel, swsize=args.swsize, atol=args.tol, rtol=args.tol, split_size=args.split, hss_min=args.hss_min, hss_dexp=args.hss_dexp) if args.swlevel != 0 else dict(swlevel=0)
    t0 = time.perf_counter()
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
        S = hsdist.StagedSolver(Ap, nd, nd_loc, rank=rank, nranks=world, device=dev, profile=True, **fopts)
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
            # HBM bytes per launch from the committed PMC passes of the same workload (tools/pmc_bench.sh; counters
            # cannot be read from inside the process)
            pmc = os.path.join(ROOT, "profiles", "r01_%s_gemm_pmc_traffic.json" % args.workload)
            if world == 1 and args.swlevel == 0 and os.path.exists(pmc):
                with open(pmc) as f:
                    pj = json.load(f)
                roofline["traffic"] = pj["traffic_bytes_per_launch"]
                roofline["traffic_source"] = "profiles/" + os.path.basename(pmc) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 for gfx950)"
            peak_meas = S.backend.L.hsk_mfma_f64_peak(2, 100000)
            roofline["peak_measured_issue_rate"] = peak_meas
            roofline["frac_of_measured"] = ach / peak_meas if peak_meas > 0 else None

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
                       "compression": "none (swlevel=0)" if args.swlevel == 0 else f"low-rank off-diagonal blocks, swlevel={args.swlevel} swsize={args.swsize} atol=rtol={args.tol:g} split={args.split} hss_min={args.hss_min}",
                       "partition": f"subtree-per-rank x{world}"},
            "factor_s": st["t_total"],
            "residual": res,
            "maxrank": int(S.backend.L.hs_maxrank(S.backend._h)) if getattr(S.backend, "_h", None) else None,
            "host_symbolic_s": t_host,
            "analyze_s": t_analyze,
        }
        if flops:
            out["factor_tflops_minimal_count"] = flops / st["t_total"] / 1e12
        if world == 1 and t_ldiv > 0 and st["bytes_solve"] > 0:
            # ldiv! is HBM-bound: every stored factor entry is read once per right-hand side
            gbs = st["bytes_solve"] / t_ldiv / 1e9
            out["solve"] = {"seconds": t_ldiv, "algorithmic_bytes": st["bytes_solve"], "achieved_GBps": gbs,
                            "frac_of_measured_copy_bw_6290GBps": gbs / 6290.0}
        if roofline:
            out["roofline"] = roofline
            out["mfma_util_pct"] = 100.0 * roofline["frac"]  # BASELINE.json's metric pairs the time with the MFMA utilisation (FP64-matrix peak 78.6 TF)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(st["flops_factor"], args.cpu_sample, is_c)
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
