"""The reference's scenario (test/rungmres.jl:15-52) on a generated problem: exact factorization, compressed
factorization, GMRES(30) right-preconditioned by each.  Prints one JSON line per run (diagnostic, not a test).

    python tools/run_gmres_scenario.py poisson3d_64 [swlevel] [tol] [swsize] [hss_min] [hss_dexp]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hsamd

hs = hsamd.load()
name = sys.argv[1] if len(sys.argv) > 1 else "poisson3d_64"
swlevel = int(sys.argv[2]) if len(sys.argv) > 2 else -4
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-2
swsize = int(sys.argv[4]) if len(sys.argv) > 4 else 8
hss_min = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # > 0: also run the compressed factorization with D kept as HSS
hss_dexp = int(sys.argv[6]) if len(sys.argv) > 6 else None

A, b, nd = hs.problems.make_problem(name, rhs="randn")
nd, nd_loc = hs.symfact(nd)
perm = hs.postorder(nd)
A = A[perm - 1][:, perm - 1].tocsc()
nd = hs.permuted(nd, hs.invperm(perm))
b = b[perm - 1]
import torch

runs = [("exact", dict(swlevel=0)), ("compressed", dict(swlevel=swlevel, swsize=swsize, atol=tol, rtol=tol))]
if hss_min > 0:
    runs.append(("compressed, HSS D", dict(swlevel=swlevel, swsize=swsize, atol=tol, rtol=tol, hss_min=hss_min, hss_dexp=hss_dexp)))
for label, kw in runs:
    hs.factor(A, nd, nd_loc, **kw).free()  # warm-up (kernel load, allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    F = hs.factor(A, nd, nd_loc, **kw)
    torch.cuda.synchronize()
    tf = time.perf_counter() - t0
    t0 = time.perf_counter()
    x = hs.ldiv(F, b)
    ts = time.perf_counter() - t0
    res = float(np.linalg.norm(A @ x - b) / np.linalg.norm(b))
    t0 = time.perf_counter()
    xg, ch = hs.gmres(A, b, Pr=F, reltol=1e-9, restart=30, maxiter=30, log=True)
    tg = time.perf_counter() - t0
    resg = float(np.linalg.norm(A @ xg - b) / np.linalg.norm(b))
    print(json.dumps(dict(problem=name, n=int(A.shape[0]), dtype=str(A.dtype), factorization=label, opts=kw, factor_s_host_buffers=tf, ldiv_s=ts,
                          ldiv_residual=res, maxrank=int(hs.maxrank(F)), gmres_iters=int(ch["iters"]), gmres_converged=bool(ch["isconverged"]),
                          gmres_s=tg, gmres_residual=resg)), flush=True)
    F.free()
