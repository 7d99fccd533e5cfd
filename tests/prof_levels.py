"""Per-level timing of one profiled numeric factorization (diagnostic, not a test)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hsamd
hs = hsamd.load()
from hierarchicalsolvers_jl_amd import dist as hsdist
w = sys.argv[1] if len(sys.argv) > 1 else "poisson3d_96"
A, b, nd = hs.problems.make_problem(w, rhs="randn")
nd, nd_loc = hs.symfact(nd); perm = hs.postorder(nd)
Ap = A[perm-1][:, perm-1].tocsc(); nd = hs.permuted(nd, hs.invperm(perm))
S = hsdist.StagedSolver(Ap, nd, nd_loc, swlevel=0, profile=True)
S.numeric(); S.numeric()
st = S.stats(); print({k: (round(v,4) if isinstance(v,float) else v) for k,v in st.items() if k.startswith("t_") or k.startswith("gemm")})
