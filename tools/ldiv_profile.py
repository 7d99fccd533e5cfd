"""Factor once, then `ldiv!` a few times (run under `rocprofv3 --kernel-trace` by tools/ldiv_profile.sh): WORKLOAD [nsolves]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hsamd

hs = hsamd.load()
name = sys.argv[1] if len(sys.argv) > 1 else "poisson3d_128"
nsolve = int(sys.argv[2]) if len(sys.argv) > 2 else 4
A, b, nd = hs.problems.make_problem(name, rhs="randn")
nd, nd_loc = hs.symfact(nd)
perm = hs.postorder(nd)
A = A[perm - 1][:, perm - 1].tocsc()
nd = hs.permuted(nd, hs.invperm(perm))
b = b[perm - 1]
F = hs.factor(A, nd, nd_loc, swlevel=0)
torch.cuda.synchronize()
for _ in range(nsolve):
    x = hs.ldiv(F, b)
torch.cuda.synchronize()
print("residual", float(np.linalg.norm(A @ x - b) / np.linalg.norm(b)))
