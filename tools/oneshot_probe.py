"""Where the time of the one-shot `factor(A, nd, nd_loc)` from HOST arrays goes (8.78 s against analyze 0.50 + numeric 3.31).
Usage (GPU box): python tools/oneshot_probe.py [WORKLOAD]   -- prints wall seconds of every stage, first and second call in one process."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hsamd

hs = hsamd.load()
import torch

name = sys.argv[1] if len(sys.argv) > 1 else "poisson3d_128"
t0 = time.perf_counter()
A, b, nd = hs.problems.make_problem(name, rhs="randn")
nd, nd_loc = hs.symfact(nd)
perm = hs.postorder(nd)
Ap = A[perm - 1][:, perm - 1].tocsc()
nd = hs.permuted(nd, hs.invperm(perm))
print(f"host symbolic layer {time.perf_counter() - t0:.2f} s", flush=True)
torch.cuda.init()
for rep in range(3):
    t0 = time.perf_counter()
    os.environ["HS_VERBOSE_ONESHOT"] = "1"
    F = hs.factor(Ap, nd, nd_loc, swlevel=0)
    t1 = time.perf_counter()
    st = F.stats()
    print(f"call {rep}: hs.factor {t1 - t0:.3f} s wall; device time of the numeric phase {st['t_total']:.3f} s", flush=True)
    t0 = time.perf_counter()
    F.free()
    torch.cuda.synchronize()
    print(f"call {rep}: hs_free {time.perf_counter() - t0:.3f} s", flush=True)
print(f"hs.trim() released {hs.trim() / 2**30:.1f} GiB")
