"""Error of the matrix-free block flow at round-off tolerance (tests/test_mf_gpu.py's last assertion) for one case: CASE TOL MF."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import scipy.sparse.linalg as spla

import hsamd

hs = hsamd.load()
from helpers import prepare, relerr

shape = tuple(int(x) for x in sys.argv[1].split("x"))
kind = sys.argv[2]
mf = sys.argv[3] if len(sys.argv) > 3 else "block"
P = prepare(hs, shape, rhs="randn", kind=kind, nmax=512)
xr = spla.splu(P["A"]).solve(P["b"])
for tol in [float(x) for x in os.environ.get('PROBE_TOLS', '1e-6,1e-9,1e-12').split(',')]:
    Fx = hs.factor(P["A"], P["nd"], P["nd_loc"], mf=mf, swlevel=2, swsize=8, atol=tol, rtol=tol, leafsize=128)
    print(f"{shape} {kind} mf={mf} tol={tol:g}: err {relerr(hs.ldiv(Fx, P['b']), xr):.3e} maxrank {hs.maxrank(Fx)}", flush=True)
Fd = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
print(f"exact: err {relerr(hs.ldiv(Fd, P['b']), xr):.3e}")
