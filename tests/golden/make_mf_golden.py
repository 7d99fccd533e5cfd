"""Generate tests/golden/mf_oracle.json: what the CPU restatement of the matrix-free compressed branch (oracle/hs_oracle_mf.py,
`dmode="single"` and `dmode="block"`) produces on the cases of tests/test_mf_gpu.py -- solution error against SuperLU, maxrank, and per matrix-free front the
HSS rank of D, rank(L), rank(R) and the HSS rank of S.  DATA ONLY; the oracle takes about a minute per case, too long for the GPU suite.

    python tests/golden/make_mf_golden.py

The reference (Julia + HssMatrices.jl + LowRankApprox.jl) cannot run here: PARITY UNPINNED, these numbers pin the device path to the
CPU restatement of the same data flow, not to the Julia package."""
import json
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hsamd  # noqa: E402
from helpers import prepare, relerr  # noqa: E402
from oracle import hs_hss as HS, hs_oracle_lr as OL, hs_oracle_mf as OM  # noqa: E402

CASES = {
    "poisson3d_32": (((32, 32, 32), dict(kind="poisson", nmax=512)), 3),
    "helmholtz3d_24": (((24, 24, 24), dict(kind="helmholtz", nmax=512)), 2),
}
TOLS = (1e-2, 1e-6)


def walk(F, out):
    for c in (F.left, F.right):
        if c is not None:
            walk(c, out)
    if F.kind in ("mf", "lr"):
        out.append(dict(kind=F.kind, ni=int(len(F.int)), nb=int(len(F.bnd)),
                        hssrank_D=int(F.D.hssrank) if isinstance(F.D, (OM.SingleD, OM.BlockD)) else 0,
                        rank_L=int(F.L.rank) if isinstance(F.L, OL.LowRankMatrix) else 0,
                        rank_R=int(F.R.rank) if isinstance(F.R, OL.LowRankMatrix) else 0,
                        hssrank_S=int(HS.hssrank(F.S)) if isinstance(F.S, HS.Hss) else 0))


def main():
    hs = hsamd.load()
    res = {}
    for cname, (name, swlevel) in CASES.items():
        P = prepare(hs, name[0], rhs="randn", **name[1])
        xr = spla.splu(P["A"]).solve(P["b"])
        for tol in TOLS:
            # "single": D of a matrix-free front as one HSS matrix (hs_options.mf = 2); "block": the reference's 2x2 blockfactor (mf = 3)
            for dmode, suffix in (("single", ""), ("block", "/block")):
                F = OM.factor(P["A"], P["ond"], P["ond_loc"], dexp=2, dmode=dmode, swlevel=swlevel, swsize=8, atol=tol, rtol=tol, leafsize=128)
                nodes = []
                walk(F, nodes)
                key = f"{cname}/tol={tol:g}{suffix}"
                res[key] = dict(err_vs_splu=float(relerr(OM.ldiv(F, P["b"]), xr)), maxrank=int(OM.maxrank(F)), fronts=nodes,
                                options=dict(swlevel=swlevel, swsize=8, atol=tol, rtol=tol, leafsize=128, dexp=2, dmode=dmode))
                print(key, res[key]["err_vs_splu"], res[key]["maxrank"], flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mf_oracle.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
