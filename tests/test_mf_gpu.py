"""The matrix-free compressed branch on the GPU (hs_options.mf; hs_mffront.h): Schur complements travel between fronts as HSS
matrices, a parent of two such fronts is assembled from their generators and the sparse couplings of A -- rows C1, C3, C5, C6,
B2', F2 of SURVEY.md section 8 (src/factorization.jl:78-112,126-140,184-249).  HssMatrices.jl / LowRankApprox.jl are absent from the
reference tree: PARITY UNPINNED; checked against SuperLU, against the dense-S compressed path and against oracle/hs_oracle_mf.py."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from helpers import prepare, relerr

pytestmark = pytest.mark.gpu

CASES = [
    (((32, 32, 32), dict(kind="poisson", nmax=512)), 3),
    (((24, 24, 24), dict(kind="helmholtz", nmax=512)), 2),
]


@pytest.mark.parametrize("name,swlevel", CASES)
@pytest.mark.parametrize("tol", [1e-2, 1e-6])
def test_matrix_free_branch_accuracy(hs, name, swlevel, tol):
    P = prepare(hs, name[0], rhs="randn", **name[1])
    kw = dict(swlevel=swlevel, swsize=8, atol=tol, rtol=tol, leafsize=128)
    Fm = hs.factor(P["A"], P["nd"], P["nd_loc"], mf=True, verbose=True, **kw)
    Fd = hs.factor(P["A"], P["nd"], P["nd_loc"], **kw)
    xr = spla.splu(P["A"]).solve(P["b"])
    em, ed = relerr(hs.ldiv(Fm, P["b"]), xr), relerr(hs.ldiv(Fd, P["b"]), xr)
    sm, sd = Fm.stats(), Fd.stats()
    print(f"{name[0]} tol={tol:g}: err matrix-free {em:.2e}, dense-S path {ed:.2e}; maxrank {hs.maxrank(Fm)} / {hs.maxrank(Fd)}; "
          f"factor bytes {sm['bytes_factors'] / 2**20:.1f} / {sd['bytes_factors'] / 2**20:.1f} MiB; factor {sm['t_total'] * 1e3:.0f} / {sd['t_total'] * 1e3:.0f} ms")
    assert hs.maxrank(Fm) > 0
    assert em <= max(100 * ed, 1e4 * tol), (em, ed)
    assert relerr(hs.ldiv(Fm, P["b"]), xr) <= max(100 * ed, 1e4 * tol)  # a second solve gives the same answer
