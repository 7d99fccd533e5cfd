"""CPU tests of the oracle of the compressed branch with an HSS interior block (oracle/hs_oracle_hss.py) and, on the GPU, the
product's `hss_min` path against it.  PARITY UNPINNED (HssMatrices.jl / LowRankApprox.jl are not in the reference tree):
both sides are O(tol) perturbations of the same factorization, so solution errors and ranks are compared, not entries."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from helpers import prepare, relerr
from oracle import hs_oracle as O, hs_oracle_hss as OH


@pytest.mark.parametrize("name", [((16, 16, 16), dict(kind="poisson", nmax=64)), ((12, 12, 12), dict(kind="helmholtz", nmax=64))])
def test_oracle_hss_interior_blocks(hs, name):
    P = prepare(hs, name[0], rhs="randn", **name[1])
    xr = spla.splu(P["A"]).solve(P["b"])
    errs = {}
    for tol in (1e-3, 1e-9):
        F = OH.factor(P["A"], P["ond"], P["ond_loc"], hss_min=200, hss_leaf=32, dexp=2, swlevel=2, swsize=8, atol=tol, rtol=tol)
        assert F.hss and F.D.hssrank > 0  # the root keeps D as HSS
        errs[tol] = relerr(OH.ldiv(F, P["b"]), xr)
        assert OH.maxrank(F) > 0
    assert errs[1e-9] < 1e-7 and errs[1e-3] < 5e-2 and errs[1e-9] < errs[1e-3]
    # with the switch level at 0 nothing is compressed: the dense oracle's answer
    F0 = OH.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)
    assert relerr(OH.ldiv(F0, P["b"]), xr) < 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("tol", [1e-2, 1e-6])
def test_product_against_oracle_hss(hs, tol):
    P = prepare(hs, (32, 32, 32), rhs="randn", kind="poisson", nmax=512)
    kw = dict(swlevel=3, swsize=8, atol=tol, rtol=tol)
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], hss_min=1024, **kw)
    Fo = OH.factor(P["A"], P["ond"], P["ond_loc"], hss_min=1024, hss_leaf=256, dexp=2, **kw)
    xr = spla.splu(P["A"]).solve(P["b"])
    e_gpu, e_orc = relerr(hs.ldiv(F, P["b"]), xr), relerr(OH.ldiv(Fo, P["b"]), xr)
    print(f"tol={tol:g}: err(product)={e_gpu:.2e} err(oracle)={e_orc:.2e} maxrank {hs.maxrank(F)} / {OH.maxrank(Fo)}")
    assert e_gpu <= max(10 * e_orc, 100 * tol), (e_gpu, e_orc)
    # ranks: pivots of a sketched LU (product) against a pivoted QR (oracle), here compounded over the HSS levels of D
    assert hs.maxrank(F) <= 3 * OH.maxrank(Fo) + 16
