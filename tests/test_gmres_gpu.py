"""Scenario of the reference's script (test/rungmres.jl:32,47-48) with the exact factorization as right
preconditioner: GMRES(30) to reltol 1e-9, everything on the device."""
import numpy as np
import pytest

from helpers import prepare, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["poisson2d_p1_h64_nmax100", "helmholtz2d_p1_h64_nmax100"])
def test_rungmres_scenario_exact_preconditioner(hs, name):
    P = prepare(hs, name, rhs="randn")
    Fa = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)  # "Computing factorization without compression..."
    x1, ch1 = hs.gmres(P["A"], P["b"], Pr=Fa, reltol=1e-9, restart=30, log=True, maxiter=30)
    assert ch1["isconverged"] and ch1["iters"] <= 2, ch1  # exact preconditioner: one Krylov step
    assert ch1["resnorm"][-1] <= 1e-9 * ch1["resnorm"][0]
    assert np.linalg.norm(P["A"] @ x1 - P["b"]) <= 1e-8 * np.linalg.norm(P["b"])
    # without the preconditioner 30 iterations are nowhere near 1e-9 (the point of the reference's plot)
    x0, ch0 = hs.gmres(P["A"], P["b"], reltol=1e-9, restart=30, log=True, maxiter=30)
    assert not ch0["isconverged"] and ch0["iters"] == 30
    assert ch0["resnorm"][-1] > 1e-3 * ch0["resnorm"][0]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(ch0["resnorm"], ch0["resnorm"][1:]))  # GMRES residuals never increase


def test_gmres_matches_direct_solve_small(hs):
    import scipy.sparse.linalg as spla

    P = prepare(hs, (15, 15), kind="helmholtz", nmax=20, rhs="randn")
    x, ch = hs.gmres(P["A"], P["b"], reltol=1e-12, restart=60, log=True, maxiter=240)
    assert ch["isconverged"]
    assert relerr(x, spla.splu(P["A"]).solve(P["b"])) < 1e-8
