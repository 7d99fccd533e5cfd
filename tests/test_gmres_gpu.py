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


# ---- the same solver behind the C ABI (hs_gmres_{d,z}: hand-written SpMV / Gram-Schmidt / Givens kernels in libhs_solver.so) ----------
SCENARIOS = [
    ("poisson2d_p1_h64_nmax100", dict(swlevel=-2, swsize=8, atol=1e-2, rtol=1e-2)),
    ("helmholtz2d_p1_h64_nmax100", dict(swlevel=-2, swsize=8, atol=1e-2, rtol=1e-2)),
    (((24, 24, 24), dict(kind="poisson", nmax=512)), dict(swlevel=2, swsize=8, atol=1e-2, rtol=1e-2)),
    (((20, 20, 20), dict(kind="helmholtz", nmax=512)), dict(swlevel=2, swsize=8, atol=1e-2, rtol=1e-2)),
]


@pytest.mark.parametrize("name,copts", SCENARIOS)
def test_native_gmres_matches_the_mirror(hs, name, copts):
    """`hs_gmres_*` (the package's `gmres`) vs the torch mirror (tests/gmres_mirror.py) on the scenario of test/rungmres.jl:32-48: exact and compressed preconditioners and no
    preconditioner at all -- same iteration counts, same residual histories to rounding, same solution."""
    from gmres_mirror import gmres_mirror
    from hierarchicalsolvers_jl_amd.gmres import gmres_native

    P = prepare(hs, name, rhs="randn") if isinstance(name, str) else prepare(hs, name[0], rhs="randn", **name[1])
    for label, F in (("exact", hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)), ("compressed", hs.factor(P["A"], P["nd"], P["nd_loc"], **copts)), ("none", None)):
        kw = dict(Pr=F, reltol=1e-9, restart=30, log=True, maxiter=30 if F is not None else 45)
        x1, c1 = gmres_mirror(P["A"], P["b"], **kw)
        x2, c2 = gmres_native(P["A"], P["b"], **kw)
        print(name if isinstance(name, str) else name[0], label, "iterations: mirror", c1["iters"], "native", c2["iters"])
        assert c2["iters"] == c1["iters"] and c2["isconverged"] == c1["isconverged"], (label, c1["iters"], c2["iters"])
        h1, h2 = np.array(c1["resnorm"]), np.array(c2["resnorm"])
        assert np.allclose(h1, h2, rtol=1e-6, atol=1e-12 * h1[0]), (label, h1, h2)
        assert relerr(x2, x1) < 1e-6
        if c2["isconverged"]:
            assert np.linalg.norm(P["A"] @ x2 - P["b"]) <= 1e-8 * np.linalg.norm(P["b"])


def test_native_gmres_arguments(hs):
    from hierarchicalsolvers_jl_amd.gmres import gmres_native

    P = prepare(hs, (15, 15), kind="helmholtz", nmax=20, rhs="randn")
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    # an initial guess: same behaviour as the mirror (the residual of the guess is the reference of `reltol`)
    import scipy.sparse.linalg as spla

    x0 = spla.splu(P["A"]).solve(P["b"]) * (1 + 1e-3)
    x, c = gmres_native(P["A"], P["b"], Pr=F, reltol=1e-6, restart=30, maxiter=10, log=True, x0=x0)
    from gmres_mirror import gmres_mirror

    xm, cm = gmres_mirror(P["A"], P["b"], Pr=F, reltol=1e-6, restart=30, maxiter=10, log=True, x0=x0)
    assert c["iters"] == cm["iters"] and c["isconverged"] and relerr(x, xm) < 1e-8
    # b = 0: nothing to do
    x, c = gmres_native(P["A"], 0 * P["b"], Pr=F, reltol=1e-9, restart=30, maxiter=10, log=True)
    assert c["iters"] == 0 and c["isconverged"] and not np.any(x)
    # a real factorization cannot precondition a complex system (MethodError in Julia)
    Pr = prepare(hs, (15, 15), kind="poisson", nmax=20, rhs="randn")
    Fr = hs.factor(Pr["A"], Pr["nd"], Pr["nd_loc"], swlevel=0)
    with pytest.raises(hs.DimensionMismatch):
        gmres_native(P["A"], P["b"], Pr=Fr, reltol=1e-9, restart=30, maxiter=10)
    with pytest.raises(ValueError):
        gmres_native(P["A"], P["b"], Pr=F, restart=1000, maxiter=10)
