"""CPU tests of the C restatement of the reference's dense path (oracle/hs_oracle_c.c, SURVEY.md 8(d): the CPU baseline) -- no GPU.

Pinned node by node against the NumPy restatement (oracle/hs_oracle.py: |S_i|_F of every front, the solution), against SuperLU, and against the
committed golden fixtures; with both kernel tables (SciPy's BLAS / LAPACK and the plain loops of the C file)."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from helpers import prepare, relerr
from oracle import hs_oracle as O
from oracle import hs_oracle_c as OC
from test_oracle import GOLD, load_fixture, onodes


@pytest.mark.parametrize("kernels", ["blas", "loops"])
@pytest.mark.parametrize("name,kw", [("poisson2d_p1_h64_nmax100", {}), ("helmholtz2d_p1_h64_nmax100", {}), ((9, 8, 7), dict(kind="poisson", nmax=30)),
                                     ((8, 7, 9), dict(kind="helmholtz", nmax=40))])
def test_c_restatement_against_the_numpy_one_and_splu(hs, name, kw, kernels):
    P = prepare(hs, name, **kw)
    F = O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)
    x_py = O.ldiv(F, P["b"])
    x_c, info = OC.factor_solve(P["A"], P["ond"], P["ond_loc"], P["b"], kernels=kernels)
    assert relerr(x_c, x_py) < 1e-11
    assert relerr(x_c, spla.splu(P["A"]).solve(P["b"])) < 1e-10
    nodes = onodes(F)
    s_py = np.array([np.linalg.norm(O._dense(f.S)) for f in nodes])
    assert len(s_py) == len(info["snorm"])
    assert np.allclose(info["snorm"], s_py, rtol=1e-10, atol=1e-12 * (s_py.max() + 1))
    # the executed work is the reference's, not the minimal count: 13 LUs per dense branch (blockfactor 1, blockrdiv 6, blockldiv 6), 2 per leaf;
    # every ldiv! factors again: 3 per branch, 1 per leaf
    nb = sum(f.left is not None for f in nodes)
    nl = len(nodes) - nb
    assert 6 * nb < info["factor_getrf"] <= 13 * nb + 2 * nl  # (a solve with no right-hand side -- an empty block -- skips its LU)
    assert info["ldiv_getrf"] <= 3 * nb + nl
    mult = 4.0 if np.iscomplexobj(x_c) else 1.0
    assert info["factor_flops"] > 1.2 * mult * O.tree_flops(P["ond"])  # (2-3x on trees with large fronts)
    assert info["singular"] == 0


def test_several_right_hand_sides_and_a_lone_leaf(hs):
    P = prepare(hs, (12, 11), kind="poisson", nmax=20)
    rng = np.random.default_rng(1)
    B = rng.standard_normal((P["A"].shape[0], 3))
    X, _ = OC.factor_solve(P["A"], P["ond"], P["ond_loc"], B)
    assert relerr(X, spla.splu(P["A"]).solve(B)) < 1e-11
    P1 = prepare(hs, (5, 4), kind="poisson", nmax=100)  # the tree is one leaf
    assert O.isleaf(P1["ond"])
    x, info = OC.factor_solve(P1["A"], P1["ond"], P1["ond_loc"], P1["b"])
    assert relerr(x, spla.splu(P1["A"]).solve(P1["b"])) < 1e-12 and len(info["snorm"]) == 1


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))), ids=os.path.basename)
def test_golden_fixtures(hs, path):
    z = np.load(path, allow_pickle=False)
    if "fathers" not in z.files:
        pytest.skip("not a dense-path fixture")
    fx = load_fixture(path)
    nd = O.parse_elimtree(*fx["tree"])
    nd, nd_loc = O.symfact(nd)
    x, info = OC.factor_solve(fx["A"], nd, nd_loc, fx["b"])
    assert relerr(x, fx["x"]) < 1e-10
    assert np.allclose(info["snorm"], [np.linalg.norm(s) for s in fx["S"]], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("cplx", [False, True])
def test_plain_loops_against_the_blas_table(cplx):
    rng = np.random.default_rng(3)
    n, m = 150, 37
    A = rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0) + 5 * np.eye(n)
    B = rng.standard_normal((n, m)) + (1j * rng.standard_normal((n, m)) if cplx else 0)
    X1, Xr1 = OC.selftest_solve(A, B, "blas")
    X2, Xr2 = OC.selftest_solve(A, B, "loops")
    assert relerr(X1, np.linalg.solve(A, B)) < 1e-12 and relerr(X2, X1) < 1e-12
    assert relerr(Xr1, np.linalg.solve(A.T, B).T) < 1e-12 and relerr(Xr2, Xr1) < 1e-12


def test_one_child_is_refused(hs):
    P = prepare(hs, (6, 6), kind="poisson", nmax=10)
    nd = P["ond"]
    keep = nd.right
    nd.right = None
    try:
        with pytest.raises(RuntimeError, match="binary tree"):  # factorization.jl:25
            OC.factor_solve(P["A"], nd, P["ond_loc"], P["b"])
    finally:
        nd.right = keep
