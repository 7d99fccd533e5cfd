"""CPU tests of the oracle (no GPU): T0 exactness vs SuperLU, algebraic identities, golden fixtures."""
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from helpers import prepare, relerr
from oracle import hs_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_fixture(path):
    z = np.load(path, allow_pickle=False)
    n = int(z["nnodes"])
    A = sp.csc_matrix((z["data"], z["indices"], z["indptr"]), shape=tuple(z["shape"]))
    tree = (z["fathers"], z["lsons"], z["rsons"], z["ninter"], z["inter"], z["nbound"], z["bound"])
    return dict(A=A, b=z["b"], x=z["x"], tree=tree, D=[z[f"D{k}"] for k in range(n)], L=[z[f"L{k}"] for k in range(n)],
                R=[z[f"R{k}"] for k in range(n)], S=[z[f"S{k}"] for k in range(n)])


def onodes(F):
    out = []

    def walk(f):
        if f.left is not None:
            walk(f.left)
        if f.right is not None:
            walk(f.right)
        out.append(f)

    walk(F)
    return out


@pytest.mark.parametrize("name", ["poisson2d_p1_h64_nmax100", "helmholtz2d_p1_h64_nmax100"])
def test_t0_exact_vs_splu(hs, name):
    P = prepare(hs, name)
    F = O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)
    x = O.ldiv(F, P["b"])
    assert relerr(x, spla.splu(P["A"]).solve(P["b"])) < 1e-11
    assert relerr(x, np.ones(len(x))) < 1e-10  # b = A*1
    assert O.maxrank(F) == 0
    # sizing model of SURVEY.md section 8(d): 127 nodes, depth 7, largest front 130, 2.8e7 flops
    nodes = O.postorder_nodes(P["ond"])
    assert len(nodes) == 127 and O.depth(P["ond"]) == 7
    assert max(len(x.int) + len(x.bnd) for x in nodes) == 130
    assert 2.7e7 < O.tree_flops(P["ond"]) < 2.9e7


def test_node_identities(hs):
    """L*D = A_bi, D*R = A_ib, S = A_bb - A_bi*R on every assembled front."""
    P = prepare(hs, (20, 17), kind="helmholtz", nmax=25)
    F = O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)
    A = P["A"]
    # global Schur complement identity: eliminating everything below a node leaves S = Schur(A) on its bnd
    for f in onodes(F):
        if len(f.bnd) == 0:
            continue
        sub = np.concatenate([g.int for g in onodes(f)])
        Aee = A[sub - 1][:, sub - 1].toarray()
        Abe = A[f.bnd - 1][:, sub - 1].toarray()
        Aeb = A[sub - 1][:, f.bnd - 1].toarray()
        Sref = A[f.bnd - 1][:, f.bnd - 1].toarray() - Abe @ np.linalg.solve(Aee, Aeb)
        perm = np.concatenate([f.int_loc, f.bnd_loc]) - 1
        assert relerr(O._dense(f.S), Sref[np.ix_(perm, perm)]) < 1e-10


def test_block_identity():
    """blockldiv!(blockfactor(M), M*X) == X and the right-hand twin."""
    rng = np.random.default_rng(1)
    n1, n2, k = 13, 9, 4
    M = rng.standard_normal((n1 + n2, n1 + n2)) + 5 * np.eye(n1 + n2)
    B = O.BlockMatrix(M[:n1, :n1], M[:n1, n1:], M[n1:, :n1], M[n1:, n1:])
    Fb = O.blockfactor(B)
    X = rng.standard_normal((n1 + n2, k))
    assert relerr(O.blockldiv_inplace(Fb, M @ X), X) < 1e-12
    Y = rng.standard_normal((k, n1 + n2))
    assert relerr(O.blockrdiv_inplace(Y @ M, Fb), Y) < 1e-12
    Bx = O.BlockMatrix(X[:n1, :2], X[:n1, 2:], X[n1:, :2], X[n1:, 2:])
    MB = O.BlockMatrix(*(lambda Z: (Z[:n1, :2], Z[:n1, 2:], Z[n1:, :2], Z[n1:, 2:]))(M @ X))
    assert relerr(O.blockldiv(Fb, MB).dense(), Bx.dense()) < 1e-12
    with pytest.raises(ValueError):
        O.BlockMatrix(M[:n1, :n1], M[: n1 - 1, n1:], M[n1:, :n1], M[n1:, n1:])


def test_options():
    o = O.SolverOptions()
    assert (o.swlevel, o.swsize, o.atol, o.rtol, o.c_tol, o.leafsize, o.kest, o.stepsize, o.verbose) == (5, 1, 1e-6, 1e-6, 0.5, 32, -1, 10, False)
    for bad in (dict(swsize=0), dict(atol=-1.0), dict(rtol=-1e-3), dict(c_tol=0.0), dict(c_tol=1.5), dict(leafsize=0)):
        with pytest.raises(ValueError):
            O.chkopts(O.SolverOptions(**bad))
    with pytest.raises(TypeError):
        O.SolverOptions(nosuchfield=1)


def test_parse_elimtree_errors():
    with pytest.raises(ValueError):
        O.parse_elimtree([-1, -1], [-1, -1], [-1, -1], [1, 1], np.ones((1, 2)), [0, 0], np.ones((1, 2)))  # two roots
    with pytest.raises(ValueError):
        O.parse_elimtree([-1], [-1, -1], [-1], [1], np.ones((1, 1)), [0], np.ones((1, 1)))  # inconsistent lengths


def test_one_child_node_raises(hs):
    P = prepare(hs, (9, 9), nmax=12)
    P["ond"].right = None
    with pytest.raises(RuntimeError, match="binary tree"):
        O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_oracle_reproduces_golden(path):
    """The committed fixtures are what the oracle computes today (guards against silent oracle drift)."""
    fx = load_fixture(path)
    o = O.parse_elimtree(*fx["tree"])
    o, o_loc = O.symfact(o)
    perm = O.postorder(o)
    Ap = fx["A"][perm - 1][:, perm - 1].tocsc()
    o = O.permuted(o, O.invperm(perm))
    F = O.factor(Ap, o, o_loc, swlevel=0)
    xp = O.ldiv(F, fx["b"][perm - 1])
    assert relerr(xp, fx["x"][perm - 1]) < 1e-12
    assert relerr(fx["A"] @ fx["x"], fx["b"]) < 1e-12
    for k, f in enumerate(onodes(F)):
        assert relerr(O._dense(f.L), fx["L"][k]) < 1e-12 or fx["L"][k].size == 0
        assert relerr(O._dense(f.R), fx["R"][k]) < 1e-12 or fx["R"][k].size == 0
