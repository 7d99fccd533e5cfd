"""Nested dissection from the graph alone (`problems.graph_nested_dissection`): the tree must be one
the reference's symbolic layer accepts, and the factorization built on it exact (oracle vs SuperLU on CPU; the GPU test runs
the product on the same tree)."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from oracle import hs_oracle as O


def unstructured_problem(npts=1500, seed=0, complex_=False):
    """P1 stiffness-like matrix of a Delaunay triangulation of random points (graph Laplacian + mass-like diagonal)."""
    from scipy.spatial import Delaunay

    rng = np.random.default_rng(seed)
    pts = rng.random((npts, 2))
    tri = Delaunay(pts).simplices
    i = np.concatenate([tri[:, 0], tri[:, 1], tri[:, 2]])
    j = np.concatenate([tri[:, 1], tri[:, 2], tri[:, 0]])
    W = sp.coo_matrix((np.ones(len(i)), (i, j)), shape=(npts, npts)).tocsr()
    W = ((W + W.T) > 0).astype(float)
    A = sp.diags(np.asarray(W.sum(axis=1)).ravel() + 0.1) - W
    if complex_:
        A = A.astype(np.complex128) - (0.3 + 0.2j) * sp.identity(npts)
    return sp.csc_matrix(A)


def leaves(nd):
    return [nd] if nd.left is None else leaves(nd.left) + leaves(nd.right)


def check_tree(hs, A, nd, nmax):
    n = A.shape[0]
    owned = np.concatenate([np.concatenate([x.int, x.bnd]) for x in leaves(nd)])
    assert sorted(owned.tolist()) == list(range(1, n + 1))  # disjoint ownership: every DOF in exactly one leaf
    assert all(len(x.int) + len(x.bnd) <= nmax for x in leaves(nd))
    assert len(nd.bnd) == 0
    G = sp.csr_matrix(abs(A) + abs(A).T)

    def walk(x):
        if x.left is None:
            box = np.concatenate([x.int, x.bnd]) - 1
        else:
            box = np.concatenate([walk(x.left), walk(x.right)])
            # a parent eliminates exactly the children's boundary DOFs that are not on its own boundary
            cb = np.union1d(x.left.bnd, x.right.bnd)
            assert np.array_equal(np.sort(x.int), np.setdiff1d(cb, x.bnd))
        inbox = np.zeros(n, dtype=bool)
        inbox[box] = True
        out = np.asarray((G[box] @ (~inbox).astype(float))).ravel() > 0
        assert np.array_equal(np.sort(x.bnd - 1), np.sort(box[out]))  # bnd = DOFs with a neighbour outside the box
        return box

    walk(nd)


@pytest.mark.parametrize("complex_", [False, True])
def test_graph_nd_tree_and_exact_factorization_on_cpu(hs, complex_):
    A = unstructured_problem(1200, complex_=complex_)
    nd = hs.problems.graph_nested_dissection(A, nmax=60)
    check_tree(hs, A, nd, 60)
    arrays = hs.serialize_elimtree(nd)
    o = O.parse_elimtree(*arrays)
    o, o_loc = O.symfact(o)
    perm = O.postorder(o)
    Ap = A[perm - 1][:, perm - 1].tocsc()
    o = O.permuted(o, O.invperm(perm))
    rng = np.random.default_rng(1)
    b = rng.standard_normal(A.shape[0]) + (1j * rng.standard_normal(A.shape[0]) if complex_ else 0)
    x = O.ldiv(O.factor(Ap, o, o_loc, swlevel=0), b)
    xr = spla.splu(Ap).solve(b)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < 1e-10


def test_graph_nd_on_a_grid_matrix(hs):
    A = hs.problems.grid_matrix((12, 11, 10), "poisson")
    nd = hs.problems.graph_nested_dissection(A, nmax=80)
    check_tree(hs, A, nd, 80)
    nd2, nd_loc = hs.symfact(nd)  # the product's symbolic layer accepts it too
    assert len(hs.postorder(nd2)) == A.shape[0]


@pytest.mark.gpu
@pytest.mark.parametrize("complex_", [False, True])
def test_graph_nd_factorization_on_gpu(hs, complex_):
    A = unstructured_problem(3000, seed=2, complex_=complex_)
    nd = hs.problems.graph_nested_dissection(A, nmax=100)
    nd, nd_loc = hs.symfact(nd)
    perm = hs.postorder(nd)
    Ap = A[perm - 1][:, perm - 1].tocsc()
    nd = hs.permuted(nd, hs.invperm(perm))
    rng = np.random.default_rng(3)
    b = rng.standard_normal(A.shape[0]) + (1j * rng.standard_normal(A.shape[0]) if complex_ else 0)
    F = hs.factor(Ap, nd, nd_loc, swlevel=0)
    x = hs.ldiv(F, b)
    xr = spla.splu(Ap).solve(b)
    assert np.linalg.norm(x - xr) / np.linalg.norm(xr) < 1e-10
    # compressed fronts on the same tree: a preconditioner-quality solve
    Fc = hs.factor(Ap, nd, nd_loc, swlevel=3, swsize=8, atol=1e-6, rtol=1e-6)
    assert np.linalg.norm(hs.ldiv(Fc, b) - xr) / np.linalg.norm(xr) < 1e-3


@pytest.mark.parametrize("complex_", [False, True])
def test_native_graph_nd_matches_the_python_mirror(hs, complex_):
    """`hs_symbolic_from_graph` (C++: graph -> tree -> symfact! -> postorder -> permuted!) against the Python pipeline on the same
    matrix: same elimination order, same flat tree; and the oracle's factorization on it is exact."""
    A = unstructured_problem(900, seed=4, complex_=complex_)
    tree, perm = hs.native_graph_symbolic(A, nmax=50)
    nd = hs.problems.graph_nested_dissection(A, nmax=50)
    nd2, nd_loc = hs.symfact(nd)
    perm_py = hs.postorder(nd2)
    assert np.array_equal(perm, perm_py)
    flat = hs.flatten_tree(hs.permuted(nd2, hs.invperm(perm_py)), nd_loc)
    assert tree["nnodes"] == flat["nnodes"]
    for k in ("left", "right", "int_ptr", "int_idx", "bnd_ptr", "bnd_idx", "iloc_ptr", "iloc_idx", "bloc_ptr", "bloc_idx"):
        assert np.array_equal(np.asarray(tree[k]), np.asarray(flat[k])), k
    assert sorted(perm.tolist()) == list(range(1, A.shape[0] + 1))
    # a grid matrix too, and argument errors
    G = hs.problems.grid_matrix((9, 8, 7), "poisson")
    t2, p2 = hs.native_graph_symbolic(G, nmax=40)
    assert sorted(p2.tolist()) == list(range(1, G.shape[0] + 1)) and t2["nnodes"] >= 15
    with pytest.raises(ValueError):
        hs.native_graph_symbolic(G, nmax=0)


@pytest.mark.gpu
def test_native_graph_nd_factorization_on_gpu(hs):
    A = unstructured_problem(2500, seed=5)
    tree, perm = hs.native_graph_symbolic(A, nmax=80)
    Ap = A[perm - 1][:, perm - 1].tocsc()
    b = np.random.default_rng(6).standard_normal(A.shape[0])
    F = hs.factor(Ap, tree, None, swlevel=0)
    x = hs.ldiv(F, b)
    assert np.linalg.norm(x - spla.splu(Ap).solve(b)) / np.linalg.norm(b) < 1e-10
