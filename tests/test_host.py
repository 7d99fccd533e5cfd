"""CPU tests of the host logic: symbolic layer vs the oracle's restatement, generators, .mat I/O,
and that the C-ABI library loads, exports every declared symbol and fails loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

from conftest import ROOT, have_gpu
from oracle import hs_oracle as O


def same_tree(a, b):
    if (a is None) != (b is None):
        return False
    if a is None:
        return True
    return np.array_equal(a.int, b.int) and np.array_equal(a.bnd, b.bnd) and same_tree(a.left, b.left) and same_tree(a.right, b.right)


@pytest.mark.parametrize("shape,nmax", [((9, 9), 12), ((33, 17), 40), ((6, 6, 6), 30), ((10, 7, 5), 25), ((3, 3), 100)])
def test_symbolic_layer_matches_oracle(hs, shape, nmax):
    A, b, nd = hs.problems.make_problem(shape, nmax=nmax)
    arrays = hs.serialize_elimtree(nd)
    nd2 = hs.parse_elimtree(*arrays)
    assert same_tree(nd, nd2)
    ond = O.parse_elimtree(*arrays)
    assert same_tree(nd, ond)
    nd, nd_loc = hs.symfact(nd)
    ond, ond_loc = O.symfact(ond)
    assert same_tree(nd, ond) and same_tree(nd_loc, ond_loc)
    perm = hs.postorder(nd)
    assert np.array_equal(perm, O.postorder(ond))
    assert np.array_equal(np.sort(perm), np.arange(1, A.shape[0] + 1))  # every DOF eliminated exactly once
    nd = hs.permuted(nd, hs.invperm(perm))
    ond = O.permuted(ond, O.invperm(perm))
    assert same_tree(nd, ond)
    for x in hs.postorder_nodes(nd):  # after the postorder permutation every int is a unit range
        if len(x.int):
            assert isinstance(hs.contigious(x.int), range)
    assert hs.depth(nd) == O.depth(ond)


def test_generator_sizing_model(hs):
    """2D h=1/128 -> 511 nodes, depth 9, fronts <= 258."""
    A, b, nd = hs.problems.make_problem("poisson2d_p1_h128_nmax100")
    nodes = hs.postorder_nodes(nd)
    assert A.shape == (16641, 16641) and len(nodes) == 511 and hs.depth(nd) == 9
    assert max(len(x.int) + len(x.bnd) for x in nodes) == 258
    # disjoint ownership: leaves partition the DOFs
    leaves = [x for x in nodes if hs.isleaf(x)]
    alld = np.sort(np.concatenate([np.concatenate([x.int, x.bnd]) for x in leaves]))
    assert np.array_equal(alld, np.arange(1, A.shape[0] + 1))
    Ah, _, _ = hs.problems.make_problem("helmholtz2d_p1_h64_nmax100")
    assert np.iscomplexobj(Ah.data) and abs(Ah - Ah.T).max() < 1e-14 and abs(Ah - Ah.conj().T).max() > 1e-3  # complex symmetric, non-Hermitian


def test_3d_sizing_model(hs):
    """Poisson 64^3 with leaf <= 4096: top fronts (8192, 0), (3968, 4096) -- the 128^3 model of SURVEY.md 8(d) at half size."""
    nd = hs.problems.grid_nested_dissection((64, 64, 64), 4096)
    nodes = hs.postorder_nodes(nd)
    assert len(nodes) == 127
    assert (len(nd.int), len(nd.bnd)) == (8192, 0)
    assert (len(nd.left.int), len(nd.left.bnd)) == (3968, 4096)


def test_mat_roundtrip(hs, tmp_path):
    A, b, nd = hs.problems.make_problem((9, 7), kind="helmholtz", nmax=10, rhs="randn")
    p = str(tmp_path / "prob.mat")
    hs.problems.write_problem(p, A, b, nd)
    A2, b2, nd2 = hs.problems.read_problem(p)
    assert abs(A - A2).max() == 0 and np.array_equal(b, b2) and same_tree(nd, nd2)


def test_parse_elimtree_validation(hs):
    with pytest.raises(ValueError, match="root"):
        hs.parse_elimtree([-1, -1], [-1, -1], [-1, -1], [1, 1], np.ones((1, 2)), [0, 0], np.ones((1, 2)))
    with pytest.raises(ValueError, match="DimensionMismatch"):
        hs.parse_elimtree([-1], [-1, -1], [-1], [1], np.ones((1, 1)), [0], np.ones((1, 1)))


def test_options_mirror(hs):
    o = hs.SolverOptions()
    assert (o.swlevel, o.swsize, o.atol, o.rtol, o.c_tol, o.leafsize, o.kest, o.stepsize, o.verbose) == (5, 1, 1e-6, 1e-6, 0.5, 32, -1, 10, False)
    o2 = o.copy(swlevel=-2, atol=1e-2)
    assert (o2.swlevel, o2.atol, o.swlevel) == (-2, 1e-2, 5)
    for bad in (dict(swsize=0), dict(atol=-1.0), dict(c_tol=0.0), dict(leafsize=0)):
        with pytest.raises(ValueError):
            hs.chkopts(hs.SolverOptions(**bad))
    with pytest.raises(TypeError):
        hs.SolverOptions(bogus=1)


def test_library_exports_every_declared_symbol(hs):
    lib = hs._lib.lib()
    declared = set()
    for hdr in ("hs_solver.h", "hs_kernels.h", "hs_symbolic.h", "hs_hss.h"):
        txt = open(os.path.join(ROOT, "include", hdr)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        declared |= set(re.findall(r"\b(hsk?_[a-z0-9_]+)\s*\(", txt))
    assert declared == set(hs._lib.EXPORTS), declared ^ set(hs._lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    # the defaults cross the ABI intact (no GPU needed)
    o = hs._lib.hs_options()
    lib.hs_options_default(C.byref(o))
    assert (o.swlevel, o.swsize, o.atol, o.rtol, o.c_tol, o.leafsize, o.kest, o.stepsize, o.verbose) == (5, 1, 1e-6, 1e-6, 0.5, 32, -1, 10, 0)


@pytest.mark.skipif(have_gpu(), reason="checks the no-device failure mode")
def test_product_path_fails_loudly_without_gpu(hs):
    A, b, nd = hs.problems.make_problem((9, 9), nmax=12)
    nd, nd_loc = hs.symfact(nd)
    with pytest.raises(hs.DeviceError, match="no CPU fallback"):
        hs.factor(A, nd, nd_loc, swlevel=0)


def test_product_never_imports_oracle():
    """The product path must not import, link or execute anything under oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, "hierarchicalsolvers.jl_amd")
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+\s*oracle)|oracle/|hs_oracle", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not pat.search(txt), f


@pytest.mark.parametrize("name,kw", [("poisson2d_p1_h64_nmax100", {}), ((12, 10, 9), dict(kind="helmholtz", nmax=40)), ((33, 17), dict(kind="poisson", nmax=25))])
def test_native_symbolic_layer_matches_the_python_mirror(name, kw):
    """C++ parse_elimtree -> symfact! -> postorder -> permuted! (include/hs_symbolic.h) == the Python mirror == the oracle."""
    import hsamd

    hs = hsamd.load()
    A, b, nd = hs.problems.make_problem(name, **kw)
    arrays = hs.serialize_elimtree(nd)
    tree, perm = hs.native_symbolic(*arrays)
    nd2, nd_loc = hs.symfact(hs.parse_elimtree(*arrays))
    perm_py = hs.postorder(nd2)
    nd2 = hs.permuted(nd2, hs.invperm(perm_py))
    flat = hs.flatten_tree(nd2, nd_loc)
    assert np.array_equal(perm, perm_py)
    assert tree["nnodes"] == flat["nnodes"]
    for k in ("left", "right", "int_ptr", "int_idx", "bnd_ptr", "bnd_idx", "iloc_ptr", "iloc_idx", "bloc_ptr", "bloc_idx"):
        assert np.array_equal(tree[k], flat[k]), k
    from oracle import hs_oracle as O

    o = O.parse_elimtree(*arrays)
    o, _ = O.symfact(o)
    assert np.array_equal(O.postorder(o), perm)
    # the flat tree is accepted wherever (nd, nd_loc) is: host-only plan
    h = hs.dist.plan_only(A[perm - 1][:, perm - 1].tocsc(), tree, None)
    assert hs._lib.lib().hs_nlevels(h) == hs.depth(nd2)
    hs._lib.lib().hs_free(h)


def test_native_symbolic_layer_errors():
    import hsamd

    hs = hsamd.load()
    A, b, nd = hs.problems.make_problem((9, 9), kind="poisson", nmax=10)
    f, ls, rs, ni, inter, nb, bound = [np.array(a, copy=True) for a in hs.serialize_elimtree(nd)]
    f2 = f.copy()
    f2[0] = -1  # two roots
    with pytest.raises(ValueError):
        hs.native_symbolic(f2, ls, rs, ni, inter, nb, bound)
    # a branch DOF that no child carries: silently dropped by the reference (nesteddissection.jl:63), an error here
    root = int(np.nonzero(f == -1)[0][0])
    inter2 = np.vstack([inter, np.zeros((1, inter.shape[1]), np.int64)])
    ni2 = ni.copy()
    leaf = int(np.nonzero(ls == -1)[0][0])
    inter2[ni2[root], root] = inter[0, leaf]  # an interior DOF of a leaf claimed by the root as well
    ni2[root] += 1
    with pytest.raises(hs.DimensionMismatch):
        hs.native_symbolic(f, ls, rs, ni2, inter2, nb, bound)


def test_hss_block_order_is_a_compact_bisection(hs):
    """`hsk_bisect_perm` (host only): the order of an HSS interior block is a permutation whose index ranges are compact patches.
    Two coupled layers of a 64 x 64 grid, handed over in strip order as the elimination tree does."""
    N = 64
    A = hs.problems.grid_matrix((N, N, 2), "poisson")  # x fastest, then y, then the layer
    n = A.shape[0]
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    ids = np.arange(1, n + 1, dtype=np.int64)
    perm = np.zeros(n, dtype=np.int64)
    L = hs._lib.lib()
    hs._lib.check(L.hsk_bisect_perm(n, colptr.ctypes.data_as(hs._lib.p_i64), rowval.ctypes.data_as(hs._lib.p_i64), n,
                                    ids.ctypes.data_as(hs._lib.p_i64), perm.ctypes.data_as(hs._lib.p_i64)))
    assert sorted(perm.tolist()) == list(range(n))
    pos = np.empty(n, dtype=np.int64)
    pos[perm] = np.arange(n)
    G = sp.csr_matrix(A)

    def mean_boundary(order_pos, leaf):
        cnt = []
        for lo in range(0, n, leaf):
            v = np.flatnonzero((order_pos >= lo) & (order_pos < lo + leaf))
            nb = G[v].indices
            out = (order_pos[nb] // leaf) != lo // leaf
            rows = np.repeat(np.arange(len(v)), np.diff(G[v].indptr))
            cnt.append(len(np.unique(rows[out])))
        return float(np.mean(cnt))

    # leaves of 256 positions: in the handed-down order they are 4 grid lines (every DOF has a neighbour outside), bisected they are patches
    assert mean_boundary(np.arange(n), 256) > 250
    assert mean_boundary(pos, 256) < 110
    assert L.hsk_bisect_perm(n, None, None, n, ids.ctypes.data_as(hs._lib.p_i64), perm.ctypes.data_as(hs._lib.p_i64)) == hs._lib.HS_ERR_ARGUMENT


def test_plan_rejects_fronts_of_a_level_that_share_a_dof(hs):
    """The fronts of one tree level are assembled by the same grouped launches: a DOF in the [int; bnd] of two same-level
    fronts (a vertex-separator tree) or listed twice in one front must be refused, not raced on (host-only plan)."""
    A, b, nd = hs.problems.make_problem((9, 9), kind="poisson", nmax=10)
    tree, perm = hs.native_symbolic(*hs.serialize_elimtree(nd))
    Ap = A[perm - 1][:, perm - 1].tocsc()
    h = hs.dist.plan_only(Ap, tree, None)  # the honest tree is accepted
    hs._lib.lib().hs_free(h)
    leaves = [i for i in range(tree["nnodes"]) if tree["left"][i] < 0]
    a, c = leaves[0], leaves[1]

    def with_extra_bnd(node, dof):
        """`node` lists one more boundary DOF that its parent never maps (symfact! would drop it): every per-node check passes."""
        t = {k: np.array(v, copy=True) if isinstance(v, np.ndarray) else v for k, v in tree.items()}
        at = t["bnd_ptr"][node + 1]
        t["bnd_idx"] = np.insert(t["bnd_idx"], at, dof)
        t["bnd_ptr"][node + 1 :] += 1
        return t

    ya = tree["bnd_idx"][tree["bnd_ptr"][a]]
    with pytest.raises(hs.DimensionMismatch, match="same tree level"):
        hs.dist.plan_only(Ap, with_extra_bnd(c, ya), None)  # a boundary DOF of leaf a also in leaf c's front
    yc = tree["bnd_idx"][tree["bnd_ptr"][c]]
    with pytest.raises(hs.DimensionMismatch, match="listed twice"):
        hs.dist.plan_only(Ap, with_extra_bnd(c, yc), None)


def test_size_model_restates_the_tree(hs):
    """tools/size_model.py derives every front's (ni, nb) from the boxes alone (no index lists: 256^3 in seconds) -- the same numbers as the real
    geometric nested dissection, in post-order; and its exact part (dense factor bytes) equals hs_plan's."""
    import importlib.util
    import os

    spec = importlib.util.spec_from_file_location("size_model", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "size_model.py"))
    sm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sm)
    for shape, nmax in (((12, 12, 12), 64), ((16, 16, 16), 128), ((9, 14, 11), 40)):
        nd = hs.problems.grid_nested_dissection(shape, nmax)
        real = [(len(x.int), len(x.bnd)) for x in hs.postorder_nodes(nd)]
        model = [(x["ni"], x["nb"]) for x in sm.tree_sizes(shape, nmax)]
        assert real == model, (shape, nmax)
    # the dense bytes of the model = the library's own plan (one rank, exact)
    from helpers import prepare

    P = prepare(hs, (16, 16, 16), kind="poisson", nmax=128)
    h = hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=1)
    try:
        import ctypes as C

        st = hs._lib.hs_stats()
        hs._lib.check(hs._lib.lib().hs_get_stats(h, C.byref(st)))
        plan_bytes = st.bytes_factors
    finally:
        hs._lib.lib().hs_free(h)
    nodes = sm.tree_sizes(16, 128)
    model_bytes = sum(sm.dense_front_elems(x["ni"], x["nb"]) for x in nodes) * 8
    assert abs(model_bytes - plan_bytes) <= 0.02 * plan_bytes, (model_bytes, plan_bytes)  # (the plan pads leading dimensions and blocks of 32)
