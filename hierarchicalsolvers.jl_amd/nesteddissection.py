"""Host-side mirror of the reference's symbolic layer (``src/nesteddissection.jl``).

Same names and argument meaning as the Julia API (``NDNode``, ``symfact!``,
``postorder``, ``permuted!``, ``contigious``, ``parse_elimtree``,
``getinterior``, ``getboundary``); Julia's ``!`` suffix is dropped.  All index
vectors are 1-based ``int64`` exactly as the Julia host holds them -- they are
what crosses the C ABI (``include/hs_solver.h``: ``hs_tree``).

Unlike the reference (``findall(in(...))`` per node, O(|parent|*|child|) with
vector ``in``), membership tests use one shared scratch map so ``symfact`` is
O(sum of index-set lengths); results are identical.
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "NDNode",
    "isleaf",
    "isbranch",
    "depth",
    "symfact",
    "postorder",
    "postorder_nodes",
    "permuted",
    "invperm",
    "contigious",
    "parse_elimtree",
    "serialize_elimtree",
    "getinterior",
    "getboundary",
    "flatten_tree",
    "native_symbolic",
    "native_graph_symbolic",
]


def _ivec(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.int64).reshape(-1))


class NDNode:
    """``NDNode(int, bnd[, left, right])`` (nesteddissection.jl:19-21)."""

    __slots__ = ("int", "bnd", "left", "right")

    def __init__(self, int_, bnd, left=None, right=None):
        self.int = _ivec(int_)
        self.bnd = _ivec(bnd)
        self.left = left
        self.right = right

    @classmethod
    def join(cls, left, right):
        """``NDNode(left, right)``: int = union(left.bnd, right.bnd), bnd = [] (nesteddissection.jl:21)."""
        u, first = np.unique(np.concatenate([left.bnd, right.bnd]), return_index=True)
        return cls(u[np.argsort(first)], [], left, right)


def isleaf(nd):
    return nd.left is None and nd.right is None


def isbranch(nd):
    return nd.left is not None and nd.right is not None


def depth(nd):
    """``HssMatrices.depth`` of a BinaryNode: number of levels (a lone leaf has depth 1).
    Iterative: trees from graded meshes may be deep."""
    best, stack = 0, [(nd, 1)]
    while stack:
        x, d = stack.pop()
        best = max(best, d)
        if x.left is not None:
            stack.append((x.left, d + 1))
        if x.right is not None:
            stack.append((x.right, d + 1))
    return best


def postorder_nodes(nd):
    """Nodes in ``AbstractTrees.PostOrderDFS`` order (left, right, self)."""
    out, stack = [], [(nd, False)]
    while stack:
        x, seen = stack.pop()
        if seen:
            out.append(x)
            continue
        stack.append((x, True))
        if x.right is not None:
            stack.append((x.right, False))
        if x.left is not None:
            stack.append((x.left, False))
    return out


def _maxdof(nd):
    m = 0
    for x in postorder_nodes(nd):
        if len(x.int):
            m = max(m, int(x.int.max()))
        if len(x.bnd):
            m = max(m, int(x.bnd.max()))
    return m


def symfact(nd):
    """``symfact!(nd) -> (nd, nd_loc)`` (nesteddissection.jl:29-69).

    Mutates ``nd`` (parents' ``int``/``bnd`` become ``[left part; right part]``)
    and returns the tree of local index maps: for each child, the 1-based
    positions inside the child's ``bnd`` that land in the parent's ``int`` /
    ``bnd``.  Root: ``nd_loc.int = 1:|bnd|``, ``nd_loc.bnd = []``."""
    mark = np.zeros(_maxdof(nd) + 2, dtype=np.int8)
    loc_of = {}
    for x in postorder_nodes(nd):  # children before parents, exactly the recursion order of _symfact!
        if isleaf(x):
            loc_of[id(x)] = NDNode([], [])
            continue
        parts_i, parts_b, locs = [], [], []
        mark[x.int] = 1
        mark[x.bnd] = 2
        for child in (x.left, x.right):
            if child is None:
                locs.append(None)
                continue
            cl = loc_of.pop(id(child))
            m = mark[child.bnd]
            cl.int = np.nonzero(m == 1)[0].astype(np.int64) + 1  # findall(in(nd.int), child.bnd)
            cl.bnd = np.nonzero(m == 2)[0].astype(np.int64) + 1  # findall(in(nd.bnd), child.bnd)
            parts_i.append(child.bnd[cl.int - 1])
            parts_b.append(child.bnd[cl.bnd - 1])
            locs.append(cl)
        mark[x.int] = 0
        mark[x.bnd] = 0
        x.int = np.concatenate(parts_i) if parts_i else np.zeros(0, np.int64)
        x.bnd = np.concatenate(parts_b) if parts_b else np.zeros(0, np.int64)
        loc_of[id(x)] = NDNode([], [], locs[0], locs[1])
    nd_loc = loc_of[id(nd)]
    nd_loc.int = np.arange(1, len(nd.bnd) + 1, dtype=np.int64)
    nd_loc.bnd = np.zeros(0, dtype=np.int64)
    return nd, nd_loc


def postorder(nd):
    """Elimination permutation: every node's ``int`` in post-order, then the root ``bnd`` (nesteddissection.jl:73-79)."""
    parts = [x.int for x in postorder_nodes(nd)]
    parts.append(nd.bnd)
    return np.concatenate(parts)


def invperm(p):
    p = np.asarray(p, dtype=np.int64)
    ip = np.empty_like(p)
    ip[p - 1] = np.arange(1, len(p) + 1, dtype=np.int64)
    return ip


def permuted(nd, perm):
    """``permuted!(nd, perm)``: ``int <- perm[int]``, ``bnd <- perm[bnd]`` on every node (nesteddissection.jl:82-88)."""
    perm = np.asarray(perm, dtype=np.int64)
    for x in postorder_nodes(nd):
        x.int = perm[x.int - 1]
        x.bnd = perm[x.bnd - 1]
    return nd


def contigious(idx):
    """``contigious(idx)`` (nesteddissection.jl:91, reference spelling): a ``range`` when idx is a unit range."""
    idx = np.asarray(idx, dtype=np.int64)
    if len(idx) and idx[-1] - idx[0] + 1 == len(idx) and np.array_equal(np.arange(idx[0], idx[-1] + 1), idx):
        return range(int(idx[0]), int(idx[-1]) + 1)
    return idx


def getinterior(nd):
    """Second (overriding) method in the reference: ``1:nd.int[end]`` (nesteddissection.jl:100)."""
    return range(1, int(nd.int[-1]) + 1)


def getboundary(nd):
    return nd.bnd


def parse_elimtree(fathers, lsons, rsons, ninter, inter, nbound, bound):
    """De-serialise the 7-array elimination-tree format (nesteddissection.jl:105-148;
    field layout ``util/read_problem.jl:14-20``).  1-based node ids, ``-1`` = none;
    column ``i`` of ``inter``/``bound`` holds the first ``ninter[i]``/``nbound[i]`` DOF ids."""
    fathers, lsons, rsons = _ivec(fathers), _ivec(lsons), _ivec(rsons)
    ninter, nbound = _ivec(ninter), _ivec(nbound)
    inter = np.asarray(inter, dtype=np.int64)
    bound = np.asarray(bound, dtype=np.int64)
    inter = inter.reshape(inter.shape[0], -1) if inter.ndim == 2 else inter.reshape(-1, 1)
    bound = bound.reshape(bound.shape[0], -1) if bound.ndim == 2 else bound.reshape(-1, 1)
    n = len(fathers)
    if not (n == len(lsons) == len(rsons) == len(ninter) == len(nbound) == inter.shape[1] == bound.shape[1]):
        raise ValueError("DimensionMismatch: dimensions inconsistent among inputs")
    roots = np.nonzero(fathers == -1)[0]
    if len(roots) != 1:
        raise ValueError("ArgumentError: found either less than or more than one root.")
    built = {}
    stack = [(int(roots[0]) + 1, False)]
    while stack:
        i, seen = stack.pop()
        ls, rs = int(lsons[i - 1]), int(rsons[i - 1])
        if not seen:
            stack.append((i, True))
            if rs != -1:
                stack.append((rs, False))
            if ls != -1:
                stack.append((ls, False))
            continue
        built[i] = NDNode(
            inter[: ninter[i - 1], i - 1],
            bound[: nbound[i - 1], i - 1],
            built.pop(ls) if ls != -1 else None,
            built.pop(rs) if rs != -1 else None,
        )
    return built[int(roots[0]) + 1]


def serialize_elimtree(nd):
    """Inverse of :func:`parse_elimtree`: the 7 arrays ``util/read_problem.jl`` reads from a ``.mat`` file."""
    nodes = postorder_nodes(nd)
    ids = {id(x): k + 1 for k, x in enumerate(nodes)}
    n = len(nodes)
    fathers = np.full(n, -1, np.int64)
    lsons = np.full(n, -1, np.int64)
    rsons = np.full(n, -1, np.int64)
    ninter = np.array([len(x.int) for x in nodes], np.int64)
    nbound = np.array([len(x.bnd) for x in nodes], np.int64)
    inter = np.zeros((max(1, int(ninter.max())), n), np.int64)
    bound = np.zeros((max(1, int(nbound.max())), n), np.int64)
    for k, x in enumerate(nodes):
        inter[: ninter[k], k] = x.int
        bound[: nbound[k], k] = x.bnd
        if x.left is not None:
            lsons[k] = ids[id(x.left)]
            fathers[ids[id(x.left)] - 1] = k + 1
        if x.right is not None:
            rsons[k] = ids[id(x.right)]
            fathers[ids[id(x.right)] - 1] = k + 1
    return fathers, lsons, rsons, ninter, inter, nbound, bound


def flatten_tree(nd, nd_loc):
    """Flat post-ordered arrays for ``hs_tree`` (include/hs_solver.h).

    Node ids are 0-based post-order positions (root last, ``-1`` = no child);
    DOF ids and local positions stay 1-based as Julia holds them.  A dict that already is such a flat tree
    (:func:`native_symbolic`) is passed through (``nd_loc`` is then ignored)."""
    if isinstance(nd, dict) and "nnodes" in nd:
        return nd
    nodes, locs = [], []
    stack = [(nd, nd_loc, False)]
    while stack:
        x, xl, seen = stack.pop()
        if seen:
            nodes.append(x)
            locs.append(xl)
            continue
        stack.append((x, xl, True))
        if x.right is not None:
            stack.append((x.right, xl.right, False))
        if x.left is not None:
            stack.append((x.left, xl.left, False))
    ids = {id(x): k for k, x in enumerate(nodes)}
    n = len(nodes)
    left = np.array([ids[id(x.left)] if x.left is not None else -1 for x in nodes], np.int64)
    right = np.array([ids[id(x.right)] if x.right is not None else -1 for x in nodes], np.int64)

    def pack(vs):
        ptr = np.zeros(n + 1, np.int64)
        ptr[1:] = np.cumsum([len(v) for v in vs])
        idx = np.concatenate(vs).astype(np.int64) if n else np.zeros(0, np.int64)
        return ptr, np.ascontiguousarray(idx)

    int_ptr, int_idx = pack([x.int for x in nodes])
    bnd_ptr, bnd_idx = pack([x.bnd for x in nodes])
    iloc_ptr, iloc_idx = pack([x.int for x in locs])
    bloc_ptr, bloc_idx = pack([x.bnd for x in locs])
    return dict(
        nnodes=n, left=left, right=right,
        int_ptr=int_ptr, int_idx=int_idx, bnd_ptr=bnd_ptr, bnd_idx=bnd_idx,
        iloc_ptr=iloc_ptr, iloc_idx=iloc_idx, bloc_ptr=bloc_ptr, bloc_idx=bloc_idx,
    )


def _symbolic_result(L, h):
    import ctypes as C  # noqa: F401

    from . import _lib

    try:
        n = int(L.hs_symbolic_size(h))
        perm = np.ctypeslib.as_array(L.hs_symbolic_perm(h), shape=(n,)).copy() if n else np.zeros(0, np.int64)
        t = _lib.hs_tree()
        _lib.check(L.hs_symbolic_tree(h, C.byref(t)))
        k = int(t.nnodes)

        def arr(ptr, m):
            return np.ctypeslib.as_array(ptr, shape=(m,)).copy() if m else np.zeros(0, np.int64)

        tree = dict(nnodes=k, left=arr(t.left, k), right=arr(t.right, k))
        for name in ("int", "bnd", "iloc", "bloc"):
            ptr = arr(getattr(t, name + "_ptr"), k + 1)
            tree[name + "_ptr"] = ptr
            tree[name + "_idx"] = arr(getattr(t, name + "_idx"), int(ptr[-1]))
    finally:
        L.hs_symbolic_free(h)
    return tree, perm


def native_graph_symbolic(A, nmax=100):
    """Nested dissection of a general sparse matrix from its graph, then ``symfact!`` -> ``postorder`` -> ``permuted!``, all in the C++
    symbolic layer (``hs_symbolic_from_graph``, include/hs_symbolic.h).  Returns ``(tree, perm)`` like :func:`native_symbolic`.
    The Python mirror of the tree builder is ``problems.graph_nested_dissection``.  No GPU needed."""
    import ctypes as C

    import scipy.sparse as sp

    from . import _lib

    A = sp.csc_matrix(A)
    A.sort_indices()
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.hs_symbolic_from_graph(A.shape[0], colptr.ctypes.data_as(_lib.p_i64), rowval.ctypes.data_as(_lib.p_i64), int(nmax), C.byref(h)))
    return _symbolic_result(L, h)


def native_symbolic(fathers, lsons, rsons, ninter, inter, nbound, bound):
    """The whole host pipeline of the reference's scenario (``parse_elimtree`` -> ``symfact!`` -> ``postorder`` ->
    ``permuted!(nd, invperm(perm))``, test/rungmres.jl:15-19) in the C++ symbolic layer (``include/hs_symbolic.h``).

    Returns ``(tree, perm)``: ``tree`` is the flat post-ordered tree in the permuted numbering -- pass it as ``nd``
    (with ``nd_loc=None``) to :func:`factor` / ``StagedSolver`` -- and ``perm`` (1-based) is the elimination order:
    factor ``A[perm-1][:, perm-1]``.  No GPU needed."""
    import ctypes as C

    from . import _lib

    fathers, lsons, rsons, ninter, nbound = (_ivec(a) for a in (fathers, lsons, rsons, ninter, nbound))
    inter = np.asfortranarray(np.asarray(inter, dtype=np.int64))
    bound = np.asfortranarray(np.asarray(bound, dtype=np.int64))
    inter = inter.reshape(inter.shape[0], -1, order="F") if inter.ndim == 2 else inter.reshape(-1, 1, order="F")
    bound = bound.reshape(bound.shape[0], -1, order="F") if bound.ndim == 2 else bound.reshape(-1, 1, order="F")
    nn = len(fathers)
    if not (nn == len(lsons) == len(rsons) == len(ninter) == len(nbound) == inter.shape[1] == bound.shape[1]):
        raise _lib.DimensionMismatch("DimensionMismatch: dimensions inconsistent among inputs")
    L = _lib.lib()
    h = C.c_void_p()
    p = lambda a: a.ctypes.data_as(_lib.p_i64)  # noqa: E731
    _lib.check(L.hs_symbolic_from_elimtree(nn, p(fathers), p(lsons), p(rsons), p(ninter), p(inter), inter.shape[0], p(nbound), p(bound),
                                           bound.shape[0], C.byref(h)))
    return _symbolic_result(L, h)
