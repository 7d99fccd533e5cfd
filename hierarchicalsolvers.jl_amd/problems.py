"""Synthetic problems in the reference's input format.

The four ``.mat`` inputs the reference's script reads are stripped from the
checkout (``/root/reference/.MISSING_LARGE_BLOBS``), so inputs are generated:

* matrix: regular-grid Poisson (5/7-point) or Helmholtz (``-Lap - k^2`` with a
  first-order absorbing term on the domain boundary -> complex symmetric,
  non-Hermitian) in CSC, lexicographic numbering (x fastest);
* tree: geometric nested dissection by recursive coordinate bisection along the
  longest axis, in the reference's *disjoint-ownership* form -- every DOF
  belongs to exactly one leaf box, ``bnd(B)`` = DOFs of ``B`` with a stencil
  neighbour outside ``B``, parent ``int = (bnd(l) | bnd(r)) - bnd(parent)`` --
  which is what ``parse_elimtree`` (``src/nesteddissection.jl:105-148``) yields
  from the serialized 7-array format;
* file I/O: MATLAB v5 ``.mat`` with ``A``, ``b`` and struct ``elim_tree`` holding
  ``fathers, lsons, rsons, ninter, nbound`` (1 x nnodes) and ``inter, bound``
  (maxlen x nnodes), the layout ``util/read_problem.jl:7-20`` reads.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from .nesteddissection import NDNode, parse_elimtree, serialize_elimtree

__all__ = ["grid_matrix", "grid_nested_dissection", "graph_nested_dissection", "make_problem", "write_problem", "read_problem", "NAMED"]


def _lap1d(n):
    return sp.diags([-np.ones(n - 1), 2.0 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")


def grid_matrix(shape, kind="poisson", ppw=10.0, dtype=None):
    """Stencil matrix on a regular grid of ``shape`` points (all grid nodes are DOFs).

    ``poisson``: 2d on the diagonal, -1 to each neighbour (h^2-scaled Laplacian).
    ``helmholtz``: Laplacian - (kh)^2 I - i (kh) * (#exposed faces) on boundary nodes,
    ``kh = 2 pi / ppw`` (``ppw`` points per wavelength)."""
    shape = tuple(int(s) for s in shape)
    d = len(shape)
    eyes = [sp.identity(s, format="csr") for s in shape]
    A = None
    for ax in range(d):
        term = None
        # numbering: x (axis 0) fastest => kron order is reversed axes
        for a in reversed(range(d)):
            f = _lap1d(shape[a]) if a == ax else eyes[a]
            term = f if term is None else sp.kron(term, f, format="csr")
        A = term if A is None else A + term
    if kind == "poisson":
        A = A.astype(dtype or np.float64)
    elif kind == "helmholtz":
        kh = 2.0 * np.pi / ppw
        faces = np.zeros(shape[::-1], dtype=np.float64)  # C-order array indexed [z][y][x]
        for ax in range(d):
            sl_lo = [slice(None)] * d
            sl_hi = [slice(None)] * d
            sl_lo[d - 1 - ax] = 0
            sl_hi[d - 1 - ax] = -1
            faces[tuple(sl_lo)] += 1.0
            faces[tuple(sl_hi)] += 1.0
        diag = -(kh**2) * np.ones(A.shape[0]) - 1j * kh * faces.reshape(-1)
        A = (A.astype(np.complex128) + sp.diags(diag)).astype(dtype or np.complex128)
    else:
        raise ValueError(kind)
    A = sp.csc_matrix(A)
    A.sort_indices()
    return A


def _box_ids(lo, hi, shape):
    """Global 1-based lexicographic ids of the box [lo, hi) as an array of the box's shape (axis 0 = x fastest)."""
    d = len(shape)
    strides = np.cumprod((1,) + tuple(shape[:-1]))
    ids = np.zeros([hi[a] - lo[a] for a in range(d)], dtype=np.int64)
    for a in range(d):
        sh = [1] * d
        sh[a] = hi[a] - lo[a]
        ids = ids + (np.arange(lo[a], hi[a], dtype=np.int64) * strides[a]).reshape(sh)
    return ids + 1


def _bnd_mask(lo, hi, shape):
    """Mask (box-shaped) of DOFs having a stencil neighbour outside the box but inside the grid."""
    d = len(shape)
    m = np.zeros([hi[a] - lo[a] for a in range(d)], dtype=bool)
    for a in range(d):
        if lo[a] > 0:
            sl = [slice(None)] * d
            sl[a] = 0
            m[tuple(sl)] = True
        if hi[a] < shape[a]:
            sl = [slice(None)] * d
            sl[a] = -1
            m[tuple(sl)] = True
    return m


def _order(ids, mask):
    # Fortran order = x fastest = ascending global id
    return ids.reshape(-1, order="F")[mask.reshape(-1, order="F")]


def grid_nested_dissection(shape, nmax):
    """Geometric nested dissection of a regular grid: returns the raw :class:`NDNode` tree
    (global 1-based DOF ids, before ``symfact``)."""
    shape = tuple(int(s) for s in shape)
    d = len(shape)

    def build(lo, hi):
        ext = [hi[a] - lo[a] for a in range(d)]
        ids = _box_ids(lo, hi, shape)
        bmask = _bnd_mask(lo, hi, shape)
        if int(np.prod(ext)) <= nmax or max(ext) < 2:
            return NDNode(_order(ids, ~bmask), _order(ids, bmask))
        ax = int(np.argmax(ext))  # longest axis, first on ties
        mid = lo[ax] + ext[ax] // 2
        hi_l = list(hi)
        hi_l[ax] = mid
        lo_r = list(lo)
        lo_r[ax] = mid
        left = build(tuple(lo), tuple(hi_l))
        right = build(tuple(lo_r), tuple(hi))
        # int = (bnd(l) | bnd(r)) - bnd(box): assemble child masks in box coordinates
        cm = np.zeros(ext, dtype=bool)
        sl = [slice(None)] * d
        sl[ax] = slice(0, mid - lo[ax])
        cm[tuple(sl)] = _bnd_mask(tuple(lo), tuple(hi_l), shape)
        sl[ax] = slice(mid - lo[ax], ext[ax])
        cm[tuple(sl)] = _bnd_mask(tuple(lo_r), tuple(hi), shape)
        return NDNode(_order(ids, cm & ~bmask), _order(ids, bmask), left, right)

    return build(tuple([0] * d), shape)


def graph_nested_dissection(A, nmax=100):
    """Nested dissection of a GENERAL sparse matrix from its graph alone (no coordinates): the raw :class:`NDNode` tree in the
    disjoint-ownership form the reference consumes (``nesteddissection.jl:19-21,105-148``; the generator that made the
    reference's ``.mat`` trees is not part of it) -- every DOF belongs to exactly one leaf, ``bnd(B)`` = DOFs of ``B`` with a
    neighbour outside ``B``, a parent's ``int = (bnd(l) | bnd(r)) - bnd(parent)``.

    A set is halved by a breadth-first sweep of its induced subgraph from a pseudo-peripheral vertex (the first half of the
    sweep against the rest; further components follow the first one), until it holds at most ``nmax`` DOFs.  Global ids are
    1-based and ascending inside every index vector, like the grid generator's."""
    from scipy.sparse.csgraph import breadth_first_order

    A = sp.csr_matrix(A)
    n = A.shape[0]
    G = sp.csr_matrix((np.ones(A.nnz, dtype=np.int8), A.indices, A.indptr), shape=A.shape)
    G = ((G + G.T) > 0).astype(np.int8).tolil()
    G.setdiag(0)
    G = sp.csr_matrix(G)
    G.eliminate_zeros()
    deg = np.diff(G.indptr)

    def sweep(H, start):
        """Breadth-first order of every vertex of H: the component of `start` first, then the others."""
        m = H.shape[0]
        order = list(breadth_first_order(H, start, directed=False, return_predecessors=False))
        if len(order) < m:
            seen = np.zeros(m, dtype=bool)
            seen[order] = True
            while len(order) < m:
                nxt = int(np.flatnonzero(~seen)[0])
                o = breadth_first_order(H, nxt, directed=False, return_predecessors=False)
                seen[o] = True
                order.extend(o)
        return np.asarray(order, dtype=np.int64)

    def build(V, H):
        """V: ascending global 0-based ids, H: induced subgraph on V."""
        inside = np.diff(H.indptr)
        bmask = deg[V] > inside  # a neighbour outside V
        if len(V) <= nmax or len(V) < 2:
            return NDNode(V[~bmask] + 1, V[bmask] + 1)
        start = 0
        for _ in range(2):  # pseudo-peripheral vertex: the last vertex of a sweep, twice
            start = int(breadth_first_order(H, start, directed=False, return_predecessors=False)[-1])
        order = sweep(H, start)
        half = (len(V) + 1) // 2
        sel = np.zeros(len(V), dtype=bool)
        sel[order[:half]] = True
        i1, i2 = np.flatnonzero(sel), np.flatnonzero(~sel)
        H1, H2 = H[i1][:, i1], H[i2][:, i2]
        left, right = build(V[i1], H1), build(V[i2], H2)
        cb = np.zeros(len(V), dtype=bool)  # boundary of either child, in V's numbering
        cb[i1] = deg[V[i1]] > np.diff(H1.indptr)
        cb[i2] = deg[V[i2]] > np.diff(H2.indptr)
        return NDNode(V[cb & ~bmask] + 1, V[bmask] + 1, left, right)

    import sys

    lim = sys.getrecursionlimit()
    sys.setrecursionlimit(max(lim, 10000))
    try:
        return build(np.arange(n, dtype=np.int64), G)
    finally:
        sys.setrecursionlimit(lim)


# name -> (shape, kind, nmax); sizes inferred from the stripped blobs' names
NAMED = {
    "poisson2d_p1_h64_nmax100": ((65, 65), "poisson", 100),
    "poisson2d_p1_h128_nmax100": ((129, 129), "poisson", 100),
    "helmholtz2d_p1_h64_nmax100": ((65, 65), "helmholtz", 100),
    "helmholtz2d_p1_h128_nmax100": ((129, 129), "helmholtz", 100),
    "poisson3d_32": ((32, 32, 32), "poisson", 512),
    "poisson3d_64": ((64, 64, 64), "poisson", 4096),
    "poisson3d_96": ((96, 96, 96), "poisson", 4096),
    "poisson3d_128": ((128, 128, 128), "poisson", 4096),
    "helmholtz3d_32": ((32, 32, 32), "helmholtz", 512),
    "helmholtz3d_64": ((64, 64, 64), "helmholtz", 4096),
    "helmholtz3d_96": ((96, 96, 96), "helmholtz", 4096),
    "helmholtz3d_112": ((112, 112, 112), "helmholtz", 4096),  # largest complex 3-D problem whose dense factors fit one MI355X (163 GiB)
    "helmholtz3d_128": ((128, 128, 128), "helmholtz", 4096),  # 278 GiB of dense complex factors: needs >= 2 GPUs
    # the 8-GPU compressed configurations (hs_options.mf over ranks; tools/size_model.py: per-rank bytes with the rank constant measured on
    # Helmholtz 112^3 at 1e-4): leaves of 1,024 DOFs -- with 4,096 the dense leaf blocks alone are 137 GiB per rank at 256^3
    "helmholtz3d_192": ((192, 192, 192), "helmholtz", 1024),  # mf = 2 / 3 fit 8 x 288 GB (130 / 151 GiB on the busiest rank)
    "helmholtz3d_224": ((224, 224, 224), "helmholtz", 1024),  # mf = 2 only (217 GiB on the busiest rank)
    "helmholtz3d_256": ((256, 256, 256), "helmholtz", 1024),  # BASELINE.json config 5: modelled at 338 GiB per rank with mf = 2 -- does NOT fit yet
}


def make_problem(name_or_shape, kind=None, nmax=None, seed=123, rhs="ones"):
    """Returns ``(A, b, nd)``: CSC matrix, right-hand side and raw elimination tree.

    ``rhs='ones'`` gives ``b = A*1`` (exact solution known); ``'randn'`` a seeded Gaussian
    (the reference script seeds Julia's RNG with 123, ``test/rungmres.jl:7``)."""
    if isinstance(name_or_shape, str):
        shape, kind_, nmax_ = NAMED[name_or_shape]
        kind = kind or kind_
        nmax = nmax or nmax_
    else:
        shape = tuple(name_or_shape)
        kind = kind or "poisson"
        nmax = nmax or 100
    A = grid_matrix(shape, kind)
    nd = grid_nested_dissection(shape, nmax)
    n = A.shape[0]
    if rhs == "ones":
        b = A @ np.ones(n, dtype=A.dtype)
    else:
        rng = np.random.default_rng(seed)
        b = rng.standard_normal(n).astype(A.dtype)
        if np.iscomplexobj(b):
            b = b + 1j * rng.standard_normal(n)
    return A, b, nd


def write_problem(path, A, b, nd):
    """Write ``A``, ``b``, ``elim_tree`` as a MATLAB v5 file in the layout ``util/read_problem.jl`` expects."""
    import scipy.io

    fathers, lsons, rsons, ninter, inter, nbound, bound = serialize_elimtree(nd)
    row = lambda v: np.asarray(v, dtype=np.float64).reshape(1, -1)  # noqa: E731  (1 x nnodes, dropdims(dims=1))
    tree = dict(
        fathers=row(fathers), lsons=row(lsons), rsons=row(rsons), ninter=row(ninter), nbound=row(nbound),
        inter=inter.astype(np.float64), bound=bound.astype(np.float64),
    )
    scipy.io.savemat(path, dict(A=sp.csc_matrix(A), b=np.asarray(b).reshape(-1, 1), elim_tree=tree), do_compression=True)


def read_problem(path):
    """``read_problem(filepath) -> (A, b, nd)`` (util/read_problem.jl:5-25)."""
    import scipy.io

    m = scipy.io.loadmat(path, squeeze_me=False, struct_as_record=True)
    t = m["elim_tree"][0, 0]
    vec = lambda k: np.asarray(t[k]).reshape(-1).astype(np.int64)  # noqa: E731
    nd = parse_elimtree(
        vec("fathers"), vec("lsons"), vec("rsons"), vec("ninter"),
        np.asarray(t["inter"]).astype(np.int64), vec("nbound"), np.asarray(t["bound"]).astype(np.int64),
    )
    return sp.csc_matrix(m["A"]), np.asarray(m["b"]).reshape(-1), nd
