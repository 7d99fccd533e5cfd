"""Oracle for the compressed branch with an HSS INTERIOR BLOCK (TEST INFRASTRUCTURE -- never on the product path).

Restates `_factor_branch(..., Val(true))` (reference src/factorization.jl:78-112) the way the device path of
`hs_options.hss_d` builds it: `D = Aii` is an HSS matrix (`hs_hss.compress`, standing in for the `BlockFactorization` over
`HssMatrix` blocks of src/blockmatrix.jl:121-130) and every `Aii^-1 X` goes through its ULV-type solve (`blockldiv!`,
:134-156); `L`, `R` are low-rank (`pqrfact`, src/factorization.jl:99-100,171-182); `S = Abb - Abi*R` with that low-rank `R`
(:228-242), kept dense between fronts (the HSS hand-over of C2/C3 is not restated -- it is not built either).  The root is
never flagged (:15) but receives its children's blocks and factors an HSS `D` too (:67,126).

HssMatrices.jl and LowRankApprox.jl are absent from the reference tree: PARITY UNPINNED; tests compare solution errors and
GMRES iteration counts of the product with this restatement, and this restatement with SuperLU.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import breadth_first_order

from . import hs_hss as HS
from . import hs_oracle as O
from . import hs_oracle_lr as OL


def bisect_order(G):
    """Recursive bisection of the graph G (scipy CSR pattern): first half of a breadth-first sweep from a pseudo-peripheral
    vertex against the rest, split where the HSS cluster tree splits its index range (ceil(size/2)), down to 32 vertices."""
    n = G.shape[0]
    order = np.arange(n)

    def sweep(H, start):
        o = list(breadth_first_order(H, start, directed=False, return_predecessors=False))
        seen = np.zeros(H.shape[0], dtype=bool)
        seen[o] = True
        while len(o) < H.shape[0]:
            nxt = int(np.flatnonzero(~seen)[0])
            more = breadth_first_order(H, nxt, directed=False, return_predecessors=False)
            seen[more] = True
            o.extend(more)
        return np.asarray(o)

    def rec(lo, hi):
        if hi - lo <= 32:
            return
        idx = order[lo:hi]
        H = G[idx][:, idx]
        start = 0
        for _ in range(2):
            start = int(sweep(H, start)[-1])
        order[lo:hi] = idx[sweep(H, start)]
        mid = lo + (hi - lo + 1) // 2
        rec(lo, mid)
        rec(mid, hi)

    rec(0, n)
    return order


class HssD:
    """`D = Aii` as an HSS matrix of `Aii[q, q]` with its elimination; `solve(X) = Aii^-1 X` in the front's own order."""

    def __init__(self, Aii, q, leafsize, atol, rtol, kest=64):
        self.q = q
        M = Aii[np.ix_(q, q)] if q is not None else Aii
        self.H = HS.compress(M, leafsize=leafsize, atol=atol, rtol=rtol, kest=kest, level_scale=0.5, fill=0.8)
        self.F = HS.rs_factor(self.H)
        self.hssrank = HS.hssrank(self.H)

    def solve(self, B):
        if self.q is None:
            return HS.rs_solve(self.F, B)
        X = np.empty(B.shape, dtype=np.result_type(B.dtype, self.H.dtype))
        X[self.q] = HS.rs_solve(self.F, B[self.q])
        return X


class Node:
    __slots__ = ("D", "Abi", "R", "S", "int", "bnd", "left", "right", "hss")

    def __init__(self, D, Abi, R, S, int_, bnd, left, right, hss):
        self.D, self.Abi, self.R, self.S, self.int, self.bnd, self.left, self.right, self.hss = D, Abi, R, S, int_, bnd, left, right, hss


def factor(A, nd, nd_loc, hss_min=1024, hss_leaf=256, dexp=2, opts=None, **kw):
    opts = (opts or O.SolverOptions()).copy(**kw)
    O.chkopts(opts)
    swlevel = max(O.depth(nd) + opts.swlevel, 0) if opts.swlevel < 0 else opts.swlevel
    A = sp.csc_matrix(A)
    G = sp.csr_matrix((abs(A) + abs(A).T) > 0)
    return _factor(A, G, nd, nd_loc, 1, swlevel, opts, hss_min, hss_leaf, 10.0 ** (-dexp))


def _factor(A, G, nd, nd_loc, level, swlevel, opts, hss_min, hss_leaf, dsc):
    if O.isleaf(nd):
        f = O._factor_leaf(A, nd, nd_loc, False, opts)
        return Node(f.D, f.L, f.R, f.S, nd.int, nd.bnd, None, None, False)
    Fl = _factor(A, G, nd.left, nd_loc.left, level + 1, swlevel, opts, hss_min, hss_leaf, dsc)
    Fr = _factor(A, G, nd.right, nd_loc.right, level + 1, swlevel, opts, hss_min, hss_leaf, dsc)
    int1 = nd.left.bnd[nd_loc.left.int - 1]
    bnd1 = nd.left.bnd[nd_loc.left.bnd - 1]
    int2 = nd.right.bnd[nd_loc.right.int - 1]
    bnd2 = nd.right.bnd[nd_loc.right.bnd - 1]
    Aii, Aib, Abi, Abb = O._assemble_blocks(A, O._dense(Fl.S), O._dense(Fr.S), int1, int2, bnd1, bnd2)
    perm = np.concatenate([nd_loc.int, nd_loc.bnd]) - 1
    flag = (level <= swlevel) and (len(nd.bnd) >= opts.swsize)
    use_hss = level <= swlevel and swlevel > 0 and len(nd.int) >= hss_min and (flag or len(nd.bnd) == 0)
    if not use_hss:
        if flag and len(nd.bnd) and len(nd.int):  # plain compressed front (hs_oracle_lr)
            D = O.blockfactor(Aii)
            L = OL._lgauss(D, Abi, 0.5 * opts.atol, 0.5 * opts.rtol)
            R = OL._rgauss(D, Aib, 0.5 * opts.atol, 0.5 * opts.rtol)
            S = Abb.dense() - (Abi.dense() @ R.U) @ R.V.conj().T
            return Node(D, L, R, S[np.ix_(perm, perm)], nd.int, nd.bnd, Fl, Fr, False)
        f = O._factor_branch(A, _as_fn(Fl), _as_fn(Fr), nd, nd_loc, False, opts)
        return Node(f.D, f.L, f.R, f.S, nd.int, nd.bnd, Fl, Fr, False)
    ids = nd.int - 1
    q = bisect_order(G[ids][:, ids])
    D = HssD(Aii.dense(), q, hss_leaf, opts.atol * dsc, opts.rtol * dsc)
    if len(nd.bnd) == 0:
        return Node(D, None, None, np.zeros((0, 0), Aii.dense().dtype), nd.int, nd.bnd, Fl, Fr, True)
    QL, RL, pL = OL.pqrfact(Abi.dense(), 0.5 * opts.atol, 0.5 * opts.rtol)
    QR, RR, pR = OL.pqrfact(Aib.dense(), 0.5 * opts.atol, 0.5 * opts.rtol)
    Abi_lr = OL.LowRankMatrix(QL, RL[:, np.argsort(pL)].conj().T)  # Abi ~= Q * R[:, invperm p]
    R = OL.LowRankMatrix(D.solve(QR), RR[:, np.argsort(pR)].conj().T)  # R = Aii^-1 Aib
    S = Abb.dense() - QL @ ((RL[:, np.argsort(pL)] @ R.U) @ R.V.conj().T)
    return Node(D, Abi_lr, R, S[np.ix_(perm, perm)], nd.int, nd.bnd, Fl, Fr, True)


def _as_fn(x):
    """A child as the FactorNode `_factor_branch` of the dense oracle expects (it only reads S)."""
    return O.FactorNode(None, x.S, None, None, x.int, x.bnd, [], [], None, None)


def ldiv(F, B):
    B = np.asarray(B)
    vec = B.ndim == 1
    C = np.array(B.reshape(len(B), -1), dtype=np.result_type(B.dtype, np.float64 if not np.iscomplexobj(O._dense(F.S)) else np.complex128))
    keep = {}
    _fwd(F, C, keep)
    _bwd(F, C, keep)
    return C[:, 0] if vec else C


def _fwd(F, rhs, keep):
    for c in (F.left, F.right):
        if c is not None:
            _fwd(c, rhs, keep)
    i, b = F.int - 1, F.bnd - 1
    if F.hss:  # t = D^-1 rhs[int];  rhs[bnd] -= Abi t
        t = F.D.solve(rhs[i])
        keep[id(F)] = t
        if len(b):
            rhs[b] = rhs[b] - F.Abi.U @ (F.Abi.V.conj().T @ t)
    else:  # factornode.jl:77-82, 89-99
        if len(b):
            rhs[b] = rhs[b] - O._dense(F.Abi) @ rhs[i]


def _bwd(F, rhs, keep):
    i, b = F.int - 1, F.bnd - 1
    if F.hss:
        t = keep[id(F)]
        rhs[i] = t - (F.R.U @ (F.R.V.conj().T @ rhs[b]) if len(b) else 0)
    else:
        d = O.blockldiv_inplace(F.D, rhs[i]) if isinstance(F.D, O.BlockFactorization) else O._ldiv(F.D, rhs[i])
        rhs[i] = d - (O._dense(F.R) @ rhs[b] if len(b) else 0)
    for c in (F.left, F.right):
        if c is not None:
            _bwd(c, rhs, keep)


def maxrank(F):
    r = 0
    for c in (F.left, F.right):
        if c is not None:
            r = max(r, maxrank(c))
    if F.hss:
        r = max(r, F.D.hssrank)
    for M in (F.Abi, F.R):
        if isinstance(M, OL.LowRankMatrix):
            r = max(r, M.rank)
    return r
