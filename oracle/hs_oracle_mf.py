"""Oracle for the MATRIX-FREE compressed branch: Schur complements travel between fronts as HSS matrices
(TEST INFRASTRUCTURE -- never on the product path; the device counterpart is NOT built yet).

Restates `_factor_branch(..., Val(true))` (reference src/factorization.jl:78-112) with the data flow of the reference:

* C3 `_assemble_blocks` for HSS children (:126-140): `Aii = [S1.A11  A[int1,int2]; A[int2,int1]  S2.A11]` with the children's
  HSS diagonal blocks and SPARSE couplings, `Aib`, `Abi` = low-rank generators of the children (`U*B12`, `V`) + sparse blocks,
  `Abb = [S1.A22  A[bnd1,bnd2]; A[bnd2,bnd1]  S2.A22]` -- nothing is densified;
* B2' `blockfactor` over HSS blocks (src/blockmatrix.jl:121-130): `S22 = A22 - A21*(A11\\A12)` -- here compressed matrix-free from
  the operator `X -> A22 X - A21 (A11^-1 (A12 X))` and its entries (the role of the HSS-by-HSS arithmetic + `recompress!`,
  which HssMatrices.jl does on generators; that package is absent from the reference tree);
* B5 `blockldiv!` (:134-156) with HSS solves for `A11` and `S22`;
* C5 low-rank Gauss transforms from the children's generators and the sparse couplings (:184-209); the truncation itself is
  the pivoted QR of hs_oracle_lr on the assembled thin factors (`pqrfact`; LowRankApprox.jl is absent too);
* C6 `_schur_complement` / `_sample_schur!` / `_getindex_schur` (:228-249): `S = P (Abb - Abi*R) P'` as an operator with products
  and entries, compressed by `randcompress_adaptive` (:110) over `bisection_cluster((|int_loc|, |bnd|))` (:109);
* F2 `_factor_leaf(..., Val(true))` (:45-59): `compress(S[perm,perm], cl, cl)`.
C2 `_equilibrate_clusters` (:143-168) has no counterpart: the blocks are never added generator by generator, every compression
samples an operator, so the children's cluster trees need not be compatible.  PARITY UNPINNED.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from . import hs_hss as HS
from . import hs_oracle as O
from . import hs_oracle_lr as OL


def hss_transpose(H):
    """H^T (plain transpose): D -> D^T, B12 <-> B21^T; the bases are shared (U = V)."""
    nodes = []
    for x in H.nodes:
        y = HS.HssNode(x.lo, x.hi, x.level, x.parent)
        y.left, y.right, y.m, y.r, y.p, y.T, y.sk = x.left, x.right, x.m, x.r, x.p, x.T, x.sk
        y.D = None if x.D is None else x.D.T
        y.B12 = None if x.B21 is None else x.B21.T
        y.B21 = None if x.B12 is None else x.B12.T
        nodes.append(y)
    return HS.Hss(H.n, nodes, H.dtype)


class SBlock:
    """One child's Schur complement `S` (ordered [int_loc; bnd_loc], split at n1 = |int_loc|), dense or HSS: the four blocks
    the parent reads (factorization.jl:127-137)."""

    def __init__(self, S, n1):
        self.S, self.n1 = S, n1
        self.n = S.n if isinstance(S, HS.Hss) else S.shape[0]
        self.hss = isinstance(S, HS.Hss)
        if self.hss and n1 == self.n:  # the whole boundary becomes the parent's interior (children of the root): S = A11
            z = np.zeros((0, 0), dtype=S.dtype)
            self.A, self.At = [S, None], [hss_transpose(S), None]
            self.U12, self.V12 = np.zeros((self.n, 0), dtype=S.dtype), z
            self.U21, self.V21 = z, np.zeros((self.n, 0), dtype=S.dtype)
        elif self.hss:
            self.A = [HS.hss_child(S, 0), HS.hss_child(S, 1)]
            self.At = [hss_transpose(a) for a in self.A]
            U1, B12, U2, B21 = HS.hss_offdiag(S)
            self.U12, self.V12 = U1 @ B12, U2  # A12 = U12 V12^T
            self.U21, self.V21 = U2 @ B21, U1  # A21 = U21 V21^T

    def diag_mul(self, k, X, trans=False):  # A11 (k = 0) or A22 (k = 1) times X
        if X.shape[0] == 0:
            return X.copy()
        if self.hss:
            return HS.hss_matvec(self.At[k] if trans else self.A[k], X)
        a = slice(0, self.n1) if k == 0 else slice(self.n1, self.n)
        M = self.S[a, a]
        return (M.T if trans else M) @ X

    def diag_get(self, k, I, J):
        if self.hss:
            return HS.hss_getindex(self.A[k], I, J)
        o = 0 if k == 0 else self.n1
        return self.S[np.ix_(np.asarray(I) + o, np.asarray(J) + o)]

    def off(self, k):  # dense thin factors (U, V) of A12 (k = 0) or A21 (k = 1): block = U V^T
        if self.hss:
            return (self.U12, self.V12) if k == 0 else (self.U21, self.V21)
        B = self.S[: self.n1, self.n1 :] if k == 0 else self.S[self.n1 :, : self.n1]
        return B, np.eye(B.shape[1], dtype=B.dtype)


class Op:
    """Square operator seen through products and entries (HssMatrices' LinearMap, factorization.jl:234)."""

    def __init__(self, n, dtype, mul, mulT, get):
        self.shape, self.dtype, self.mul, self.mulT, self._get = (n, n), dtype, mul, mulT, get

    def __getitem__(self, ij):  # called as A[np.ix_(I, J)]
        I, J = ij
        return self._get(np.asarray(I).reshape(-1), np.asarray(J).reshape(-1))


def _compress(op, opts, first_split=None, scale=1.0):
    return HS.compress(op, leafsize=opts.leafsize, atol=opts.atol * scale, rtol=opts.rtol * scale, kest=max(opts.kest, 16), first_split=first_split,
                       mul=op.mul, mulT=op.mulT, level_scale=0.5, fill=0.8)


class BlockD:
    """`D = blockfactor(Aii)` over HSS blocks (blockmatrix.jl:121-130): A11 (HSS, from the left child), sparse A12 / A21, and
    S22 = A22 - A21 A11^-1 A12 compressed from its operator; `solve` = blockldiv! (:134-144)."""

    def __init__(self, c1, c2, C12, C21, opts, dsc):
        n1, n2 = c1.n1, c2.n1
        dt = np.result_type(C12.dtype, np.float64)
        self.n1, self.n2 = n1, n2
        self.C12, self.C21 = sp.csr_matrix(C12), sp.csr_matrix(C21)
        self.c1 = c1
        # A11^-1: HSS elimination of the left child's A11 (a dense child: LU through the dense solve)
        if c1.hss:
            f11 = HS.rs_factor(c1.A[0])
            f11t = HS.rs_factor(c1.At[0])
            self.s11 = lambda B: HS.rs_solve(f11, B)
            self.s11t = lambda B: HS.rs_solve(f11t, B)
        else:
            M = c1.S[:n1, :n1]
            self.s11 = lambda B: O._ldiv(M, B)
            self.s11t = lambda B: O._ldiv(M.T, B)
        C12d, C21d = self.C12, self.C21

        def mul(X):
            return c2.diag_mul(0, X) - C21d @ self.s11(C12d @ X)

        def mulT(X):
            return c2.diag_mul(0, X, trans=True) - C12d.T @ self.s11t(C21d.T @ X)

        def get(I, J):
            return c2.diag_get(0, I, J) - (C21d[I] @ self.s11(C12d[:, J].toarray()))

        self.S22 = _compress(Op(n2, dt, mul, mulT, get), opts, scale=dsc)
        self.f22 = HS.rs_factor(self.S22)
        self.hssrank = max(HS.hssrank(self.S22), HS.hssrank(c1.A[0]) if c1.hss else 0)

    def solve(self, B):  # blockldiv!: [A11 A12; A21 A22] \ B with S22
        B1, B2 = B[: self.n1], B[self.n1 :]
        y1 = self.s11(B1)
        x2 = HS.rs_solve(self.f22, B2 - self.C21 @ y1)
        return np.concatenate([y1 - self.s11(self.C12 @ x2), x2], axis=0)


class SingleD:
    """`D = Aii` as ONE HSS matrix over a recursive-bisection order of the front's interior graph, compressed MATRIX-FREE from the
    operator `[S1.A11  A12; A21  S2.A11]` (children's HSS blocks: products and entry access; sparse couplings) -- the formulation the
    device path of `hs_options.hss_d` uses, minus its dense assembly.  No `A11^-1 * A12` is ever needed: products cost two HSS matvecs
    and a sparse product, a block of entries two `hss_getindex` calls and a sparse gather."""

    def __init__(self, c1, c2, C12, C21, G, opts, dsc):
        from .hs_oracle_hss import bisect_order

        n1, n2 = c1.n1, c2.n1
        n = n1 + n2
        dt = np.result_type(C12.dtype, np.float64)
        C12, C21 = sp.csr_matrix(C12), sp.csr_matrix(C21)
        q = bisect_order(G)  # q[new position] = position in [int1; int2]
        self.q = q

        def mul0(X, trans=False):
            X1, X2 = X[:n1], X[n1:]
            if not trans:
                return np.concatenate([c1.diag_mul(0, X1) + C12 @ X2, C21 @ X1 + c2.diag_mul(0, X2)], axis=0)
            return np.concatenate([c1.diag_mul(0, X1, True) + C21.T @ X2, C12.T @ X1 + c2.diag_mul(0, X2, True)], axis=0)

        def get0(I, J):
            out = np.zeros((len(I), len(J)), dtype=dt)
            i1, i2, j1, j2 = I < n1, I >= n1, J < n1, J >= n1
            if i1.any() and j1.any():
                out[np.ix_(i1, j1)] = c1.diag_get(0, I[i1], J[j1])
            if i2.any() and j2.any():
                out[np.ix_(i2, j2)] = c2.diag_get(0, I[i2] - n1, J[j2] - n1)
            if i1.any() and j2.any():
                out[np.ix_(i1, j2)] = C12[I[i1]][:, J[j2] - n1].toarray()
            if i2.any() and j1.any():
                out[np.ix_(i2, j1)] = C21[I[i2] - n1][:, J[j1]].toarray()
            return out

        def mul(X):  # (M[q, q]) X
            Y = np.zeros_like(X, dtype=np.result_type(X.dtype, dt))
            Y[q] = X
            return mul0(Y)[q]

        def mulT(X):
            Y = np.zeros_like(X, dtype=np.result_type(X.dtype, dt))
            Y[q] = X
            return mul0(Y, True)[q]

        self.H = _compress(Op(n, dt, mul, mulT, lambda I, J: get0(q[I], q[J])), opts, scale=dsc)
        self.F = HS.rs_factor(self.H)
        self.hssrank = HS.hssrank(self.H)

    def solve(self, B):
        X = np.empty(B.shape, dtype=np.result_type(B.dtype, self.H.dtype))
        X[self.q] = HS.rs_solve(self.F, B[self.q])
        return X


class Node:
    __slots__ = ("D", "L", "R", "S", "int", "bnd", "left", "right", "kind", "n1")

    def __init__(self, D, L, R, S, int_, bnd, left, right, kind, n1):
        self.D, self.L, self.R, self.S, self.int, self.bnd, self.left, self.right, self.kind, self.n1 = D, L, R, S, int_, bnd, left, right, kind, n1


def factor(A, nd, nd_loc, dexp=2, dmode="block", opts=None, **kw):
    """dmode = "block": `D = blockfactor` over the children's HSS blocks (the reference's 2x2 form); "single": `D` as one HSS matrix over
    a bisection order of the interior graph, compressed matrix-free (the device formulation)."""
    opts = (opts or O.SolverOptions()).copy(**kw)
    O.chkopts(opts)
    swlevel = max(O.depth(nd) + opts.swlevel, 0) if opts.swlevel < 0 else opts.swlevel
    A = sp.csc_matrix(A)
    _factor.G = sp.csr_matrix((abs(A) + abs(A).T) > 0) if dmode == "single" else None
    return _factor(A, nd, nd_loc, 1, swlevel, opts, 10.0 ** (-dexp))


def _factor(A, nd, nd_loc, level, swlevel, opts, dsc):
    """Post-order recursion (factorization.jl:14-27): children first, then `factor_node`."""
    if O.isleaf(nd):
        return factor_node(A, nd, nd_loc, level, swlevel, opts, dsc, None, None)
    Fl = _factor(A, nd.left, nd_loc.left, level + 1, swlevel, opts, dsc)
    Fr = _factor(A, nd.right, nd_loc.right, level + 1, swlevel, opts, dsc)
    return factor_node(A, nd, nd_loc, level, swlevel, opts, dsc, Fl, Fr)


def remote_child(S, nd_child, nd_loc_child):
    """What a parent needs of a child it did not eliminate itself (a front of another rank, tests/test_dist_cpu.py): its Schur complement
    `S` (HSS or dense, already in the order [int_loc; bnd_loc]), |int_loc| and the index sets."""
    return Node(None, None, None, S, nd_child.int, nd_child.bnd, None, None, "remote", len(nd_loc_child.int))


def factor_node(A, nd, nd_loc, level, swlevel, opts, dsc, Fl, Fr):
    """Elimination of ONE front given its children's results (`_factor_leaf` / `_factor_branch`, factorization.jl:30-112): the unit a
    level-by-level (and multi-rank) schedule calls."""
    flag = (level <= swlevel) and (len(nd.bnd) >= opts.swsize)
    n1 = len(nd_loc.int)
    perm = np.concatenate([nd_loc.int, nd_loc.bnd]) - 1
    if O.isleaf(nd):
        f = O._factor_leaf(A, nd, nd_loc, False, opts)  # S already permuted to [int_loc; bnd_loc]
        S = f.S
        if flag and n1 > 0 and len(nd.bnd) > opts.leafsize:  # F2: compress(S[perm,perm], cl, cl), cl = bisection_cluster((n1, nb))
            S = HS.compress(S, leafsize=opts.leafsize, atol=opts.atol, rtol=opts.rtol, kest=max(opts.kest, 16), first_split=n1 if n1 < len(nd.bnd) else None,
                            level_scale=0.5, fill=0.8)
        return Node(f.D, f.L, f.R, S, nd.int, nd.bnd, None, None, "dense", n1)
    int1 = nd.left.bnd[nd_loc.left.int - 1]
    bnd1 = nd.left.bnd[nd_loc.left.bnd - 1]
    int2 = nd.right.bnd[nd_loc.right.int - 1]
    bnd2 = nd.right.bnd[nd_loc.right.bnd - 1]
    hss_children = isinstance(Fl.S, HS.Hss) and isinstance(Fr.S, HS.Hss)
    if not hss_children or len(int1) == 0 or len(int2) == 0:
        # dense assembly (children's S expanded when only one of them is HSS): the branches of hs_oracle / hs_oracle_lr
        Sl = HS.hss_full(Fl.S) if isinstance(Fl.S, HS.Hss) else Fl.S
        Sr = HS.hss_full(Fr.S) if isinstance(Fr.S, HS.Hss) else Fr.S
        Aii, Aib, Abi, Abb = O._assemble_blocks(A, Sl, Sr, int1, int2, bnd1, bnd2)
        if flag and len(nd.bnd) and len(nd.int):
            D = O.blockfactor(Aii)
            L = OL._lgauss(D, Abi, 0.5 * opts.atol, 0.5 * opts.rtol)
            R = OL._rgauss(D, Aib, 0.5 * opts.atol, 0.5 * opts.rtol)
            S = (Abb.dense() - (Abi.dense() @ R.U) @ R.V.conj().T)[np.ix_(perm, perm)]
            if n1 > 0 and len(nd.bnd) > opts.leafsize:
                S = HS.compress(S, leafsize=opts.leafsize, atol=opts.atol, rtol=opts.rtol, kest=max(opts.kest, 16), first_split=n1 if n1 < len(nd.bnd) else None,
                                level_scale=0.5, fill=0.8)
            return Node(D, L, R, S, nd.int, nd.bnd, Fl, Fr, "lr", n1)
        f = O._factor_branch(A, O.FactorNode(None, Sl, None, None, Fl.int, Fl.bnd, [], [], None, None),
                             O.FactorNode(None, Sr, None, None, Fr.int, Fr.bnd, [], [], None, None), nd, nd_loc, False, opts)
        return Node(f.D, f.L, f.R, f.S, nd.int, nd.bnd, Fl, Fr, "dense", n1)
    # ---- matrix-free branch: both children hand over HSS Schur complements (this includes the root, factorization.jl:67,126)
    c1, c2 = SBlock(Fl.S, Fl.n1), SBlock(Fr.S, Fr.n1)
    g = lambda I, J: A[I - 1][:, J - 1]  # noqa: E731  sparse couplings, 1-based index vectors
    if getattr(_factor, "G", None) is not None:
        ids = np.concatenate([int1, int2]) - 1
        D = SingleD(c1, c2, g(int1, int2), g(int2, int1), _factor.G[ids][:, ids], opts, dsc)
    else:
        D = BlockD(c1, c2, g(int1, int2), g(int2, int1), opts, dsc)
    nb1, nb2 = len(bnd1), len(bnd2)
    if nb1 + nb2 == 0:
        return Node(D, None, None, np.zeros((0, 0), A.dtype), nd.int, nd.bnd, Fl, Fr, "mf", n1)
    # C3 / C5: Aib = [U12_1 V12_1^T  A[int1,bnd2]; A[int2,bnd1]  U12_2 V12_2^T] as thin factors + sparse, truncated by pqrfact
    dt = np.result_type(A.dtype, np.float64)
    ni1, ni2 = len(int1), len(int2)

    def assemble(k, sa, sb):  # k = 0: Aib (rows int), k = 1: Abi (rows bnd); sa, sb = the two sparse couplings
        (Ua, Va), (Ub, Vb) = c1.off(k), c2.off(k)
        r1, r2 = (ni1, ni2) if k == 0 else (nb1, nb2)
        q1, q2 = (nb1, nb2) if k == 0 else (ni1, ni2)
        M = np.zeros((r1 + r2, q1 + q2), dtype=dt)
        M[:r1, :q1] = Ua @ Va.T
        M[r1:, q1:] = Ub @ Vb.T
        M[:r1, q1:] = sa.toarray()
        M[r1:, :q1] = sb.toarray()
        return M

    Aib = assemble(0, g(int1, bnd2), g(int2, bnd1))
    Abi = assemble(1, g(bnd1, int2), g(bnd2, int1))
    QL, RL, pL = OL.pqrfact(Abi, 0.5 * opts.atol, 0.5 * opts.rtol)
    QR, RR, pR = OL.pqrfact(Aib, 0.5 * opts.atol, 0.5 * opts.rtol)
    Abi_lr = OL.LowRankMatrix(QL, RL[:, np.argsort(pL)].conj().T)
    R = OL.LowRankMatrix(D.solve(QR), RR[:, np.argsort(pR)].conj().T)  # R = Aii^-1 Aib
    # C6: S = P (Abb - Abi R) P' as an operator (products: factorization.jl:242, entries: :248)
    W = (RL[:, np.argsort(pL)] @ R.U)  # rank(L) x rank(R): Abi R = QL W R.V^H
    C12b, C21b = sp.csr_matrix(g(bnd1, bnd2)), sp.csr_matrix(g(bnd2, bnd1))
    nb = nb1 + nb2
    ip = np.argsort(perm)
    Vh = R.V.conj().T

    def abb_mul(X, trans=False):
        X1, X2 = X[:nb1], X[nb1:]
        if not trans:
            return np.concatenate([c1.diag_mul(1, X1) + C12b @ X2, C21b @ X1 + c2.diag_mul(1, X2)], axis=0)
        return np.concatenate([c1.diag_mul(1, X1, True) + C21b.T @ X2, C12b.T @ X1 + c2.diag_mul(1, X2, True)], axis=0)

    def abb_get(I, J):
        out = np.zeros((len(I), len(J)), dtype=dt)
        i1, i2, j1, j2 = I < nb1, I >= nb1, J < nb1, J >= nb1
        if i1.any() and j1.any():
            out[np.ix_(i1, j1)] = c1.diag_get(1, I[i1], J[j1])
        if i2.any() and j2.any():
            out[np.ix_(i2, j2)] = c2.diag_get(1, I[i2] - nb1, J[j2] - nb1)
        if i1.any() and j2.any():
            out[np.ix_(i1, j2)] = C12b[I[i1]][:, J[j2] - nb1].toarray()
        if i2.any() and j1.any():
            out[np.ix_(i2, j1)] = C21b[I[i2] - nb1][:, J[j1]].toarray()
        return out

    def mul(X):  # y[iperm] = Abb x[iperm] - U.U (U.V' x[iperm])
        Xo = X[ip]
        return (abb_mul(Xo) - QL @ (W @ (Vh @ Xo)))[perm]

    def mulT(X):
        Xo = X[ip]
        return (abb_mul(Xo, True) - Vh.T @ (W.T @ (QL.T @ Xo)))[perm]

    def get(I, J):
        Io, Jo = perm[I], perm[J]
        return abb_get(Io, Jo) - (QL[Io] @ W) @ Vh[:, Jo]

    op = Op(nb, dt, mul, mulT, get)
    if n1 > 0 and nb > opts.leafsize and flag:
        S = _compress(op, opts, first_split=n1 if n1 < nb else None)  # randcompress_adaptive over bisection_cluster((n1, nb))
    else:
        S = op.mul(np.eye(nb, dtype=dt))
    return Node(D, Abi_lr, R, S, nd.int, nd.bnd, Fl, Fr, "mf", n1)


def ldiv(F, B):
    B = np.asarray(B)
    vec = B.ndim == 1
    C = np.array(B.reshape(len(B), -1), dtype=np.result_type(B.dtype, np.float64))
    if np.iscomplexobj(C) is False and _is_complex(F):
        C = C.astype(np.complex128)
    keep = {}
    _fwd(F, C, keep)
    _bwd(F, C, keep)
    return C[:, 0] if vec else C


def _is_complex(F):
    while F.left is not None:
        F = F.left
    return np.iscomplexobj(F.D)


def _fwd(F, rhs, keep):
    for c in (F.left, F.right):
        if c is not None:
            _fwd(c, rhs, keep)
    i, b = F.int - 1, F.bnd - 1
    if F.kind == "mf":
        t = F.D.solve(rhs[i])
        keep[id(F)] = t
        if len(b):
            rhs[b] = rhs[b] - F.L.U @ (F.L.V.conj().T @ t)
    elif len(b):
        rhs[b] = rhs[b] - O._dense(F.L) @ rhs[i]


def _bwd(F, rhs, keep):
    i, b = F.int - 1, F.bnd - 1
    if F.kind == "mf":
        rhs[i] = keep[id(F)] - (F.R.U @ (F.R.V.conj().T @ rhs[b]) if len(b) else 0)
    else:
        d = O.blockldiv_inplace(F.D, rhs[i]) if isinstance(F.D, O.BlockFactorization) else O._ldiv(F.D, rhs[i])
        rhs[i] = d - (O._dense(F.R) @ rhs[b] if len(b) else 0)
    for c in (F.left, F.right):
        if c is not None:
            _bwd(c, rhs, keep)


def maxrank(F):
    """max over the tree of hssrank(S), hssrank of the HSS blocks of D, rank(L), rank(R) (factornode.jl:49-57)."""
    r = 0
    for c in (F.left, F.right):
        if c is not None:
            r = max(r, maxrank(c))
    if isinstance(F.S, HS.Hss):
        r = max(r, HS.hssrank(F.S))
    if isinstance(F.D, (BlockD, SingleD)):
        r = max(r, F.D.hssrank)
    for M in (F.L, F.R):
        if isinstance(M, OL.LowRankMatrix):
            r = max(r, M.rank)
    return r


def count_kinds(F):
    out = {}

    def walk(x):
        out[x.kind] = out.get(x.kind, 0) + 1
        for c in (x.left, x.right):
            if c is not None:
                walk(c)

    walk(F)
    return out
