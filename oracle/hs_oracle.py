"""NumPy/SciPy restatement of HierarchicalSolvers.jl's nested-dissection elimination.

TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  The product path
(``hierarchicalsolvers.jl_amd``) never imports this module.

Parity pinning
--------------
PARITY UNPINNED against the Julia reference: the reference ships no golden
vectors, no ``@test`` and its four ``.mat`` inputs are stripped
(``/root/reference/.MISSING_LARGE_BLOBS:1-4``); there is no Julia runtime in
the build container, so reference outputs cannot be generated either
(SURVEY.md section 8(c)).  What pins this restatement instead:

* the dense path (``swlevel=0``) is *exact*: ``ldiv!(F, b)`` must equal
  ``A \\ b``.  ``tests/test_oracle.py`` checks it against SuperLU
  (``scipy.sparse.linalg.splu``), an independent sparse direct solver --
  the restatement of the reference's own commented check
  ``norm(A*xa-b)/norm(A\\b)`` (``test/rungmres.jl:36``);
* per-node algebraic identities ``L*D = A_bi``, ``D*R = A_ib``,
  ``S = A_bb - A_bi*R`` on the assembled front;
* ``blockldiv!(blockfactor(M), M*X) == X``.

Every function cites the reference file:line it follows.  Index vectors are
kept 1-based ``int64`` exactly as Julia holds them and shifted only at the
point of use.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla
import scipy.sparse as sp

# --------------------------------------------------------------------------
# SolverOptions  (src/HierarchicalSolvers.jl:30-79)
# --------------------------------------------------------------------------


class SolverOptions:
    """Mirror of ``mutable struct SolverOptions`` (HierarchicalSolvers.jl:30-40)
    with the defaults of the kw-constructor (``:43-59``)."""

    _fields = ("swlevel", "swsize", "atol", "rtol", "c_tol", "leafsize", "kest", "stepsize", "verbose")

    def __init__(self, **kw):
        self.swlevel = 5
        self.swsize = 1
        self.atol = 1e-6
        self.rtol = 1e-6
        self.c_tol = 0.5
        self.leafsize = 32
        self.kest = -1
        self.stepsize = 10
        self.verbose = False
        for k, v in kw.items():
            if k not in self._fields:  # setfield! on an unknown field throws in Julia
                raise TypeError(f"type SolverOptions has no field {k}")
            setattr(self, k, v)

    def copy(self, **kw):  # HierarchicalSolvers.jl:62-71
        o = SolverOptions()
        for f in self._fields:
            setattr(o, f, getattr(self, f))
        for k, v in kw.items():
            if k not in self._fields:
                raise TypeError(f"type SolverOptions has no field {k}")
            setattr(o, k, v)
        return o


def chkopts(opts: SolverOptions):  # HierarchicalSolvers.jl:73-79 (ArgumentError -> ValueError)
    if not opts.swsize >= 1:
        raise ValueError("swsize")
    if not opts.atol >= 0.0:
        raise ValueError("atol")
    if not opts.rtol >= 0.0:
        raise ValueError("rtol")
    if not (0.0 < opts.c_tol <= 1.0):
        raise ValueError("c_tol")
    if not opts.leafsize >= 1:
        raise ValueError("leafsize")


# --------------------------------------------------------------------------
# NestedDissection tree  (src/nesteddissection.jl)
# --------------------------------------------------------------------------


def _ivec(x):
    return np.asarray(x, dtype=np.int64).reshape(-1).copy()


class NDNode:
    """``BinaryNode{Tuple{int,bnd}}`` with ``.int/.bnd`` sugar (nesteddissection.jl:8-21)."""

    def __init__(self, int_, bnd, left=None, right=None):
        self.int = _ivec(int_)
        self.bnd = _ivec(bnd)
        self.left = left
        self.right = right


def isleaf(nd):
    return nd.left is None and nd.right is None


def isbranch(nd):
    return nd.left is not None and nd.right is not None


def depth(nd):  # HssMatrices.depth on BinaryNode: a lone leaf has depth 1
    if nd is None:
        return 0
    return 1 + max(depth(nd.left), depth(nd.right))


def _findall_in(haystack, values):
    """``findall(in(haystack), values)``: 1-based positions p with values[p] in haystack."""
    mask = np.isin(values, haystack)
    return np.nonzero(mask)[0].astype(np.int64) + 1


def symfact(nd):
    """``symfact!`` (nesteddissection.jl:29-34).  Mutates ``nd``; returns (nd, nd_loc)."""
    nd_loc = _symfact(nd, 1)
    nd_loc.int = np.arange(1, len(nd.bnd) + 1, dtype=np.int64)
    nd_loc.bnd = np.zeros(0, dtype=np.int64)
    return nd, nd_loc


def _symfact(nd, level):  # nesteddissection.jl:35-69
    if isleaf(nd):
        return NDNode([], [])
    if nd.left is not None:
        left_loc = _symfact(nd.left, level + 1)
        left_loc.int = _findall_in(nd.int, nd.left.bnd)
        left_loc.bnd = _findall_in(nd.bnd, nd.left.bnd)
        intl = nd.left.bnd[left_loc.int - 1]
        bndl = nd.left.bnd[left_loc.bnd - 1]
    else:
        intl = np.zeros(0, np.int64)
        bndl = np.zeros(0, np.int64)
        left_loc = None
    if nd.right is not None:
        right_loc = _symfact(nd.right, level + 1)
        right_loc.int = _findall_in(nd.int, nd.right.bnd)
        right_loc.bnd = _findall_in(nd.bnd, nd.right.bnd)
        intr = nd.right.bnd[right_loc.int - 1]
        bndr = nd.right.bnd[right_loc.bnd - 1]
    else:
        intr = np.zeros(0, np.int64)
        bndr = np.zeros(0, np.int64)
        right_loc = None
    nd.int = np.concatenate([intl, intr])
    nd.bnd = np.concatenate([bndl, bndr])
    return NDNode([], [], left_loc, right_loc)


def _postorder_nodes(nd, out):
    if nd.left is not None:
        _postorder_nodes(nd.left, out)
    if nd.right is not None:
        _postorder_nodes(nd.right, out)
    out.append(nd)
    return out


def postorder(nd):  # nesteddissection.jl:73-79
    parts = [x.int for x in _postorder_nodes(nd, [])]
    parts.append(nd.bnd)
    return np.concatenate(parts) if parts else np.zeros(0, np.int64)


def invperm(p):
    p = np.asarray(p, dtype=np.int64)
    ip = np.empty_like(p)
    ip[p - 1] = np.arange(1, len(p) + 1, dtype=np.int64)
    return ip


def permuted(nd, perm):  # permuted! (nesteddissection.jl:82-88)
    perm = np.asarray(perm, dtype=np.int64)
    if nd.left is not None:
        nd.left = permuted(nd.left, perm)
    if nd.right is not None:
        nd.right = permuted(nd.right, perm)
    nd.int = perm[nd.int - 1]
    nd.bnd = perm[nd.bnd - 1]
    return nd


def contigious(idx):  # nesteddissection.jl:91 (returns (lo, hi) for a unit range, else idx)
    idx = np.asarray(idx, dtype=np.int64)
    if len(idx) and np.array_equal(np.arange(idx[0], idx[-1] + 1), idx):
        return (int(idx[0]), int(idx[-1]))
    return idx


def parse_elimtree(fathers, lsons, rsons, ninter, inter, nbound, bound):
    """nesteddissection.jl:105-148, stack machine restated literally (1-based ids, -1 = none)."""
    fathers = _ivec(fathers)
    lsons = _ivec(lsons)
    rsons = _ivec(rsons)
    ninter = _ivec(ninter)
    nbound = _ivec(nbound)
    inter = np.asarray(inter, dtype=np.int64)
    bound = np.asarray(bound, dtype=np.int64)
    nnodes = len(fathers)
    if not (nnodes == len(lsons) == len(rsons) == len(ninter) == len(nbound) == inter.shape[1] == bound.shape[1]):
        raise ValueError("DimensionMismatch: dimensions inconsistent among inputs")  # :107
    roots = np.nonzero(fathers == -1)[0] + 1
    if len(roots) != 1:
        raise ValueError("found either less than or more than one root.")  # :111

    def mk(i, l=None, r=None):
        return NDNode(inter[: ninter[i - 1], i - 1], bound[: nbound[i - 1], i - 1], l, r)

    sind = [int(roots[0])]
    ilast = -2
    snodes = []
    while sind:
        i = sind[-1]
        ls, rs = int(lsons[i - 1]), int(rsons[i - 1])
        if rs == -1 and ls == -1:
            snodes.append(mk(i))
            ilast = sind.pop()
        elif ilast == rs:
            right = snodes.pop()
            left = snodes.pop() if ls != -1 else None
            snodes.append(mk(i, left, right))
            ilast = sind.pop()
        elif ilast == ls and rs == -1:
            left = snodes.pop()
            snodes.append(mk(i, left, None))
            ilast = sind.pop()
        elif (ilast == ls and rs != -1) or (ls == -1):  # `rsons != -1` at :139 is always true
            ilast = i
            sind.append(rs)
        else:
            ilast = i
            sind.append(ls)
    return snodes.pop()


# --------------------------------------------------------------------------
# dense helpers: Julia's `\` and `/` on square dense matrices are LU with
# partial pivoting (LinearAlgebra generic `\` -> lu(A) \ B).
# --------------------------------------------------------------------------


def _ldiv(A, B):  # A \ B
    if A.shape[0] == 0:
        return np.zeros((0,) + B.shape[1:], dtype=np.result_type(A, B))
    return sla.solve(A, B, check_finite=False)


def _rdiv(A, B):  # A / B  == (B' \ A')'
    if B.shape[0] == 0:
        return np.zeros((A.shape[0], 0), dtype=np.result_type(A, B))
    return sla.solve(B.T, A.T, check_finite=False).T


def _gather(A, I, J):
    """``Matrix(view(A, I, J))`` for CSC/CSR ``A`` and 1-based index vectors (factorization.jl:33-40)."""
    I = np.asarray(I, dtype=np.int64) - 1
    J = np.asarray(J, dtype=np.int64) - 1
    if len(I) == 0 or len(J) == 0:
        return np.zeros((len(I), len(J)), dtype=A.dtype)
    return np.asarray(A[I][:, J].todense())


# --------------------------------------------------------------------------
# BlockMatrix / BlockFactorization  (src/blockmatrix.jl) -- dense blocks
# --------------------------------------------------------------------------


class BlockMatrix:
    def __init__(self, A11, A12, A21, A22):  # blockmatrix.jl:12-18
        if A11.shape[0] != A12.shape[0]:
            raise ValueError("DimensionMismatch: first dimension of A11 and A12 do not match")
        if A11.shape[1] != A21.shape[1]:
            raise ValueError("DimensionMismatch: second dimension of A11 and A21 do not match")
        if A22.shape[0] != A21.shape[0]:
            raise ValueError("DimensionMismatch: first dimension of A22 and A21 do not match")
        if A22.shape[1] != A12.shape[1]:
            raise ValueError("DimensionMismatch: second dimension of A22 and A12 do not match")
        self.A11, self.A12, self.A21, self.A22 = A11, A12, A21, A22

    @property
    def shape(self):  # blockmatrix.jl:23
        return (self.A11.shape[0] + self.A22.shape[0], self.A11.shape[1] + self.A22.shape[1])

    def dense(self):  # Matrix(B), blockmatrix.jl:67-75
        return np.block([[self.A11, self.A12], [self.A21, self.A22]])

    def matmul_block(self, B):  # blockmatrix.jl:94-98
        A = self
        if A.shape[1] != B.shape[0] or A.A11.shape[1] != B.A11.shape[0]:
            raise ValueError("DimensionMismatch")
        return BlockMatrix(
            A.A11 @ B.A11 + A.A12 @ B.A21,
            A.A11 @ B.A12 + A.A12 @ B.A22,
            A.A21 @ B.A11 + A.A22 @ B.A21,
            A.A21 @ B.A12 + A.A22 @ B.A22,
        )


class BlockFactorization:  # blockmatrix.jl:106-108
    def __init__(self, B):
        self.B = B

    @property
    def shape(self):
        return self.B.shape


def blockfactor(A):  # blockmatrix.jl:115-120
    if A.A11.shape[0] != A.A11.shape[1]:
        raise ValueError("DimensionMismatch: First block of A is not square.")
    if A.A22.shape[0] != A.A22.shape[1]:
        raise ValueError("DimensionMismatch: Second block of A is not square.")
    S22 = A.A22 - A.A21 @ _ldiv(A.A11, A.A12)
    return BlockFactorization(BlockMatrix(A.A11, A.A12, A.A21, S22))


def blockldiv_inplace(F, B):  # blockldiv! (blockmatrix.jl:134-144); returns a new Y like the reference
    A = F.B
    n1 = A.A11.shape[1]
    Y = np.empty_like(B, dtype=np.result_type(A.A11, B))
    Y[:n1] = _ldiv(A.A11, B[:n1])
    Y[n1:] = B[n1:] - A.A21 @ Y[:n1]
    Y[n1:] = _ldiv(A.A22, Y[n1:])
    Y[:n1] = Y[:n1] - _ldiv(A.A11, A.A12 @ Y[n1:])
    return Y


def blockrdiv_inplace(Amat, F):  # blockrdiv! (blockmatrix.jl:146-156)
    B = F.B
    m1 = B.A11.shape[0]
    Y = np.empty_like(Amat, dtype=np.result_type(B.A11, Amat))
    Y[:, :m1] = _rdiv(Amat[:, :m1], B.A11)
    Y[:, m1:] = Amat[:, m1:] - Y[:, :m1] @ B.A12
    Y[:, m1:] = _rdiv(Y[:, m1:], B.A22)
    Y[:, :m1] = Y[:, :m1] - _rdiv(Y[:, m1:] @ B.A21, B.A11)
    return Y


def blockldiv(F, B):  # blockmatrix.jl:159-172
    A = F.B
    B11 = _ldiv(A.A11, B.A11)
    B21 = B.A21 - A.A21 @ B11
    B21 = _ldiv(A.A22, B21)
    B11 = B11 - _ldiv(A.A11, A.A12 @ B21)
    B12 = _ldiv(A.A11, B.A12)
    B22 = B.A22 - A.A21 @ B12
    B22 = _ldiv(A.A22, B22)
    B12 = B12 - _ldiv(A.A11, A.A12 @ B22)
    return BlockMatrix(B11, B12, B21, B22)


def blockrdiv(B, F):  # blockmatrix.jl:174-187
    A = F.B
    B11 = _rdiv(B.A11, A.A11)
    B12 = B.A12 - B11 @ A.A12
    B12 = _rdiv(B12, A.A22)
    B11 = B11 - _rdiv(B12 @ A.A21, A.A11)
    B21 = _rdiv(B.A21, A.A11)
    B22 = B.A22 - B21 @ A.A12
    B22 = _rdiv(B22, A.A22)
    B21 = B21 - _rdiv(B22 @ A.A21, A.A11)
    return BlockMatrix(B11, B12, B21, B22)


# --------------------------------------------------------------------------
# FactorNode + ldiv!  (src/factornode.jl)
# --------------------------------------------------------------------------


class FactorNode:  # factornode.jl:7-39
    def __init__(self, D, S, L, R, int_, bnd, int_loc, bnd_loc, left=None, right=None):
        self.D, self.S, self.L, self.R = D, S, L, R
        self.int = _ivec(int_)
        self.bnd = _ivec(bnd)
        self.int_loc = _ivec(int_loc)
        self.bnd_loc = _ivec(bnd_loc)
        self.left, self.right = left, right


def _dense(M):
    return M.dense() if hasattr(M, "dense") else M


def maxrank(F):  # factornode.jl:49-57 -- dense path: every rank is 0
    rkl = maxrank(F.left) if F.left is not None else 0
    rkr = maxrank(F.right) if F.right is not None else 0
    rk = max(getattr(F.S, "hssrank", 0), getattr(F.L, "rank", 0), getattr(F.R, "rank", 0))
    return max(rkl, rkr, rk)


def ldiv(F, B):
    """``ldiv!(C, F, B)`` (factornode.jl:62-74).  Returns C; B is left untouched
    (the reference's 2-arg form allocates ``similar(B)``)."""
    B = np.asarray(B)
    vec = B.ndim == 1
    C = np.array(B.reshape(len(B), -1), dtype=np.result_type(B.dtype, _dense(F.L).dtype if F.L is not None else B.dtype))
    _lsolve(F, C)
    _dsolve(F, C)
    if len(F.bnd):
        C[F.bnd - 1] = _ldiv(_dense(F.S), C[F.bnd - 1])  # :72
    _rsolve(F, C)
    return C[:, 0] if vec else C


def _lsolve(F, rhs):  # factornode.jl:77-82
    if F.left is not None:
        _lsolve(F.left, rhs)
    if F.right is not None:
        _lsolve(F.right, rhs)
    rhs[F.bnd - 1] = rhs[F.bnd - 1] - _dense(F.L) @ rhs[F.int - 1]


def _rsolve(F, rhs):  # factornode.jl:83-88
    rhs[F.int - 1] = rhs[F.int - 1] - _dense(F.R) @ rhs[F.bnd - 1]
    if F.left is not None:
        _rsolve(F.left, rhs)
    if F.right is not None:
        _rsolve(F.right, rhs)


def _dsolve(F, rhs):  # factornode.jl:89-99
    if F.left is not None:
        _dsolve(F.left, rhs)
    if F.right is not None:
        _dsolve(F.right, rhs)
    if isinstance(F.D, BlockFactorization):
        rhs[F.int - 1] = blockldiv_inplace(F.D, rhs[F.int - 1])
    else:
        rhs[F.int - 1] = _ldiv(F.D, rhs[F.int - 1])


# --------------------------------------------------------------------------
# factor  (src/factorization.jl) -- dense path
# --------------------------------------------------------------------------


def factor(A, nd, nd_loc, opts=None, **kw):  # factorization.jl:5-11
    opts = (opts or SolverOptions()).copy(**kw)
    chkopts(opts)
    swlevel = max(depth(nd) + opts.swlevel, 0) if opts.swlevel < 0 else opts.swlevel
    A = sp.csc_matrix(A)
    return _factor(A, nd, nd_loc, 1, swlevel=swlevel, opts=opts)


def _factor(A, nd, nd_loc, level, *, swlevel, opts):  # factorization.jl:14-27
    compression_flag = (level <= swlevel) and (len(nd.bnd) >= opts.swsize)
    if isleaf(nd):
        return _factor_leaf(A, nd, nd_loc, compression_flag, opts)
    elif isbranch(nd):
        Fl = _factor(A, nd.left, nd_loc.left, level + 1, swlevel=swlevel, opts=opts)
        Fr = _factor(A, nd.right, nd_loc.right, level + 1, swlevel=swlevel, opts=opts)
        return _factor_branch(A, Fl, Fr, nd, nd_loc, compression_flag, opts)
    raise RuntimeError("Expected nested dissection to be a binary tree. Found a node with only one child.")  # :25


def _factor_leaf(A, nd, nd_loc, compress, opts):  # factorization.jl:30-42
    if compress:
        from . import hs_oracle_hss  # compressed leaf (factorization.jl:45-59)

        return hs_oracle_hss.factor_leaf_compressed(A, nd, nd_loc, opts)
    int_, bnd = nd.int, nd.bnd
    D = _gather(A, int_, int_)
    Abi = _gather(A, bnd, int_)
    L = _rdiv(Abi, D)
    R = _ldiv(D, _gather(A, int_, bnd))
    perm = np.concatenate([nd_loc.int, nd_loc.bnd]) - 1
    S = _gather(A, bnd, bnd) - Abi @ R
    return FactorNode(D, S[np.ix_(perm, perm)], L, R, int_, bnd, nd_loc.int, nd_loc.bnd)


def _assemble_blocks(A, S1, S2, int1, int2, bnd1, bnd2):  # factorization.jl:115-123
    ni1, nb1 = len(int1), len(bnd1)
    ni2, nb2 = len(int2), len(bnd2)
    Aii = BlockMatrix(S1[:ni1, :ni1], _gather(A, int1, int2), _gather(A, int2, int1), S2[:ni2, :ni2])
    Aib = BlockMatrix(S1[:ni1, ni1 : ni1 + nb1], _gather(A, int1, bnd2), _gather(A, int2, bnd1), S2[:ni2, ni2 : ni2 + nb2])
    Abi = BlockMatrix(S1[ni1 : ni1 + nb1, :ni1], _gather(A, bnd1, int2), _gather(A, bnd2, int1), S2[ni2 : ni2 + nb2, :ni2])
    Abb = BlockMatrix(
        S1[ni1 : ni1 + nb1, ni1 : ni1 + nb1], _gather(A, bnd1, bnd2), _gather(A, bnd2, bnd1), S2[ni2 : ni2 + nb2, ni2 : ni2 + nb2]
    )
    return Aii, Aib, Abi, Abb


def _factor_branch(A, Fl, Fr, nd, nd_loc, compress, opts):  # factorization.jl:62-75
    if compress:
        from . import hs_oracle_hss  # compressed branch (factorization.jl:78-112)

        return hs_oracle_hss.factor_branch_compressed(A, Fl, Fr, nd, nd_loc, opts)
    int1 = nd.left.bnd[nd_loc.left.int - 1]
    bnd1 = nd.left.bnd[nd_loc.left.bnd - 1]
    int2 = nd.right.bnd[nd_loc.right.int - 1]
    bnd2 = nd.right.bnd[nd_loc.right.bnd - 1]
    Aii, Aib, Abi, Abb = _assemble_blocks(A, _dense(Fl.S), _dense(Fr.S), int1, int2, bnd1, bnd2)
    D = blockfactor(Aii)
    L = blockrdiv(Abi, D)
    R = blockldiv(D, Aib)
    S = Abb.dense() - Abi.matmul_block(R).dense()  # :72 (generic broadcast over scalar getindex)
    perm = np.concatenate([nd_loc.int, nd_loc.bnd]) - 1
    return FactorNode(D, S[np.ix_(perm, perm)], L, R, nd.int, nd.bnd, nd_loc.int, nd_loc.bnd, Fl, Fr)


# --------------------------------------------------------------------------
# flop model used by bench.py / DESIGN.md
# --------------------------------------------------------------------------


def front_flops(ni, nb):
    """Minimal multifrontal LU count F(ni,nb) = 2/3 ni^3 + 2 ni^2 nb + 2 ni nb^2 (real flops, x4 for complex)."""
    return (2.0 / 3.0) * ni**3 + 2.0 * ni * ni * nb + 2.0 * ni * nb * nb


def tree_flops(nd):
    return sum(front_flops(len(x.int), len(x.bnd)) for x in _postorder_nodes(nd, []))


def postorder_nodes(nd):
    return _postorder_nodes(nd, [])
