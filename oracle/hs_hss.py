"""Oracle for the HSS arithmetic of the compressed branch (TEST INFRASTRUCTURE -- never on the product path).

The reference keeps the Schur complement `S` of a compressed front and its interior block `D = Aii` as
`HssMatrix` objects of HssMatrices.jl 0.1.2 (src/factorization.jl:56-57 `compress(S[perm,perm], cl, cl)`,
:109-110 `randcompress_adaptive(hssS, cl, cl; kest, atol, rtol)`, src/blockmatrix.jl:121-130 `A11\\A12`,
src/factornode.jl:53 `hssrank`).  That package is NOT part of /root/reference (Manifest.toml:263-269 pins it,
the sources are absent), so its arithmetic is restated here from the published algorithms -- PARITY UNPINNED:

* `bisection_cluster` -- binary cluster tree over an index range, leaves of at most `leafsize` indices; the
  tuple form `bisection_cluster((n1, n))` forces the first split at `n1` (src/factorization.jl:56,109 rely on it
  so that the top-level split of S is [int | bnd]);
* `compress` / `randcompress_adaptive` -- randomized HSS compression with nested INTERPOLATIVE bases from
  matrix-vector samples and entry access (P.G. Martinsson, "A fast randomized algorithm for computing a
  hierarchically semiseparable representation of a matrix", SIMAX 32 (2011), Alg. 4), the adaptive variant
  doubling the number of samples until every rank stays below it;
* `hss_matvec`, `hss_full`, `hssrank`;
* `rs_factor` / `rs_solve` -- the ID-based ULV-type elimination ("recursive skeletonization": Martinsson &
  Rokhlin, J. Comput. Phys. 205 (2005); Ho & Greengard, SISC 34 (2012)), the role of HSS `\\` in
  `blockfactor` / `blockldiv!` (src/blockmatrix.jl:121-156).

Representation (what the device module hs_hss.hip builds, field for field).  One skeleton per node serves rows
AND columns: the row ID of the two sample blocks side by side, [A(I,far)*Omega | A(far,I)^T*Psi], gives local
positions p = [p_S ; p_R] (r skeleton positions first) and T ((m-r) x r) with

    A(I[p_R], far) ~= T * A(I[p_S], far)         A(far, I[p_R]) ~= A(far, I[p_S]) * T^T   (plain transpose)

so U_i = V_i = P^T [I; T], the sibling couplings are SUBMATRICES of A (B12 = A(sk_l, sk_r), B21 = A(sk_r, sk_l))
and the nested bases of a parent act on [skeleton of left child; skeleton of right child].
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla


class HssNode:
    __slots__ = ("lo", "hi", "left", "right", "parent", "level", "p", "r", "T", "D", "B12", "B21", "sk", "m")

    def __init__(self, lo, hi, level, parent=-1):
        self.lo, self.hi, self.level, self.parent = lo, hi, level, parent
        self.left = self.right = -1
        self.p = None  # local positions, skeleton first
        self.r = 0
        self.T = None
        self.D = None
        self.B12 = self.B21 = None
        self.sk = None  # global indices of the skeleton
        self.m = 0  # local size: hi-lo for a leaf, r_left + r_right otherwise


class Hss:
    def __init__(self, n, nodes, dtype):
        self.n, self.nodes, self.dtype = n, nodes, dtype

    @property
    def nlevels(self):
        return 1 + max(x.level for x in self.nodes)

    def level(self, lv):
        return [i for i, x in enumerate(self.nodes) if x.level == lv]

    def isleaf(self, i):
        return self.nodes[i].left < 0


def bisection_cluster(n, leafsize=64, first_split=None):
    """Nodes in breadth-first order (node 0 = root); a node with more than `leafsize` indices is halved
    (the left half gets the extra index), the root at `first_split` when given."""
    nodes = [HssNode(0, n, 0)]
    q = [0]
    while q:
        i = q.pop(0)
        x = nodes[i]
        sz = x.hi - x.lo
        forced = first_split is not None and i == 0 and 0 < first_split < n
        if sz <= leafsize and not forced:
            continue
        mid = first_split if forced else x.lo + (sz + 1) // 2
        x.left = len(nodes)
        nodes.append(HssNode(x.lo, mid, x.level + 1, i))
        x.right = len(nodes)
        nodes.append(HssNode(mid, x.hi, x.level + 1, i))
        q += [x.left, x.right]
    return nodes


def row_id(M, atol, rtol, top=None):
    """Row interpolative decomposition M[p[r:]] ~= T M[p[:r]] by a column-pivoted QR of M^T truncated at
    |R_kk| <= max(atol, rtol |R_11|) (the stopping rule of hs_oracle_lr.pqrfact).  `top` (a one-element list) receives |R_11|."""
    m = M.shape[0]
    if m == 0 or M.shape[1] == 0:
        return np.arange(m), 0, np.zeros((m, 0), M.dtype)
    _, R, p = sla.qr(M.T, mode="economic", pivoting=True, check_finite=False)  # plain transpose on purpose
    d = np.abs(np.diag(R))
    if top is not None:
        top[0] = max(top[0], float(d[0]))
    tau = max(atol, rtol * d[0])
    r = int(np.sum(d > tau))
    if r == 0:
        return p, 0, np.zeros((m, 0), M.dtype)
    T = sla.solve_triangular(R[:r, :r], R[:r, r:], check_finite=False).T  # (m-r) x r
    return p, r, T


def compress(A, leafsize=64, atol=1e-9, rtol=1e-9, kest=32, first_split=None, seed=123, pad=8, mul=None, mulT=None, level_scale=1.0, fill=1.0):
    """`randcompress_adaptive`: HSS form of the n x n operator A.  `A` is indexable (`A[np.ix_(i, j)]`); the
    samples come from `mul(X) = A X` and `mulT(X) = A^T X` (plain transpose) when given, else from A itself.
    The sample count doubles until every node's rank is at most k - pad.  `level_scale` < 1 tightens the tolerance of
    tree level l by level_scale^(l-1), `fill` < 1 trusts k samples only up to rank fill*k - pad (what the device module does,
    include/hs_hss.h: 0.5 and 0.8)."""
    n = A.shape[0]
    dtype = np.result_type(A.dtype, np.float64)
    rng = np.random.default_rng(seed)
    k = max(int(kest), 8)
    while True:
        H = _compress_fixed(A, n, dtype, leafsize, atol, rtol, k, first_split, rng, pad, mul, mulT, level_scale, fill)
        if H is not None:
            return H
        k *= 2


def _randn(rng, shape, dtype):
    X = rng.standard_normal(shape)
    if np.issubdtype(dtype, np.complexfloating):
        X = X + 1j * rng.standard_normal(shape)
    return X.astype(dtype)


def _compress_fixed(A, n, dtype, leafsize, atol, rtol, k, first_split, rng, pad, mul, mulT, level_scale=1.0, fill=1.0):
    nodes = bisection_cluster(n, leafsize, first_split)
    H = Hss(n, nodes, dtype)
    k = min(k, n)
    Om, Ps = _randn(rng, (n, k), dtype), _randn(rng, (n, k), dtype)
    Yr = mul(Om) if mul is not None else A @ Om
    Yc = mulT(Ps) if mulT is not None else A.T @ Ps
    Sr, Sc, Ot, Pt = {}, {}, {}, {}  # per node: samples on the skeleton rows, compressed test matrices
    gscale = 0.0  # largest |R_11| of the levels below: floor of the scale the relative tolerance refers to (a block that is
    # only the noise of its children's truncation has full rank relative to ITSELF); what the device module does
    for lv in range(H.nlevels - 1, 0, -1):
        tops = [0.0]
        for i in H.level(lv):
            x = nodes[i]
            if x.left < 0:
                J = np.arange(x.lo, x.hi)
                x.D = np.array(A[np.ix_(J, J)], dtype=dtype)
                sr = Yr[J] - x.D @ Om[J]
                sc = Yc[J] - x.D.T @ Ps[J]
                ol, pl = Om[J], Ps[J]
            else:
                l, rr = nodes[x.left], nodes[x.right]
                J = np.concatenate([l.sk, rr.sk])
                x.B12 = np.array(A[np.ix_(l.sk, rr.sk)], dtype=dtype)
                x.B21 = np.array(A[np.ix_(rr.sk, l.sk)], dtype=dtype)
                sr = np.vstack([Sr[x.left] - x.B12 @ Ot[x.right], Sr[x.right] - x.B21 @ Ot[x.left]])
                sc = np.vstack([Sc[x.left] - x.B21.T @ Pt[x.right], Sc[x.right] - x.B12.T @ Pt[x.left]])
                ol, pl = np.vstack([Ot[x.left], Ot[x.right]]), np.vstack([Pt[x.left], Pt[x.right]])
            x.m = len(J)
            sc_ = level_scale ** (lv - 1)
            p, r, T = row_id(np.hstack([sr, sc]), max(atol, rtol * gscale) * sc_, rtol * sc_, tops)
            if r == 0 and x.m > 0:  # no coupling at all: one nominal skeleton position, T = 0
                p, r, T = np.arange(x.m), 1, np.zeros((x.m - 1, 1), dtype)
            if r > int(fill * k) - pad and r < x.m and k < n:
                return None  # not enough samples for this rank: the caller doubles k
            x.p, x.r, x.T = p, r, T
            x.sk = J[p[:r]]
            Sr[i], Sc[i] = sr[p[:r]], sc[p[:r]]
            Ot[i] = ol[p[:r]] + T.T @ ol[p[r:]]
            Pt[i] = pl[p[:r]] + T.T @ pl[p[r:]]
        gscale = max(gscale, tops[0])
    x = nodes[0]
    if x.left < 0:
        x.D = np.array(A[np.ix_(np.arange(n), np.arange(n))], dtype=dtype)
        x.m = n
    else:
        l, rr = nodes[x.left], nodes[x.right]
        x.B12 = np.array(A[np.ix_(l.sk, rr.sk)], dtype=dtype)
        x.B21 = np.array(A[np.ix_(rr.sk, l.sk)], dtype=dtype)
        x.m = l.r + rr.r
    return H


def hssrank(H):
    """Largest rank of an off-diagonal block (HssMatrices.hssrank; src/factornode.jl:53)."""
    return max([x.r for x in H.nodes[1:]] + [0])


def _restrict(x, v):  # V_i^T v  with V_i = P^T [I; T]
    return v[x.p[: x.r]] + x.T.T @ v[x.p[x.r :]]


def _expand(x, g):  # U_i g
    u = np.empty((x.m,) + g.shape[1:], dtype=np.result_type(g.dtype, x.T.dtype))
    u[x.p[: x.r]] = g
    u[x.p[x.r :]] = x.T @ g
    return u


def hss_matvec(H, X):
    """Y = H X (X: n or n x q): upward pass V^T x, sibling couplings, downward pass."""
    X2 = X.reshape(H.n, -1).astype(np.result_type(H.dtype, X.dtype))
    nodes = H.nodes
    xt = {}
    for lv in range(H.nlevels - 1, 0, -1):
        for i in H.level(lv):
            x = nodes[i]
            loc = X2[x.lo : x.hi] if x.left < 0 else np.vstack([xt[x.left], xt[x.right]])
            xt[i] = _restrict(x, loc)
    Y = np.zeros_like(X2)
    g = {}
    for lv in range(0, H.nlevels):
        for i in H.level(lv):
            x = nodes[i]
            if x.left < 0:
                Y[x.lo : x.hi] = x.D @ X2[x.lo : x.hi] + (_expand(x, g[i]) if i in g else 0)
                continue
            l, rr = nodes[x.left], nodes[x.right]
            u = _expand(x, g[i]) if i in g else np.zeros((l.r + rr.r, X2.shape[1]), X2.dtype)
            g[x.left] = u[: l.r] + x.B12 @ xt[x.right]
            g[x.right] = u[l.r :] + x.B21 @ xt[x.left]
    return Y.reshape(X.shape)


def hss_full(H):
    return hss_matvec(H, np.eye(H.n, dtype=H.dtype))


class RsFactor:
    """Per node: LU of X_RR, the eliminated couplings X_SR X_RR^-1 and X_RR^-1 X_RS; root: LU of the last block."""

    def __init__(self, H):
        self.H = H
        self.lu = {}
        self.Lc = {}  # X_SR * X_RR^-1   (r x (m-r))
        self.Rc = {}  # X_RR^-1 * X_RS   ((m-r) x r)
        self.root_lu = None


def rs_factor(H):
    nodes = H.nodes
    F = RsFactor(H)
    Sh = {}  # Schur complement of a node on its skeleton
    for lv in range(H.nlevels - 1, 0, -1):
        for i in H.level(lv):
            x = nodes[i]
            M = x.D if x.left < 0 else np.block([[Sh[x.left], x.B12], [x.B21, Sh[x.right]]])
            pS, pR, T = x.p[: x.r], x.p[x.r :], x.T
            MRR, MRS, MSR, MSS = M[np.ix_(pR, pR)], M[np.ix_(pR, pS)], M[np.ix_(pS, pR)], M[np.ix_(pS, pS)]
            XRS = MRS - T @ MSS
            XSR = MSR - MSS @ T.T
            XRR = MRR - T @ MSR - XRS @ T.T  # = MRR - T MSR - MRS T^T + T MSS T^T
            if XRR.shape[0] > 0:
                lu = sla.lu_factor(XRR, check_finite=False)
                F.lu[i] = lu
                F.Rc[i] = sla.lu_solve(lu, XRS, check_finite=False)
                F.Lc[i] = sla.lu_solve(lu, XSR.T, trans=1, check_finite=False).T
                Sh[i] = MSS - XSR @ F.Rc[i]
            else:
                F.lu[i] = None
                F.Rc[i] = np.zeros((0, x.r), H.dtype)
                F.Lc[i] = np.zeros((x.r, 0), H.dtype)
                Sh[i] = MSS
    x = nodes[0]
    M = x.D if x.left < 0 else np.block([[Sh[x.left], x.B12], [x.B21, Sh[x.right]]])
    F.root_lu = sla.lu_factor(M, check_finite=False)
    return F


def rs_solve(F, B):
    """X = H^-1 B through the skeletonization factors (forward: leaves to root, backward: root to leaves)."""
    H, nodes = F.H, F.H.nodes
    B2 = B.reshape(H.n, -1).astype(np.result_type(H.dtype, B.dtype))
    bh, zR = {}, {}
    for lv in range(H.nlevels - 1, 0, -1):
        for i in H.level(lv):
            x = nodes[i]
            loc = B2[x.lo : x.hi] if x.left < 0 else np.vstack([bh[x.left], bh[x.right]])
            bS, bR = loc[x.p[: x.r]], loc[x.p[x.r :]]
            bR = bR - x.T @ bS  # E = [I -T; 0 I]
            zR[i] = sla.lu_solve(F.lu[i], bR, check_finite=False) if F.lu[i] is not None else bR
            bh[i] = bS - (F.Lc[i] @ bR if bR.shape[0] else 0)  # b_S - X_SR X_RR^-1 b_R
    x = nodes[0]
    loc = B2 if x.left < 0 else np.vstack([bh[x.left], bh[x.right]])
    xs = {0: sla.lu_solve(F.root_lu, loc, check_finite=False)}
    X = np.zeros_like(B2)
    for lv in range(0, H.nlevels):
        for i in H.level(lv):
            x = nodes[i]
            if i == 0:
                loc = xs[0]
            else:
                xS = xs[i]
                xR = zR[i] - F.Rc[i] @ xS
                loc = np.empty((x.m, B2.shape[1]), B2.dtype)
                loc[x.p[x.r :]] = xR
                loc[x.p[: x.r]] = xS - x.T.T @ xR  # F = [I 0; -T^T I]
            if x.left < 0:
                X[x.lo : x.hi] = loc
            else:
                rl = nodes[x.left].r
                xs[x.left], xs[x.right] = loc[:rl], loc[rl:]
    return X.reshape(B.shape)


# --------------------------------------------------------------------------------------------------------------
# access to parts of an HSS matrix: what the matrix-free assembly of a parent front needs from its children's S
# (src/factorization.jl:126-140 reads `S.A11`, `S.A22`, `generators(S.A11)`, `S.B12`, `S.B21`)
# --------------------------------------------------------------------------------------------------------------
def _local_basis(x):
    """U_i = P^T [I; T] ((m) x r) of a non-root node."""
    U = np.zeros((x.m, x.r), dtype=x.T.dtype)
    U[x.p[: x.r]] = np.eye(x.r, dtype=x.T.dtype)
    U[x.p[x.r :]] = x.T
    return U


def basis_rows(H, i, idx):
    """Rows `idx` (global indices inside node i's range, ascending or not) of the EXPANDED nested basis of node i."""
    x = H.nodes[i]
    idx = np.asarray(idx, dtype=np.int64)
    if x.left < 0:
        return _local_basis(x)[idx - x.lo]
    l, r = H.nodes[x.left], H.nodes[x.right]
    inl = idx < l.hi
    W = np.zeros((len(idx), l.r + r.r), dtype=H.dtype)
    if inl.any():
        W[inl, : l.r] = basis_rows(H, x.left, idx[inl])
    if (~inl).any():
        W[~inl, l.r :] = basis_rows(H, x.right, idx[~inl])
    return W @ _local_basis(x) if i != 0 else W


def hss_getindex(H, I, J):
    """H[I, J] for index arrays (0-based): the entry access `randcompress_adaptive` asks of its operator."""
    I, J = np.asarray(I, dtype=np.int64), np.asarray(J, dtype=np.int64)
    out = np.zeros((len(I), len(J)), dtype=H.dtype)

    def rec(i, ri, rj):  # ri / rj: positions into I / J that fall into node i's range
        x = H.nodes[i]
        if len(ri) == 0 or len(rj) == 0:
            return
        if x.left < 0:
            out[np.ix_(ri, rj)] = x.D[np.ix_(I[ri] - x.lo, J[rj] - x.lo)]
            return
        mid = H.nodes[x.left].hi
        il, ir = ri[I[ri] < mid], ri[I[ri] >= mid]
        jl, jr = rj[J[rj] < mid], rj[J[rj] >= mid]
        rec(x.left, il, jl)
        rec(x.right, ir, jr)
        if len(il) and len(jr):
            out[np.ix_(il, jr)] = basis_rows(H, x.left, I[il]) @ x.B12 @ basis_rows(H, x.right, J[jr]).T
        if len(ir) and len(jl):
            out[np.ix_(ir, jl)] = basis_rows(H, x.right, I[ir]) @ x.B21 @ basis_rows(H, x.left, J[jl]).T

    rec(0, np.arange(len(I)), np.arange(len(J)))
    return out


def hss_child(H, which):
    """`H.A11` (which = 0) or `H.A22` (which = 1): the diagonal block of the top-level split as an HSS matrix of its own."""
    root = H.nodes[0]
    if root.left < 0:
        raise ValueError("One of the Schur complements turned into a leaf. Aborting.")  # factorization.jl:164
    top = root.left if which == 0 else root.right
    off = H.nodes[top].lo
    ids, stack = [], [top]
    while stack:  # breadth-first renumbering of the subtree
        nxt = []
        for i in stack:
            ids.append(i)
            if H.nodes[i].left >= 0:
                nxt += [H.nodes[i].left, H.nodes[i].right]
        stack = nxt
    new = {old: k for k, old in enumerate(ids)}
    nodes = []
    for old in ids:
        x = H.nodes[old]
        y = HssNode(x.lo - off, x.hi - off, x.level - 1, new.get(x.parent, -1))
        y.left, y.right = (new[x.left], new[x.right]) if x.left >= 0 else (-1, -1)
        y.m, y.D, y.B12, y.B21 = x.m, x.D, x.B12, x.B21
        if old != top:
            y.p, y.r, y.T = x.p, x.r, x.T
            y.sk = None if x.sk is None else x.sk - off
        nodes.append(y)
    return Hss(H.nodes[top].hi - off, nodes, H.dtype)


def hss_offdiag(H):
    """The two off-diagonal blocks of the top-level split in low-rank form: A12 = U1 B12 V2^T, A21 = U2 B21 V1^T with the
    expanded bases (U = V here).  Returns (U1, B12, U2, B21)."""
    root = H.nodes[0]
    l, r = H.nodes[root.left], H.nodes[root.right]
    U1 = basis_rows(H, root.left, np.arange(l.lo, l.hi))
    U2 = basis_rows(H, root.right, np.arange(r.lo, r.hi))
    return U1, root.B12, U2, root.B21


_PACK_FIELDS = ("p", "T", "sk", "D", "B12", "B21")


def hss_pack(H):
    """(int64 header, flat payload of H.dtype): the generators of an HSS matrix as two arrays -- what crosses ranks when a child's Schur
    complement travels as an HssMatrix (the device counterpart is hs_hss_pack, include/hs_hss.h).  Index arrays ride in the
    payload as real numbers (exact below 2^53)."""
    ints, payload = [int(H.n), len(H.nodes), int(np.dtype(H.dtype) == np.complex128)], []
    for x in H.nodes:
        ints += [x.lo, x.hi, x.level, x.parent, x.left, x.right, x.m, x.r]
        for f in _PACK_FIELDS:
            a = getattr(x, f, None)
            if a is None:
                ints += [-1, 0, 0]
                continue
            a = np.asarray(a)
            ints += [a.ndim, a.shape[0], a.shape[1] if a.ndim == 2 else 1]
            payload.append(a.astype(H.dtype).ravel())
    flat = np.concatenate(payload) if payload else np.zeros(0, dtype=H.dtype)
    return np.asarray(ints, dtype=np.int64), np.ascontiguousarray(flat)


def hss_unpack(ints, flat):
    ints = [int(v) for v in ints]
    n, nn, is_c = ints[:3]
    dtype = np.complex128 if is_c else np.float64
    flat = np.asarray(flat, dtype=dtype)
    at, pos, nodes = 3, 0, []
    for _ in range(nn):
        lo, hi, level, parent, left, right, m, r = ints[at:at + 8]
        at += 8
        x = HssNode(lo, hi, level, parent)
        x.left, x.right, x.m, x.r = left, right, m, r
        for f in _PACK_FIELDS:
            nd_, rows, cols = ints[at:at + 3]
            at += 3
            if nd_ < 0:
                setattr(x, f, None)
                continue
            a = flat[pos:pos + rows * cols]
            pos += rows * cols
            a = a.reshape(rows, cols) if nd_ == 2 else a.reshape(rows)
            if f in ("p", "sk"):
                a = np.real(a).astype(np.int64)
            setattr(x, f, a.copy())
        nodes.append(x)
    assert pos == len(flat) and at == len(ints)
    return Hss(n, nodes, dtype)
