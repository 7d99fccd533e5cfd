"""HSS matrices on the device -- host-side mirror of the HssMatrices.jl calls on the reference's hot path.

The reference keeps the Schur complement ``S`` and the interior block ``D`` of a compressed front as
``HssMatrix`` objects (``compress`` ``src/factorization.jl:56-57``, ``randcompress_adaptive`` ``:109-110``,
``hssrank`` ``src/factornode.jl:53``, HSS ``\\`` inside ``blockfactor`` ``src/blockmatrix.jl:121-156``).  This
module binds the C ABI of ``include/hs_hss.h``; every operation runs on the GPU (there is no CPU fallback), the
tests compare against a CPU restatement of the same algorithms (``tests/test_hss_gpu.py``).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

__all__ = ["HssMatrix", "compress", "compress_lowrank_update", "randcompress_adaptive", "hssrank", "bisection_cluster", "SparseDevice", "BlockOperator", "equilibrate_clusters"]


def bisection_cluster(n, leafsize=64):
    """``bisection_cluster(n; leafsize)`` / ``bisection_cluster((n1, n); leafsize)``: the (first_split, n, leafsize)
    triple the device tree builder consumes (the tuple form forces the root split at ``n1``, factorization.jl:56,109)."""
    if isinstance(n, tuple):
        n1, n = n
        return int(n1), int(n), int(leafsize)
    return 0, int(n), int(leafsize)


class HssMatrix:
    """Device-resident HSS form of a dense square matrix."""

    def __init__(self, handle, is_complex, parent=None):
        self._h, self.is_complex = handle, bool(is_complex)
        self._parent = parent  # a block view shares its parent's generators: keep the parent alive
        self.dtype = np.complex128 if is_complex else np.float64
        self.L = _lib.lib()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.L.hs_hss_free(h)

    @property
    def shape(self):
        n = int(self.L.hs_hss_size(self._h))
        return (n, n)

    @property
    def rank(self):
        return int(self.L.hs_hss_rank(self._h))

    @property
    def samples(self):
        return int(self.L.hs_hss_samples(self._h))

    @property
    def num_nodes(self):
        return int(self.L.hs_hss_num_nodes(self._h))

    def times(self):
        return {"compress_s": self.L.hs_hss_time(self._h, 0), "factor_s": self.L.hs_hss_time(self._h, 1)}

    def _info(self, i):
        out = (C.c_int64 * 8)()
        _lib.check(self.L.hs_hss_node_info(self._h, i, out))
        lo, hi, left, right, level, m, r, isleaf = [int(v) for v in out]
        return dict(lo=lo, hi=hi, left=left, right=right, level=level, m=m, r=r, isleaf=bool(isleaf))

    def node(self, i):
        """Generators of node ``i`` as NumPy arrays (tests / inspection)."""
        d = self._info(i)
        m, r = d["m"], d["r"]
        # the library fills tightly packed column-major arrays: allocate the transposed shape in C order
        p = np.zeros(max(m, 1), dtype=np.int64)
        T = np.zeros((r, max(m - r, 0)), dtype=self.dtype)
        D = np.zeros((m, m), dtype=self.dtype) if d["isleaf"] else None
        B12 = B21 = None
        if not d["isleaf"]:
            rl, rr = self._info(d["left"])["r"], self._info(d["right"])["r"]
            B12 = np.zeros((rr, rl), dtype=self.dtype)
            B21 = np.zeros((rl, rr), dtype=self.dtype)

        def vp(a):
            return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None

        root = i == 0
        _lib.check(self.L.hs_hss_node_data(self._h, i, None if root else p.ctypes.data_as(_lib.p_i64), None if root else vp(T),
                                           vp(D), vp(B12), vp(B21)))
        d["p"] = p[:m]
        d["T"] = T.T.copy()
        d["D"] = D.T.copy() if D is not None else None
        d["B12"] = B12.T.copy() if B12 is not None else None
        d["B21"] = B21.T.copy() if B21 is not None else None
        return d

    def getindex(self, I, J):
        """``H[I, J]`` for 0-based index lists (entry access without expanding the matrix)."""
        I = np.ascontiguousarray(I, dtype=np.int64).reshape(-1)
        J = np.ascontiguousarray(J, dtype=np.int64).reshape(-1)
        out = np.zeros((len(J), len(I)), dtype=self.dtype)  # column-major ni x nj
        if out.size:
            _lib.check(self.L.hs_hss_getindex(self._h, I.ctypes.data_as(_lib.p_i64), len(I), J.ctypes.data_as(_lib.p_i64), len(J),
                                              out.ctypes.data_as(C.c_void_p), len(I), 0))
        return out.T.copy()

    def basis(self, node):
        """Expanded basis ``U`` of a non-root node (``size(node) x rank``)."""
        d = self._info(node)
        rows, r = d["hi"] - d["lo"], d["r"]
        out = np.zeros((r, rows), dtype=self.dtype)
        _lib.check(self.L.hs_hss_basis(self._h, node, out.ctypes.data_as(C.c_void_p), rows, 0))
        return out.T.copy()

    def offdiag(self):
        """The two off-diagonal blocks of the top-level split in low-rank form: ``(U1, B12, U2, B21)`` with
        ``A12 = U1 @ B12 @ U2.T`` and ``A21 = U2 @ B21 @ U1.T`` (``generators`` / ``S.B12`` / ``S.B21``, factorization.jl:129-137)."""
        root = self.node(0)
        return self.basis(root["left"]), root["B12"], self.basis(root["right"]), root["B21"]

    def _block(self, X):
        X = np.asarray(X)
        one = X.ndim == 1
        X2 = np.asfortranarray(X.reshape(self.shape[0], -1).astype(self.dtype))
        return X2, one

    def matmul(self, X):
        X2, one = self._block(X)
        Y = np.zeros_like(X2, order="F")
        n, q = X2.shape
        _lib.check(self.L.hs_hss_mul(self._h, X2.ctypes.data_as(C.c_void_p), n, Y.ctypes.data_as(C.c_void_p), n, q, 0))
        return Y[:, 0] if one else Y

    __matmul__ = matmul

    def rmatmul_t(self, X):
        """``H^T @ X`` (plain transpose)."""
        X2, one = self._block(X)
        Y = np.zeros_like(X2, order="F")
        n, q = X2.shape
        _lib.check(self.L.hs_hss_mul_t(self._h, X2.ctypes.data_as(C.c_void_p), n, Y.ctypes.data_as(C.c_void_p), n, q, 0))
        return Y[:, 0] if one else Y

    def offdiag_lowrank(self, which):
        """``(C, Z)`` with ``A12 = C @ Z`` (``which = 0``: ``C = U1 @ B12``, ``Z = U2.T``) or ``A21 = C @ Z`` (``which = 1``) for the top-level
        split: the thin factors a parent front takes ``Aib`` / ``Abi`` from (``Uint = generators(S.A11)[1]*S.B12``, factorization.jl:129-137)."""
        root = self._info(0)
        a, b = (root["left"], root["right"]) if which == 0 else (root["right"], root["left"])
        da, db = self._info(a), self._info(b)
        na, nb, rb = da["hi"] - da["lo"], db["hi"] - db["lo"], db["r"]
        Cm = np.zeros((max(rb, 1), na), dtype=self.dtype)  # column-major na x rb
        Z = np.zeros((nb, max(rb, 1)), dtype=self.dtype)   # column-major rb x nb
        _lib.check(self.L.hs_hss_offdiag(self._h, int(which), Cm.ctypes.data_as(C.c_void_p), na, Z.ctypes.data_as(C.c_void_p), max(rb, 1), 0))
        return Cm.T[:, :rb].copy(), Z.T[:rb].copy()

    @property
    def depth(self):
        return int(self.L.hs_hss_depth(self._h))

    def prune_leaves(self):
        """``prune_leaves!``: every node whose two children are leaves becomes a leaf; the same matrix on a shallower tree (a view sharing the
        untouched generators)."""
        h = C.c_void_p()
        _lib.check(self.L.hs_hss_prune_leaves(self._h, C.byref(h)))
        return HssMatrix(h, self.is_complex, parent=self)

    def compatible(self, other):
        """``compatible(cluster(self), cluster(other))``: the two cluster trees have the same shape."""
        return bool(self.L.hs_hss_compatible(self._h, other._h))

    def view(self):
        """All of ``H`` as a view in cluster-tree order (``H``'s own permutation dropped); shares the generators."""
        return self.block(2)

    def block(self, which):
        """``H.A11`` (0) or ``H.A22`` (1): the diagonal block of the top-level split as an HSS matrix sharing this one's generators."""
        h = C.c_void_p()
        _lib.check(self.L.hs_hss_child(self._h, int(which), C.byref(h)))
        return HssMatrix(h, self.is_complex, parent=self)

    def factor(self):
        _lib.check(self.L.hs_hss_factor(self._h))
        return self

    def ldiv(self, B):
        """``H \\ B`` (the ULV-type elimination is computed on first use)."""
        B2, one = self._block(B)
        B2 = B2.copy(order="F")
        n, q = B2.shape
        _lib.check(self.L.hs_hss_ldiv(self._h, B2.ctypes.data_as(C.c_void_p), n, q, 0))
        return B2[:, 0] if one else B2

    def pack(self, device="cuda:0"):
        """The whole matrix in ONE device buffer (a ``torch.uint8`` tensor): what crosses ranks at a join when a child's Schur complement
        travels as an ``HssMatrix`` (src/factorization.jl:126-140 reads S1, S2 as HSS; include/hs_hss.h ``hs_hss_pack``)."""
        import torch

        nb = C.c_int64(0)
        _lib.check(self.L.hs_hss_pack_size(self._h, C.byref(nb)))
        buf = torch.empty(int(nb.value), dtype=torch.uint8, device=device)
        _lib.check(self.L.hs_hss_pack(self._h, C.c_void_p(buf.data_ptr()), int(nb.value), None))
        return buf

    @staticmethod
    def unpack(buf, is_complex):
        """A new matrix from a packed buffer (``pack`` of another rank); the buffer is copied and may be dropped."""
        h = C.c_void_p()
        _lib.check(_lib.lib().hs_hss_unpack(C.c_void_p(buf.data_ptr()), int(buf.numel()), int(bool(is_complex)), None, C.byref(h)))
        return HssMatrix(h, bool(is_complex))

    def full(self):
        return self.matmul(np.eye(self.shape[0], dtype=self.dtype))

    def expand(self):
        """``Matrix(H)`` by the expansion kernels (every basis once, then the off-diagonal products: 2 n^2 r flops), in the matrix's OWN
        index order: ``A[perm][:, perm]`` when it was compressed with a permutation; block views carry none."""
        n = self.shape[0]
        out = np.zeros((n, n), dtype=self.dtype, order="F")
        if n:
            _lib.check(self.L.hs_hss_expand(self._h, out.ctypes.data_as(C.c_void_p), n, 0))
        return out


def compress(A, cl=None, *, leafsize=64, atol=1e-6, rtol=1e-6, kest=64, seed=123, pad=8, level_scale=0.5, perm=None):
    """``compress(A, cl, cl; atol, rtol)``: HSS form of the dense matrix ``A`` on the GPU (``perm``: 0-based
    permutation, the tree is built over ``A[perm][:, perm]``; products and solves keep the caller's order)."""
    A = np.asarray(A)
    n = A.shape[0]
    if A.ndim != 2 or A.shape[1] != n:
        raise _lib.DimensionMismatch(f"compress needs a square matrix, got {A.shape}")
    first, n_cl, leaf = cl if cl is not None else (0, n, leafsize)
    if n_cl != n:
        raise _lib.DimensionMismatch(f"cluster tree covers {n_cl} indices, the matrix has {n}")
    is_c = np.iscomplexobj(A)
    Af = np.asfortranarray(A.astype(np.complex128 if is_c else np.float64))
    o = _lib.hs_hss_options(leaf, first, atol, rtol, kest, pad, seed, level_scale)
    h = C.c_void_p()
    L = _lib.lib()
    f = L.hs_hss_compress_ex_z if is_c else L.hs_hss_compress_ex_d
    pp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.int64)
        if perm.shape != (n,):
            raise _lib.DimensionMismatch(f"perm has {perm.shape} entries, the matrix has {n} rows")
        pp = perm.ctypes.data_as(_lib.p_i64)
    _lib.check(f(n, Af.ctypes.data_as(C.c_void_p), n, 0, pp, C.byref(o), None, C.byref(h)))
    return HssMatrix(h, is_c)


def compress_lowrank_update(B, Cm, M, Z, cl=None, *, leafsize=64, atol=1e-6, rtol=1e-6, kest=64, seed=123, pad=8, level_scale=0.5, perm=None):
    """HSS form of the operator ``B - Cm @ M @ Z`` without forming it: the Schur complement ``S = Abb - Abi*R`` of a compressed front
    as the reference compresses it, from products and entries (``_schur_complement`` / ``_sample_schur!`` / ``_getindex_schur``,
    factorization.jl:228-249, under ``randcompress_adaptive``, :110)."""
    B, Cm, M, Z = (np.asarray(a) for a in (B, Cm, M, Z))
    n = B.shape[0]
    if B.shape != (n, n) or Cm.shape[0] != n or Z.shape[1] != n or M.shape != (Cm.shape[1], Z.shape[0]):
        raise _lib.DimensionMismatch(f"B {B.shape}, C {Cm.shape}, M {M.shape}, Z {Z.shape} do not form B - C*M*Z")
    first, n_cl, leaf = cl if cl is not None else (0, n, leafsize)
    is_c = any(np.iscomplexobj(a) for a in (B, Cm, M, Z))
    dt = np.complex128 if is_c else np.float64
    Bf, Cf, Mf, Zf = (np.asfortranarray(a.astype(dt)) for a in (B, Cm, M, Z))
    o = _lib.hs_hss_options(leaf, first, atol, rtol, kest, pad, seed, level_scale)
    h = C.c_void_p()
    L = _lib.lib()
    f = L.hs_hss_compress_lru_z if is_c else L.hs_hss_compress_lru_d
    pp = None
    if perm is not None:
        perm = np.ascontiguousarray(perm, dtype=np.int64)
        pp = perm.ctypes.data_as(_lib.p_i64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    r1, r2 = M.shape
    _lib.check(f(n, vp(Bf), n, vp(Cf), n, vp(Mf), max(r1, 1), vp(Zf), max(r2, 1), r1, r2, 0, pp, C.byref(o), None, C.byref(h)))
    return HssMatrix(h, is_c)


def compress_lowrank_update_batch(items, *, leafsize=64, atol=1e-6, rtol=1e-6, kest=64, seed=123, pad=8, level_scale=0.5, device="cuda:0"):
    """``compress_lowrank_update`` for several operators at once (``hs_hss_compress_lru_multi``): the Schur complements of the fronts of one
    tree level, compressed as ONE forest so that a stage of a cluster-tree level is one group of launches for all of them.
    ``items``: dicts with ``B`` and optionally ``C, M, Z`` (the update), ``perm``, ``cl = (first_split, n, leafsize)``."""
    import torch

    cnt = len(items)
    is_c = any(np.iscomplexobj(it[k]) for it in items for k in ("B", "C", "M", "Z") if it.get(k) is not None)
    dt = np.complex128 if is_c else np.float64
    keep, perms, opts = [], [], []

    def dev(a):  # column-major on the device
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(a).astype(dt).T)).to(device)
        keep.append(t)
        return t.data_ptr()

    n = (C.c_int64 * cnt)()
    ldb, ldc, ldm, ldz, r1, r2 = ((C.c_int64 * cnt)() for _ in range(6))
    Bp, Cp, Mp, Zp = ((C.c_void_p * cnt)() for _ in range(4))
    pp = (_lib.p_i64 * cnt)()
    op = (C.POINTER(_lib.hs_hss_options) * cnt)()
    for b, it in enumerate(items):
        Bm = np.asarray(it["B"])
        nb = Bm.shape[0]
        n[b], ldb[b], Bp[b] = nb, nb, dev(Bm)
        if it.get("C") is not None:
            Cm, M, Z = (np.asarray(it[k]) for k in ("C", "M", "Z"))
            r1[b], r2[b] = M.shape
            Cp[b], Mp[b], Zp[b] = dev(Cm), dev(M), dev(Z)
            ldc[b], ldm[b], ldz[b] = nb, max(M.shape[0], 1), max(M.shape[1], 1)
        first, _, leaf = it["cl"] if it.get("cl") is not None else (0, nb, leafsize)
        o = _lib.hs_hss_options(leaf, first, atol, rtol, kest, pad, seed + b, level_scale)
        opts.append(o)
        op[b] = C.pointer(o)
        if it.get("perm") is not None:
            q = np.ascontiguousarray(it["perm"], dtype=np.int64)
            perms.append(q)
            pp[b] = q.ctypes.data_as(_lib.p_i64)
    out = (C.c_void_p * cnt)()
    L = _lib.lib()
    f = L.hs_hss_compress_lru_multi_z if is_c else L.hs_hss_compress_lru_multi_d
    _lib.check(f(cnt, n, Bp, ldb, Cp, ldc, Mp, ldm, Zp, ldz, r1, r2, pp, op, None, out))
    torch.cuda.synchronize()
    return [HssMatrix(C.c_void_p(out[b]), is_c) for b in range(cnt)]


def trim():
    """Give the device blocks the library recycles between factorizations (up to 24 + 16 GiB) back to the driver; returns the bytes released."""
    return int(_lib.lib().hs_hss_trim())


def qr_order(mode=None):
    """Row order of the rank-revealing orthogonalisation inside every compression (``pqrfact``'s role, factorization.jl:171-182):
    ``"lu"`` (default: tournament-pivoted LU order + windowed Cholesky-QR) or ``"norm"`` (blocked column-pivoted QR by downdated residual
    norms: the greedy order itself, slower; include/hs_hss.h hs_hss_qr_order).  Returns the previous setting; ``None`` only reads."""
    code = -1 if mode is None else {"lu": 0, "norm": 1}[mode]
    return ("lu", "norm")[int(_lib.lib().hs_hss_qr_order(code))]


def randcompress_adaptive(A, cl=None, *, kest=64, **kw):
    """``randcompress_adaptive(A, cl, cl; kest, atol, rtol)`` (factorization.jl:110): the compression IS randomized and
    adaptive (samples double until every rank fits); ``kest`` is the initial number of samples."""
    return compress(A, cl, kest=kest, **kw)


def hssrank(H):
    return H.rank


class SparseDevice:
    """A sparse matrix resident on the GPU as CSC and CSR (0-based), plus the scratch map the block operators use (``hs_sparse_dev``)."""

    def __init__(self, A, device="cuda:0"):
        import scipy.sparse as sp
        import torch

        A = sp.csc_matrix(A)
        A.sort_indices()
        R = sp.csr_matrix(A)
        R.sort_indices()
        self.is_complex = bool(np.iscomplexobj(A.data))
        dt = np.complex128 if self.is_complex else np.float64
        dev = torch.device(device)
        t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, dtype=d)).to(dev)  # noqa: E731
        self.n = A.shape[0]
        self._keep = [t(A.indptr, np.int64), t(A.indices, np.int32), t(A.data, dt), t(R.indptr, np.int64), t(R.indices, np.int32), t(R.data, dt)]
        self.lpos = torch.full((self.n,), -1, dtype=torch.int32, device=dev)
        self.c = _lib.hs_sparse_dev(self.n, *[C.c_void_p(x.data_ptr()) for x in self._keep])
        self.device = dev


class BlockOperator:
    """``Op = [H1  A[g1,g2]; A[g2,g1]  H2]`` on the device, never formed (``hs_hss_blockop``): what ``_assemble_blocks`` builds from the children's
    HSS Schur complements and the sparse couplings of ``A`` (factorization.jl:126-140).  ``gid``: 0-based global ids of the n1 + n2 indices."""

    def __init__(self, H1, H2, gid, As):
        self.H1, self.H2, self.As = H1, H2, As
        self.n1 = H1.shape[0] if H1 is not None else 0
        self.n2 = H2.shape[0] if H2 is not None else 0
        self.gid = np.ascontiguousarray(gid, dtype=np.int64)
        if self.gid.shape != (self.n1 + self.n2,):
            raise _lib.DimensionMismatch(f"gid has {self.gid.shape} entries, the operator {self.n1 + self.n2} indices")
        self.is_complex = As.is_complex
        self.c = _lib.hs_hss_blockop(self.n1, self.n2, H1._h if H1 is not None else None, H2._h if H2 is not None else None,
                                     self.gid.ctypes.data_as(_lib.p_i64), C.pointer(As.c), C.c_void_p(As.lpos.data_ptr()))

    @property
    def shape(self):
        return (self.n1 + self.n2, self.n1 + self.n2)

    def matmul(self, X, trans=False):
        """``Op @ X`` (``trans``: ``Op.T @ X``) for a host block; staged through torch device tensors."""
        import torch

        n = self.n1 + self.n2
        dt = np.complex128 if self.is_complex else np.float64
        X2 = np.asarray(X).reshape(n, -1).astype(dt)
        Xd = torch.from_numpy(np.ascontiguousarray(X2.T)).to(self.As.device)  # row-major q x n == column-major n x q
        Yd = torch.zeros_like(Xd)
        _lib.check(_lib.lib().hs_hss_blockop_apply(C.byref(self.c), int(self.is_complex), C.c_void_p(Xd.data_ptr()), n, C.c_void_p(Yd.data_ptr()), n,
                                                   X2.shape[1], int(trans), None))
        torch.cuda.synchronize(self.As.device)
        Y = Yd.cpu().numpy().T
        return Y[:, 0] if np.asarray(X).ndim == 1 else Y

    def compress(self, cl=None, *, leafsize=64, atol=1e-6, rtol=1e-6, kest=64, seed=123, pad=8, level_scale=0.5, perm=None, update=None):
        """HSS form of ``(Op - C @ M @ Z)[perm][:, perm]`` compressed from products and entries (``randcompress_adaptive`` on the operator,
        factorization.jl:110,228-249); ``update = (C, M, Z)`` host arrays or ``None``."""
        import torch

        n = self.n1 + self.n2
        first, n_cl, leaf = cl if cl is not None else (0, n, leafsize)
        dt = np.complex128 if self.is_complex else np.float64
        o = _lib.hs_hss_options(leaf, first, atol, rtol, kest, pad, seed, level_scale)
        h = C.c_void_p()
        L = _lib.lib()
        f = L.hs_hss_compress_blockop_z if self.is_complex else L.hs_hss_compress_blockop_d
        pp = None
        if perm is not None:
            perm = np.ascontiguousarray(perm, dtype=np.int64)
            pp = perm.ctypes.data_as(_lib.p_i64)
        args = [None, 0, None, 0, None, 0, 0, 0]
        keep = []
        if update is not None:
            Cm, M, Z = (np.asarray(a).astype(dt) for a in update)
            r1, r2 = M.shape
            for a in (Cm, M, Z):
                keep.append(torch.from_numpy(np.ascontiguousarray(a.T)).to(self.As.device))  # column-major on the device
            args = [C.c_void_p(keep[0].data_ptr()), n, C.c_void_p(keep[1].data_ptr()), max(r1, 1), C.c_void_p(keep[2].data_ptr()), max(r2, 1), r1, r2]
        _lib.check(f(C.byref(self.c), *args, pp, C.byref(o), None, C.byref(h)))
        H = HssMatrix(h, self.is_complex)
        H._operator = self  # the diagonal blocks are only read during the compression, but keep the inputs alive for inspection
        return H


def equilibrate_clusters(S1, S2, verbose=False):
    """``_equilibrate_clusters(S1, S2)`` (factorization.jl:143-168): prune the leaves of the deeper of ``S1.A11`` / ``S2.A11`` (of both at equal
    depth) until their cluster trees are compatible; returns the two diagonal blocks.  Raises the reference's error when one of them turns
    into a leaf.  The device elimination does not need it (every compression samples an operator); provided for HSS-by-HSS hosts."""
    A1, A2 = S1.block(0), S2.block(0)
    while not A1.compatible(A2):
        d1, d2 = A1.depth, A2.depth
        if d1 > d2:
            if verbose:
                print("Pruning clusters of node 1")
            A1 = A1.prune_leaves()
        elif d1 < d2:
            if verbose:
                print("Pruning clusters of node 2")
            A2 = A2.prune_leaves()
        else:
            if verbose:
                print("Pruning both clusters")
            A1, A2 = A1.prune_leaves(), A2.prune_leaves()
        if A1.num_nodes == 1 or A2.num_nodes == 1:
            raise RuntimeError("One of the Schur complements turned into a leaf. Aborting.")
    return A1, A2
