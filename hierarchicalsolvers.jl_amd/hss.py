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

__all__ = ["HssMatrix", "compress", "compress_lowrank_update", "randcompress_adaptive", "hssrank", "bisection_cluster"]


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

    def full(self):
        return self.matmul(np.eye(self.shape[0], dtype=self.dtype))


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


def randcompress_adaptive(A, cl=None, *, kest=64, **kw):
    """``randcompress_adaptive(A, cl, cl; kest, atol, rtol)`` (factorization.jl:110): the compression IS randomized and
    adaptive (samples double until every rank fits); ``kest`` is the initial number of samples."""
    return compress(A, cl, kest=kest, **kw)


def hssrank(H):
    return H.rank
