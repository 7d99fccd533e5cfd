"""Host mirror of the reference's numeric API over the C ABI.

=====================================  ====================================================
reference (Julia)                      here
=====================================  ====================================================
``SolverOptions(; kw...)``             :class:`SolverOptions`  (HierarchicalSolvers.jl:30-79)
``factor(A, nd, nd_loc, opts; kw...)`` :func:`factor`          (factorization.jl:5-11)
``FactorNode{T}``                      :class:`FactorNode`     (factornode.jl:7-39)
``ldiv!(F, B)``, ``ldiv!(C, F, B)``    :func:`ldiv`            (factornode.jl:62-74)
``maxrank(F)``                         :func:`maxrank`         (factornode.jl:49-57)
``F \\ b``                              ``F.solve(b)``
=====================================  ====================================================

All arithmetic happens in ``libhs_solver.so`` on the GPU; this module only marshals arrays.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from .nesteddissection import flatten_tree

__all__ = ["SolverOptions", "chkopts", "factor", "factorize", "FactorNode", "ldiv", "maxrank"]


class SolverOptions:
    """Mirror of ``mutable struct SolverOptions`` and its kw-constructor defaults
    ``5, 1, 1e-6, 1e-6, 0.5, 32, -1, 10, false`` (HierarchicalSolvers.jl:43-54)."""

    _fields = ("swlevel", "swsize", "atol", "rtol", "c_tol", "leafsize", "kest", "stepsize", "verbose")
    _ext = ("keep_schur", "seed", "profile", "split_size", "hss_min", "hss_dexp", "mf", "dist_top")

    def __init__(self, **kw):
        self.swlevel, self.swsize = 5, 1
        self.atol, self.rtol, self.c_tol = 1e-6, 1e-6, 0.5
        self.leafsize, self.kest, self.stepsize = 32, -1, 10
        self.verbose = False
        self.keep_schur = False
        self.profile = False
        self.seed = 123
        self.split_size = 0  # columns per slice of a compressed front's interior block (multiple of 256, 0 = off)
        self.hss_min = 0  # > 0 (multiple of 1024): compressed-level fronts with at least this many interior DOFs keep D as HSS
        self.hss_dexp = None  # orders of magnitude by which the HSS form of D is tighter than atol, rtol (None: 2; 0: the same)
        # matrix-free compressed branch: S leaves flagged fronts as HSS, parents assemble from the children's generators.  True / 'dense': the
        # interior block D of such a parent is expanded and eliminated densely (HSS only where hss_min says so); 2 / 'hss': D is an HSS matrix
        self.mf = False
        # multi-rank factorizations: fronts above the rank cut are eliminated by all ranks of their group (csrc/hs_dist.h) instead of its first rank
        self.dist_top = False
        self._set(kw)

    def _set(self, kw):
        for k, v in kw.items():
            if k not in self._fields and k not in self._ext:
                raise TypeError(f"type SolverOptions has no field {k}")  # setfield! on an unknown field
            setattr(self, k, v)

    def copy(self, **kw):
        """``copy(opts; kw...)`` (HierarchicalSolvers.jl:62-71)."""
        o = SolverOptions()
        for f in self._fields + self._ext:
            setattr(o, f, getattr(self, f))
        o._set(kw)
        return o

    def to_c(self):
        o = _lib.hs_options()
        o.swlevel, o.swsize = int(self.swlevel), int(self.swsize)
        o.atol, o.rtol, o.c_tol = float(self.atol), float(self.rtol), float(self.c_tol)
        o.leafsize, o.kest, o.stepsize = int(self.leafsize), int(self.kest), int(self.stepsize)
        o.verbose = 1 if self.verbose else 0
        o.keep_schur = 1 if self.keep_schur else 0
        o.profile = 1 if self.profile else 0
        o.seed = int(self.seed)
        ss = int(self.split_size)
        if ss < 0 or ss % 256 or ss // 256 > 255:
            raise ValueError("split_size must be a multiple of 256 in 0:65280")
        o.split = ss // 256
        hm = int(self.hss_min)
        if hm < 0 or hm % 1024 or hm // 1024 > 255:
            raise ValueError("hss_min must be a multiple of 1024 in 0:261120")
        o.hss_d = hm // 1024
        if self.mf in (False, None, 0):
            o.mf = 0
        elif self.mf in (True, 1, "dense"):
            o.mf = 1  # interior blocks of the matrix-free fronts dense unless hss_min asks for HSS
        elif self.mf in (2, "hss"):
            o.mf = 2  # every interior block an HSS matrix (the reference's formulation)
        elif self.mf in (3, "block"):
            o.mf = 3  # interior blocks as the reference's 2x2 block factorization over the children's HSS blocks (blockmatrix.jl:121-130)
        else:
            raise ValueError("mf must be False, True / 'dense', 2 / 'hss' or 3 / 'block'")
        o.dist_top = 1 if self.dist_top else 0
        if self.hss_dexp is not None:
            if not 0 <= int(self.hss_dexp) <= 12:
                raise ValueError("hss_dexp must be in 0:12")
            o.hss_dexp = int(self.hss_dexp) + 1
        return o


def chkopts(opts):
    """``chkopts!`` (HierarchicalSolvers.jl:73-79); ``ArgumentError`` -> ``ValueError``."""
    if not opts.swsize >= 1:
        raise ValueError("ArgumentError: swsize")
    if not opts.atol >= 0.0:
        raise ValueError("ArgumentError: atol")
    if not opts.rtol >= 0.0:
        raise ValueError("ArgumentError: rtol")
    if not (0.0 < opts.c_tol <= 1.0):
        raise ValueError("ArgumentError: c_tol")
    if not opts.leafsize >= 1:
        raise ValueError("ArgumentError: leafsize")


def _p64(a):
    return a.ctypes.data_as(_lib.p_i64)


def _pf64(a):
    return a.ctypes.data_as(_lib.p_f64)


class FactorNode:
    """Handle to a device-resident factorization (the reference's ``FactorNode{T}`` tree,
    factornode.jl:7-39).  The per-node blocks stay in HBM; ``node_blocks`` exports one node for
    inspection.  Freed by ``hs_free`` when garbage-collected (the Julia shim uses a finalizer)."""

    def __init__(self, handle, dtype, n, flat):
        self._h = handle
        self.dtype = np.dtype(dtype)
        self.n = int(n)
        self._flat = flat

    def __del__(self):
        self.free()

    def free(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            _lib.lib().hs_free(h)

    @property
    def eltype(self):  # eltype(::FactorNode{T}) (factornode.jl:41)
        return self.dtype.type

    def __repr__(self):  # Base.show (factornode.jl:42)
        return f"FactorNode{{{'ComplexF64' if self.dtype.kind == 'c' else 'Float64'}}}"

    @property
    def shape(self):
        return (self.n, self.n)

    def stats(self):
        st = _lib.hs_stats()
        _lib.check(_lib.lib().hs_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def solve(self, b):
        """``F \\ b``."""
        return ldiv(self, b)

    # -- inspection (tests) -------------------------------------------------------------------
    @property
    def nnodes(self):
        return int(self._flat["nnodes"])

    def node_info(self, node):
        ni, nb, lv = _lib.i64(), _lib.i64(), _lib.i64()
        _lib.check(_lib.lib().hs_node_info(self._h, node, C.byref(ni), C.byref(nb), C.byref(lv)))
        return ni.value, nb.value, lv.value

    def node_ranks(self, node):
        """``(compressed, rank(L), rank(R))`` of one front (ranks are 0 for dense Gauss transforms)."""
        rl, rr = _lib.i64(), _lib.i64()
        st = _lib.lib().hs_node_ranks(self._h, node, C.byref(rl), C.byref(rr))
        if st < 0:
            _lib.check(st)
        return bool(st), rl.value, rr.value

    def node_blocks(self, node, with_schur=False):
        """Stored blocks of one node: ``LU`` (ni x ni packed), ``Lbi`` (nb x ni), ``Uib`` (ni x nb),
        ``rperm`` (0-based, ``(P x)[i] = x[rperm[i]]``) and optionally ``S`` (needs ``keep_schur``)."""
        ni, nb, _ = self.node_info(node)
        L = _lib.lib()
        out = {}
        for name, which, shp in (("LU", _lib.HS_BLK_LU, (ni, ni)), ("Lbi", _lib.HS_BLK_LBI, (nb, ni)), ("Uib", _lib.HS_BLK_UIB, (ni, nb))) + (
            (("S", _lib.HS_BLK_S, (nb, nb)),) if with_schur else ()
        ):
            a = np.zeros(shp, dtype=self.dtype, order="F")
            if a.size:
                _lib.check(L.hs_node_export(self._h, node, which, a.ctypes.data_as(_lib.p_f64)))
            out[name] = a
        rp = np.zeros(ni, dtype=np.int64)
        if ni:
            _lib.check(L.hs_node_export_piv(self._h, node, _p64(rp)))
        out["rperm"] = rp
        return out

    def schur_hss(self, node, **kw):
        """``F.S`` of one node as an :class:`hss.HssMatrix` (``compress(S[perm,perm], cl, cl)``, factorization.jl:56-57,109-110;
        needs ``keep_schur``).  Without keywords the factorization's own ``leafsize, atol, rtol, kest, seed`` apply; with any of
        ``leafsize, atol, rtol, kest, seed, pad, level_scale`` given, the others take the ``SolverOptions`` defaults."""
        import ctypes as C

        from . import hss

        o = None
        if kw:
            d = dict(leafsize=32, atol=1e-6, rtol=1e-6, kest=64, pad=8, seed=123, level_scale=0.5)  # SolverOptions defaults
            d.update(kw)
            o = C.byref(_lib.hs_hss_options(d["leafsize"], 0, d["atol"], d["rtol"], d["kest"], d["pad"], d["seed"], d["level_scale"]))
        h = C.c_void_p()
        _lib.check(_lib.lib().hs_node_schur_hss(self._h, node, o, C.byref(h)))
        return hss.HssMatrix(h, self.dtype == np.complex128)

    def reference_blocks(self, node, with_schur=False):
        """The reference's FactorNode fields of one node, rebuilt from the stored factors:
        ``D = P'LU`` (raw interior block), ``L = Abi*D^-1``, ``R = D^-1*Aib`` (factorization.jl:33-37,69-71)."""
        import scipy.linalg as sla

        b = self.node_blocks(node, with_schur)
        LU, rp = b["LU"], b["rperm"]
        ni = LU.shape[0]
        Lm = np.tril(LU, -1) + np.eye(ni, dtype=self.dtype)
        Um = np.triu(LU)
        PD = Lm @ Um
        D = np.empty_like(PD)
        D[rp] = PD  # (P D)[i] = D[rperm[i]]
        # L_ref = Lbi * L^-1 * P ;  R_ref = U^-1 * Uib
        X = sla.solve_triangular(Lm, b["Lbi"].T, lower=True, trans="T", unit_diagonal=True) if ni else b["Lbi"].T  # L^-T Lbi^T
        Lref = np.empty_like(b["Lbi"])
        Lref[:, rp] = X.T  # (M P)[:, rp[i]] = M[:, i]
        Rref = sla.solve_triangular(Um, b["Uib"], lower=False) if ni and b["Uib"].size else b["Uib"].copy()
        out = dict(D=D, L=Lref, R=Rref)
        if with_schur:
            out["S"] = b["S"]
        return out


def factor(A, nd, nd_loc, opts=None, **kw):
    """``factor(A::SparseMatrixCSC{T}, nd, nd_loc, opts=SolverOptions(); kw...) -> FactorNode{T}``
    (factorization.jl:5-11).  ``T`` is ``float64`` or ``complex128``."""
    opts = (opts or SolverOptions()).copy(**kw)
    chkopts(opts)
    A = sp.csc_matrix(A)
    if A.shape[0] != A.shape[1]:
        raise _lib.DimensionMismatch("DimensionMismatch: A is not square")
    A.sort_indices()
    n = A.shape[0]
    is_c = np.iscomplexobj(A.data)
    dtype = np.complex128 if is_c else np.float64
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1  # Julia's 1-based SparseMatrixCSC fields
    rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
    nzval = np.ascontiguousarray(A.data, dtype=dtype)
    flat = flatten_tree(nd, nd_loc)
    t = _lib.hs_tree()
    t.nnodes = flat["nnodes"]
    for k in ("left", "right", "int_ptr", "int_idx", "bnd_ptr", "bnd_idx", "iloc_ptr", "iloc_idx", "bloc_ptr", "bloc_idx"):
        flat[k] = np.ascontiguousarray(flat[k], dtype=np.int64)
        setattr(t, k, _p64(flat[k]))
    co = opts.to_c()
    h = C.c_void_p()
    L = _lib.lib()
    fn = L.hs_factor_z if is_c else L.hs_factor_d
    st = fn(n, _p64(colptr), _p64(rowval), nzval.ctypes.data_as(_lib.p_f64), C.byref(t), C.byref(co), C.byref(h))
    _lib.check(st)
    return FactorNode(h, dtype, n, flat)


factorize = factor  # BASELINE.json's north star uses this name


def trim():
    """Give the device blocks the library parks between factorizations (the factor arena of the last freed handle, the recycled blocks of the
    HSS / low-rank modules) back to the driver; returns the bytes released (``hs_trim``, include/hs_solver.h)."""
    return int(_lib.lib().hs_trim())


def ldiv(*args):
    """``ldiv!(F, B)`` / ``ldiv!(C, F, B)`` (factornode.jl:62-74): ``C = F^-1 B`` for a vector or an
    ``n x nrhs`` matrix.  The 2-argument form returns a new array like the reference (which
    allocates ``similar(B)``, factornode.jl:62); the 3-argument form writes into ``C`` (``C`` may be ``B``)."""
    if len(args) == 2:
        F, B = args
        Cout = None
    elif len(args) == 3:
        Cout, F, B = args
    else:
        raise TypeError("ldiv(F, B) or ldiv(C, F, B)")
    B = np.asarray(B)
    if B.shape[0] != F.n:
        raise _lib.DimensionMismatch(f"DimensionMismatch: B has {B.shape[0]} rows, F is {F.n} x {F.n}")
    if B.dtype != F.dtype:
        if F.dtype.kind == "f" and B.dtype.kind == "c":
            raise TypeError("MethodError: no method matching ldiv!(::Array{ComplexF64}, ::FactorNode{Float64}, ::Array{ComplexF64})")
        B = B.astype(F.dtype)
    vec = B.ndim == 1
    Bm = np.asfortranarray(B.reshape(F.n, -1))
    Cm = np.empty_like(Bm, order="F")
    L = _lib.lib()
    fn = L.hs_ldiv_z if F.dtype.kind == "c" else L.hs_ldiv_d
    _lib.check(fn(F._h, Cm.ctypes.data_as(_lib.p_f64), F.n, Bm.ctypes.data_as(_lib.p_f64), F.n, F.n, Bm.shape[1]))
    res = Cm[:, 0] if vec else Cm
    if Cout is not None:
        Cout[...] = res
        return Cout
    return res


def maxrank(F):
    """``maxrank(F)`` (factornode.jl:49-57)."""
    return int(_lib.lib().hs_maxrank(F._h))
