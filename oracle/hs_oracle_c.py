"""ctypes driver of the C restatement of the reference's dense path (oracle/hs_oracle_c.c).

TEST INFRASTRUCTURE -- NOT PRODUCT CODE: imported by tests/ and by bench.py's ``cpu_baseline`` leg only.

``factor_solve(A, nd, nd_loc, b, kernels="blas")`` runs ``factor(A, nd, nd_loc; swlevel=0)`` followed by ``ldiv!(F, b)`` exactly as
/root/reference/src/factorization.jl:5-75 and factornode.jl:62-99 spell them (every ``\\`` a fresh LU), with the dense kernels either from the
BLAS / LAPACK SciPy ships (``scipy.linalg.cython_blas`` / ``cython_lapack`` export plain C function pointers; OpenBLAS, threaded) or from the
plain loops of the C file (``kernels="loops"``, OpenMP).  ``nd`` / ``nd_loc`` are the trees of oracle/hs_oracle.py (after ``symfact`` and
``permuted``), flattened here into the post-order arrays the C side takes.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libhs_oracle_c.so")


def build(force=False):
    """``make -C oracle`` (gcc -O2 -fopenmp); __graft_entry__.build() calls this."""
    src = [os.path.join(_HERE, f) for f in ("hs_oracle_c.c", "hs_oracle_c_body.h", "Makefile")]
    if not force and os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src):
        return _SO
    subprocess.run(["make", "-C", _HERE] + (["-B"] if force else []), check=True, capture_output=True)
    return _SO


class _Blas(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("dgemm", "dgetrf", "dgetrs", "zgemm", "zgetrf", "zgetrs")]


class _Tree(C.Structure):
    _fields_ = [("nnodes", C.c_int), ("left", C.c_void_p), ("right", C.c_void_p)] + [
        (k, C.c_void_p) for k in ("int_ptr", "int_idx", "bnd_ptr", "bnd_idx", "li_ptr", "li_idx", "lb_ptr", "lb_idx")
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        for suffix in ("d", "z"):
            f = getattr(_lib, "hsc_factor_solve_" + suffix)
            f.restype = C.c_int
            f.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(_Tree), C.POINTER(_Blas), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
            g = getattr(_lib, "hsc_selftest_solve_" + suffix)
            g.restype = C.c_int
            g.argtypes = [C.POINTER(_Blas), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


def _capsule_pointer(capsule):
    C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
    C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
    C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
    C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    return C.pythonapi.PyCapsule_GetPointer(capsule, C.pythonapi.PyCapsule_GetName(capsule))


def blas_table():
    """Function pointers of the BLAS / LAPACK SciPy links (Fortran convention, 32-bit integers)."""
    import scipy.linalg.cython_blas as cb
    import scipy.linalg.cython_lapack as cl

    t = _Blas()
    for name in ("dgemm", "zgemm"):
        setattr(t, name, _capsule_pointer(cb.__pyx_capi__[name]))
    for name in ("dgetrf", "dgetrs", "zgetrf", "zgetrs"):
        setattr(t, name, _capsule_pointer(cl.__pyx_capi__[name]))
    return t


def flatten(nd, nd_loc):
    """(nd, nd_loc) of oracle/hs_oracle.py -> post-order arrays, 0-based."""
    left, right, ints, bnds, lis, lbs = [], [], [], [], [], []

    def rec(x, xl):
        if (x.left is None) != (x.right is None):
            raise RuntimeError("Expected nested dissection to be a binary tree. Found a node with only one child.")  # factorization.jl:25
        li = ri = -1
        if x.left is not None:
            li = rec(x.left, xl.left)
            ri = rec(x.right, xl.right)
        left.append(li)
        right.append(ri)
        ints.append(np.asarray(x.int, dtype=np.int64) - 1)
        bnds.append(np.asarray(x.bnd, dtype=np.int64) - 1)
        lis.append(np.asarray(xl.int, dtype=np.int64) - 1)
        lbs.append(np.asarray(xl.bnd, dtype=np.int64) - 1)
        return len(left) - 1

    import sys

    old = sys.getrecursionlimit()
    sys.setrecursionlimit(max(old, 10000))
    try:
        rec(nd, nd_loc)
    finally:
        sys.setrecursionlimit(old)

    def csr(parts):
        ptr = np.zeros(len(parts) + 1, dtype=np.int64)
        ptr[1:] = np.cumsum([len(p) for p in parts])
        idx = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
        return ptr, np.ascontiguousarray(idx, dtype=np.int64) if len(idx) else np.zeros(1, dtype=np.int64)

    return dict(left=np.asarray(left, dtype=np.int32), right=np.asarray(right, dtype=np.int32), int=csr(ints), bnd=csr(bnds), li=csr(lis), lb=csr(lbs))


def factor_solve(A, nd, nd_loc, b, kernels="blas"):
    """x = ldiv!(factor(A, nd, nd_loc; swlevel=0), b) in C.  Returns (x, info) with info = seconds, executed flops, getrf counts, |S_i|_F per node."""
    A = sp.csc_matrix(A)
    A.sort_indices()
    is_c = np.iscomplexobj(A.data) or np.iscomplexobj(b)
    dt = np.complex128 if is_c else np.float64
    n = A.shape[0]
    fl = flatten(nd, nd_loc)
    keep = [fl["left"], fl["right"]] + [a for k in ("int", "bnd", "li", "lb") for a in fl[k]]
    t = _Tree()
    t.nnodes = len(fl["left"])
    t.left, t.right = fl["left"].ctypes.data, fl["right"].ctypes.data
    for k in ("int", "bnd", "li", "lb"):
        setattr(t, k + "_ptr", fl[k][0].ctypes.data)
        setattr(t, k + "_idx", fl[k][1].ctypes.data)
    colptr = np.ascontiguousarray(A.indptr, dtype=np.int64)
    rowidx = np.ascontiguousarray(A.indices, dtype=np.int64)
    vals = np.ascontiguousarray(A.data, dtype=dt)
    B = np.asarray(b)
    vec = B.ndim == 1
    X = np.asfortranarray(B.reshape(n, -1).astype(dt))
    snorm = np.zeros(t.nnodes)
    stats = np.zeros(8)
    tab = blas_table() if kernels == "blas" else _Blas()
    f = lib().hsc_factor_solve_z if is_c else lib().hsc_factor_solve_d
    rc = f(n, colptr.ctypes.data, rowidx.ctypes.data, vals.ctypes.data, C.byref(t), C.byref(tab), X.ctypes.data, X.shape[1], snorm.ctypes.data, stats.ctypes.data)
    del keep
    if rc != 0:
        raise RuntimeError("Expected nested dissection to be a binary tree. Found a node with only one child.")
    info = dict(factor_s=stats[0], ldiv_s=stats[1], factor_flops=stats[2], ldiv_flops=stats[3], factor_getrf=int(stats[4]), ldiv_getrf=int(stats[5]),
                bytes_allocated=stats[6], singular=int(stats[7]), snorm=snorm, kernels=kernels)
    return (X[:, 0] if vec else X), info


def selftest_solve(Am, Bm, kernels):
    """(A \\ B, (B^T / A) read back) through the C kernels: for tests of the plain loops against the BLAS table."""
    is_c = np.iscomplexobj(Am) or np.iscomplexobj(Bm)
    dt = np.complex128 if is_c else np.float64
    Af = np.asfortranarray(Am.astype(dt))
    Bf = np.asfortranarray(Bm.astype(dt))
    n, nrhs = Bf.shape
    X = np.zeros((n, nrhs), dtype=dt, order="F")
    Xr = np.zeros((nrhs, n), dtype=dt, order="F")
    Bt = np.asfortranarray(Bf.T.copy())
    tab = blas_table() if kernels == "blas" else _Blas()
    g = lib().hsc_selftest_solve_z if is_c else lib().hsc_selftest_solve_d
    # the C side reads B twice: as n x nrhs for `\` and as nrhs x n for `/`; two calls keep the shapes explicit
    tmp = np.zeros_like(Xr)
    tmp2 = np.zeros_like(X)
    g(C.byref(tab), n, nrhs, Af.ctypes.data, Bf.ctypes.data, X.ctypes.data, tmp.ctypes.data)
    g(C.byref(tab), n, nrhs, Af.ctypes.data, Bt.ctypes.data, tmp2.ctypes.data, Xr.ctypes.data)
    return X, Xr
