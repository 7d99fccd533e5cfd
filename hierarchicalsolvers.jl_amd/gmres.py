"""Right-preconditioned restarted GMRES with every vector resident on the GPU.

The call the reference's scenario makes (``test/rungmres.jl:47-48``)::

    x, ch = gmres(A, b; Pr=F, reltol=1e-9, restart=30, log=true, maxiter=30)

``IterativeSolvers.gmres`` (0.9.0, not part of the reference tree) uses the preconditioner only through ``ldiv!``.  ``gmres`` here is
``hs_gmres_{d,z}`` of the C ABI (include/hs_solver.h, csrc/hs_gmres.hip): hand-written CSR SpMV, Gram-Schmidt and Givens kernels, the
preconditioner applied through ``hs_ldiv_dev_*`` on device pointers -- what a Julia host calls instead of ``IterativeSolvers.gmres``.
(An independent torch restatement of the same iteration lives under tests/gmres_mirror.py: test infrastructure, not product.)
Parity unpinned: IterativeSolvers is absent.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib

__all__ = ["gmres", "gmres_native", "gmres_device"]


def _csc_fields(A, dtype):
    A = sp.csc_matrix(A)
    A.sort_indices()
    return (np.ascontiguousarray(A.indptr, dtype=np.int64) + 1, np.ascontiguousarray(A.indices, dtype=np.int64) + 1,
            np.ascontiguousarray(A.data, dtype=dtype))


def gmres(A, b, Pr=None, reltol=None, abstol=0.0, restart=None, maxiter=None, log=False, x0=None, device=None):
    """Restarted GMRES(restart) on ``A x = b`` with right preconditioner ``Pr`` (a :class:`FactorNode`), behind the C ABI (``hs_gmres_{d,z}``).

    Defaults follow IterativeSolvers 0.9: ``restart = min(20, n)``, ``maxiter = n``, ``reltol = sqrt(eps)``; convergence when
    ``||b - A x|| <= max(reltol * ||r0||, abstol)``.  Returns ``x`` (NumPy, host) or ``(x, history)`` with
    ``history = dict(resnorm=[...], isconverged, iters)``.  (``device`` is accepted for compatibility; the library uses the current device.)"""
    n = A.shape[0]
    cplx = np.iscomplexobj(A.data) or np.iscomplexobj(b) or (Pr is not None and Pr.dtype.kind == "c")
    dt = np.complex128 if cplx else np.float64
    colptr, rowval, nz = _csc_fields(A, dt)
    bb = np.ascontiguousarray(b, dtype=dt)
    x = np.zeros(n, dtype=dt) if x0 is None else np.ascontiguousarray(x0, dtype=dt).copy()
    maxit = n if maxiter is None else int(maxiter)
    hist = np.zeros(maxit + 2)
    iters, conv = _lib.i64(0), C.c_int(0)
    L = _lib.lib()
    fn = L.hs_gmres_z if cplx else L.hs_gmres_d
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    _lib.check(fn(Pr._h if Pr is not None else None, n, colptr.ctypes.data_as(_lib.p_i64), rowval.ctypes.data_as(_lib.p_i64), vp(nz), vp(bb), vp(x), 0,
                  int(x0 is not None), -1.0 if reltol is None else float(reltol), float(abstol), -1 if restart is None else int(restart), maxit,
                  hist.ctypes.data_as(_lib.p_f64), C.byref(iters), C.byref(conv), None))
    if log:
        return x, dict(resnorm=[float(v) for v in hist[: iters.value + 1]], isconverged=bool(conv.value), iters=int(iters.value))
    return x


gmres_native = gmres  # the name round 2 gave the C-ABI solver while `gmres` was still a torch restatement


def gmres_device(A, b_dev, solver, reltol=1e-9, abstol=0.0, restart=30, maxiter=30):
    """``hs_gmres_*`` on a torch device vector with the factorization held by a :class:`dist.StagedSolver` (single rank) as the right
    preconditioner; returns ``(x_dev, resnorm history)``."""
    import torch

    n = A.shape[0]
    cplx = b_dev.is_complex()
    dt = np.complex128 if cplx else np.float64
    colptr, rowval, nz = _csc_fields(A, dt)
    x = torch.zeros_like(b_dev)
    hist = np.zeros(int(maxiter) + 2)
    iters, conv = _lib.i64(0), C.c_int(0)
    L = _lib.lib()
    fn = L.hs_gmres_z if cplx else L.hs_gmres_d
    stream = C.c_void_p(torch.cuda.current_stream(b_dev.device).cuda_stream)
    _lib.check(fn(solver.backend._h, n, colptr.ctypes.data_as(_lib.p_i64), rowval.ctypes.data_as(_lib.p_i64), nz.ctypes.data_as(C.c_void_p),
                  C.c_void_p(b_dev.data_ptr()), C.c_void_p(x.data_ptr()), 1, 0, float(reltol), float(abstol), int(restart), int(maxiter),
                  hist.ctypes.data_as(_lib.p_f64), C.byref(iters), C.byref(conv), stream))
    return x, [float(v) for v in hist[: iters.value + 1]]
