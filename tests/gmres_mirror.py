"""TEST INFRASTRUCTURE: an independent torch restatement of right-preconditioned restarted GMRES (the iteration of IterativeSolvers 0.9,
test/rungmres.jl:47-48), used to check `hs_gmres_{d,z}` -- iteration counts and residual histories -- in tests/test_gmres_gpu.py.
The Krylov basis, the SpMV (torch sparse CSR) and the dot products are torch ops; the Hessenberg least-squares problem is solved on the host
with Givens rotations.  Moved out of the package in round 3 (the product exports the HIP solver as `gmres`)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from hierarchicalsolvers_jl_amd import _lib


def _apply_prec(F, v, out):
    """out = F^-1 v on the device (``ldiv!`` with device pointers)."""
    import torch

    L = _lib.lib()
    fn = L.hs_ldiv_dev_z if F.dtype.kind == "c" else L.hs_ldiv_dev_d
    stream = C.c_void_p(torch.cuda.current_stream(v.device).cuda_stream)
    _lib.check(fn(F._h, C.c_void_p(out.data_ptr()), F.n, C.c_void_p(v.data_ptr()), F.n, F.n, 1, stream))
    return out


def gmres_mirror(A, b, Pr=None, reltol=None, abstol=0.0, restart=None, maxiter=None, log=False, device="cuda:0", x0=None):
    """Restarted GMRES(restart) on ``A x = b`` with right preconditioner ``Pr`` (a :class:`FactorNode`).

    Defaults follow IterativeSolvers 0.9: ``restart = min(20, n)``, ``maxiter = n``,
    ``reltol = sqrt(eps)``; convergence when ``||b - A x|| <= max(reltol * ||r0||, abstol)``.
    Returns ``x`` (NumPy, host) or ``(x, history)`` with ``history = dict(resnorm=[...], isconverged, iters)``.
    """
    import torch

    dev = torch.device(device)
    A = sp.csr_matrix(A)
    n = A.shape[0]
    cplx = np.iscomplexobj(A.data) or np.iscomplexobj(b) or (Pr is not None and Pr.dtype.kind == "c")
    tdt = torch.complex128 if cplx else torch.float64
    ndt = np.complex128 if cplx else np.float64
    Ad = torch.sparse_csr_tensor(
        torch.from_numpy(A.indptr.astype(np.int64)), torch.from_numpy(A.indices.astype(np.int64)),
        torch.from_numpy(A.data.astype(ndt)), size=A.shape, dtype=tdt, device=dev)
    bd = torch.from_numpy(np.ascontiguousarray(b, dtype=ndt)).to(dev)
    restart = min(20, n) if restart is None else int(restart)
    maxiter = n if maxiter is None else int(maxiter)
    reltol = float(np.sqrt(np.finfo(np.float64).eps)) if reltol is None else float(reltol)

    x = torch.zeros(n, dtype=tdt, device=dev) if x0 is None else torch.from_numpy(np.asarray(x0, dtype=ndt)).to(dev)
    matvec = lambda v: torch.mv(Ad, v)  # noqa: E731
    r = bd - matvec(x) if x0 is not None else bd.clone()
    beta = float(torch.linalg.vector_norm(r))
    tol = max(reltol * beta, abstol)
    hist = [beta]
    it = 0
    converged = beta <= tol
    V = torch.empty((restart + 1, n), dtype=tdt, device=dev)
    Z = torch.empty(n, dtype=tdt, device=dev)
    while not converged and it < maxiter:
        V[0] = r / beta
        H = np.zeros((restart + 1, restart), dtype=ndt)
        cs = np.zeros(restart, dtype=ndt)
        sn = np.zeros(restart, dtype=ndt)
        g = np.zeros(restart + 1, dtype=ndt)
        g[0] = beta
        k_used = 0
        for k in range(restart):
            if it >= maxiter:
                break
            z = _apply_prec(Pr, V[k], Z) if Pr is not None else V[k]
            w = matvec(z)
            # modified Gram-Schmidt (one fused projection + one re-orthogonalisation pass keeps it stable)
            h = torch.mv(V[: k + 1].conj(), w)
            w = w - torch.mv(V[: k + 1].T, h)
            h2 = torch.mv(V[: k + 1].conj(), w)
            w = w - torch.mv(V[: k + 1].T, h2)
            hk = (h + h2).cpu().numpy()
            hn = float(torch.linalg.vector_norm(w))
            H[: k + 1, k] = hk
            H[k + 1, k] = hn
            if hn > 0:
                V[k + 1] = w / hn
            # apply the previous rotations, then a new one annihilating H[k+1,k]
            for i in range(k):
                t = cs[i] * H[i, k] + sn[i] * H[i + 1, k]
                H[i + 1, k] = -np.conj(sn[i]) * H[i, k] + cs[i] * H[i + 1, k]
                H[i, k] = t
            a, bb = H[k, k], H[k + 1, k]
            den = np.sqrt(abs(a) ** 2 + abs(bb) ** 2)
            if den == 0:
                cs[k], sn[k] = 1.0, 0.0
            else:
                cs[k], sn[k] = abs(a) / den, (a / abs(a) if abs(a) > 0 else 1.0) * np.conj(bb) / den
            H[k, k] = cs[k] * a + sn[k] * bb
            H[k + 1, k] = 0.0
            g[k + 1] = -np.conj(sn[k]) * g[k]
            g[k] = cs[k] * g[k]
            it += 1
            k_used = k + 1
            res = abs(g[k + 1])
            hist.append(float(res))
            if res <= tol or hn == 0:
                converged = res <= tol
                break
        # x += Pr^-1 (V_k y),  H y = g
        if k_used:
            y = np.linalg.solve(np.triu(H[:k_used, :k_used]), g[:k_used])
            upd = torch.mv(V[:k_used].T, torch.from_numpy(y.astype(ndt)).to(dev))
            x = x + (_apply_prec(Pr, upd, Z).clone() if Pr is not None else upd)
        r = bd - matvec(x)
        beta = float(torch.linalg.vector_norm(r))
        converged = converged or beta <= tol
        if k_used == 0:
            break
    xh = x.cpu().numpy()
    if log:
        return xh, dict(resnorm=hist, isconverged=bool(converged), iters=it)
    return xh
