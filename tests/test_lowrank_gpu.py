"""Low-rank compression primitive (randomized row interpolative decomposition through the front kernels)
against NumPy: ||X - C Z|| within the requested tolerance, rank close to the numerical rank (SVD)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def compress(hs, X, atol, rtol, kinit=64, seed=1):
    rows, cols = X.shape
    cplx = np.iscomplexobj(X)
    Xf = np.asfortranarray(X)
    cap = min(rows, cols)
    Cm = np.zeros((rows, cap), dtype=X.dtype, order="F")
    Z = np.zeros((cap, cols), dtype=X.dtype)
    # outputs are written with tight leading dimensions (rows for C, r for Z): use flat buffers
    Cbuf = np.zeros(rows * cap, dtype=X.dtype)
    Zbuf = np.zeros(cap * cols, dtype=X.dtype)
    r = C.c_int64(0)
    L = hs._lib.lib()
    fn = L.hsk_lowrank_z if cplx else L.hsk_lowrank_d
    hs._lib.check(fn(rows, cols, _ptr(Xf), atol, rtol, kinit, seed, C.byref(r), _ptr(Cbuf), _ptr(Zbuf), cap))
    r = r.value
    Cm = Cbuf[: rows * r].reshape((rows, r), order="F")
    Z = Zbuf[: r * cols].reshape((r, cols), order="F")
    return r, Cm, Z


def kernel_matrix(rows, cols, cplx, seed=0, sep=1.5):
    """Smooth interaction between two separated point clouds: numerically low rank (like a far-field front block)."""
    rng = np.random.default_rng(seed)
    x = rng.random((rows, 3))
    y = rng.random((cols, 3)) + np.array([sep, 0.0, 0.0])
    d = np.linalg.norm(x[:, None, :] - y[None, :, :], axis=2)
    K = 1.0 / d
    if cplx:
        K = K * np.exp(1j * 3.0 * d)
    return K


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("rows,cols", [(300, 200), (1000, 777), (257, 1500)])
@pytest.mark.parametrize("rtol", [1e-2, 1e-6])
def test_lowrank_kernel_matrix(hs, rows, cols, cplx, rtol):
    X = kernel_matrix(rows, cols, cplx, seed=rows + cols)
    r, Cm, Z = compress(hs, X, 0.0, rtol)
    s = np.linalg.svd(X, compute_uv=False)
    r_svd = int(np.sum(s > rtol * s[0]))
    err = np.linalg.norm(X - Cm @ Z, 2) / s[0]
    assert err < 50 * rtol, (r, r_svd, err)
    import scipy.linalg as sla

    d = np.abs(np.diag(sla.qr(X.conj().T, mode="r", pivoting=True)[0]))  # the |R_jj| of the pivoted QR the reference's pqrfact stops on
    r_qr = int(np.sum(d > rtol * d[0]))
    print(f"{rows}x{cols} cplx={cplx} rtol={rtol:g}: rank {r} (pivoted QR {r_qr}, SVD {r_svd}), err {err:.2e}")
    assert 0.8 * r_qr - 2 <= r <= 1.15 * r_qr + 4, (r, r_qr, r_svd)
    assert np.max(np.abs(Cm)) < 8.0  # interpolation factor: tournament pivoting keeps |L| modest (not <= 1 like full partial pivoting)


def test_lowrank_exact_rank_and_full_rank(hs):
    rng = np.random.default_rng(5)
    A = rng.standard_normal((400, 37)) @ rng.standard_normal((37, 300))
    r, Cm, Z = compress(hs, A, 0.0, 1e-10)
    assert 37 <= r <= 37 + 8
    assert np.linalg.norm(A - Cm @ Z) / np.linalg.norm(A) < 1e-8
    B = rng.standard_normal((150, 120))  # full rank: the sketch must grow to min(rows, cols)
    r, Cm, Z = compress(hs, B, 0.0, 1e-12, kinit=32)
    assert r == 120
    assert np.linalg.norm(B - Cm @ Z) / np.linalg.norm(B) < 1e-9
    Zr = np.zeros((60, 50))
    r, Cm, Z = compress(hs, Zr, 1e-8, 1e-8)
    assert r == 0


@pytest.mark.parametrize("cplx", [False, True])
def test_lowrank_residual_norm_order_against_pivoted_qr(hs, cplx):
    """The blocked column-pivoted QR order (hs_hss_qr_order(1), include/hs_hss.h) on one block: rank within 5 % + 2 of scipy's pivoted QR at the
    same threshold (the default order is allowed 15 % + 4 above), same error level."""
    import scipy.linalg as sla

    X = kernel_matrix(1000, 777, cplx, seed=11)
    d = np.abs(np.diag(sla.qr(X.conj().T, mode="r", pivoting=True)[0]))
    s0 = np.linalg.norm(X, 2)
    try:
        for rtol in (1e-2, 1e-6, 1e-10):
            r_qr = int(np.sum(d > rtol * d[0]))
            hs.hss.qr_order("norm")
            r, Cm, Z = compress(hs, X, 0.0, rtol)
            err = np.linalg.norm(X - Cm @ Z, 2) / s0
            print(f"cplx={cplx} rtol={rtol:g}: rank {r} (pivoted QR {r_qr}) err {err:.2e} max|C| {np.max(np.abs(Cm)):.2f}")
            assert abs(r - r_qr) <= 0.05 * r_qr + 2, (r, r_qr)
            assert err < 50 * rtol
            assert np.max(np.abs(Cm)) < 8.0  # (a pivoted QR bounds |T| by 2^(r-j) only; in practice O(1) / theta)
    finally:
        hs.hss.qr_order("lu")
