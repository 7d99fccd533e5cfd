"""T2 kernel tests: each HIP kernel of the hot path against NumPy on the same seeded inputs, through
the C ABI test hooks (include/hs_kernels.h).  Tolerances: FP64 GEMM/LU, relative 1e-12 / 1e-10."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu


def _ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _rand(rng, shape, cplx):
    a = rng.standard_normal(shape)
    if cplx:
        a = a + 1j * rng.standard_normal(shape)
    return np.asfortranarray(a)


def run_gemm(hs, M, N, K, cplx, minus, pad=(0, 0, 0), seed=0):
    rng = np.random.default_rng(seed)
    lda, ldb, ldc = M + pad[0], K + pad[1], M + pad[2]
    A = _rand(rng, (lda, K), cplx)
    B = _rand(rng, (ldb, N), cplx)
    Cm = _rand(rng, (ldc, N), cplx)
    C0 = Cm.copy()
    L = hs._lib.lib()
    fn = L.hsk_gemm_z if cplx else L.hsk_gemm_d
    hs._lib.check(fn(M, N, K, _ptr(A), lda, _ptr(B), ldb, _ptr(Cm), ldc, int(minus), 0, None))
    prod = A[:M] @ B[:K]
    ref = C0[:M] - prod if minus else prod
    scale = np.abs(A[:M]) @ np.abs(B[:K]) + np.abs(C0[:M])
    err = np.max(np.abs(Cm[:M] - ref) / scale)
    assert err < 1e-13, (M, N, K, cplx, minus, err)
    if pad[2]:
        assert np.array_equal(Cm[M:], C0[M:])  # padding rows untouched


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize(
    "M,N,K",
    [(128, 128, 16), (128, 128, 64), (256, 128, 32), (64, 64, 64), (1, 1, 1), (17, 5, 3), (129, 127, 33), (300, 257, 100), (513, 64, 250), (31, 400, 16), (1000, 999, 7), (258, 300, 64), (2, 130, 32), (384, 200, 48), (130, 1, 16)],
)
def test_gemm_shapes(hs, M, N, K, cplx):
    run_gemm(hs, M, N, K, cplx, minus=True, pad=(2, 3, 1), seed=M + N + K)
    run_gemm(hs, M, N, K, cplx, minus=False, pad=(0, 0, 0), seed=M * 3 + K)


@pytest.mark.parametrize("M,N,K", [(64, 64, 2048), (64, 2048, 500), (40, 50, 1000), (64, 700, 333), (33, 129, 17), (1, 64, 64), (64, 65, 16), (7, 300, 4100)])
def test_gemm_skinny_tiles(hs, M, N, K):
    """Problem lists whose M <= 64 run the 64 x 64 / 64 x 128 tiles (gemm_probs_skinny_kernel): the grouped products of the HSS module."""
    run_gemm(hs, M, N, K, False, minus=True, pad=(2, 3, 1), seed=M + N + K)
    run_gemm(hs, M, N, K, False, minus=False, pad=(0, 0, 0), seed=M * 3 + K)


def test_gemm_mfma_layout_asymmetric(hs):
    """A = I with an asymmetric B exposes a transposed / permuted C write (cdna guide section 3)."""
    n = 128
    A = np.asfortranarray(np.eye(n))
    B = np.asfortranarray(np.arange(n * n, dtype=np.float64).reshape(n, n) % 251)
    Cm = np.zeros((n, n), order="F")
    L = hs._lib.lib()
    hs._lib.check(L.hsk_gemm_d(n, n, n, _ptr(A), n, _ptr(B), n, _ptr(Cm), n, 0, 0, None))
    assert np.array_equal(Cm, B)
    hs._lib.check(L.hsk_gemm_d(n, n, n, _ptr(B), n, _ptr(A), n, _ptr(Cm), n, 0, 0, None))
    assert np.array_equal(Cm, B)


def front_check(hs, count, ni, nb, cplx, seed=0, kind="randn"):
    rng = np.random.default_rng(seed)
    m = ni + nb
    Fs = np.zeros((count, m, m), dtype=np.complex128 if cplx else np.float64)
    for k in range(count):
        F = _rand(rng, (m, m), cplx)
        if kind == "spd":
            F = F @ F.conj().T + m * np.eye(m)
        elif kind == "needs_pivot":
            F[np.arange(min(ni, m)), np.arange(min(ni, m))] = 0.0  # zero diagonal: unpivoted LU breaks down
        Fs[k] = F
    Fcol = np.ascontiguousarray(Fs.transpose(0, 2, 1))  # each front column-major
    LF = np.zeros((count, ni, m), dtype=Fs.dtype)
    UR = np.zeros((count, nb, ni), dtype=Fs.dtype)
    SB = np.zeros((count, nb, nb), dtype=Fs.dtype)
    rperm = np.zeros((count, ni), dtype=np.int64)
    info = np.zeros(count, dtype=np.int64)
    L = hs._lib.lib()
    fn = L.hsk_front_factor_z if cplx else L.hsk_front_factor_d
    hs._lib.check(fn(count, ni, nb, _ptr(Fcol), _ptr(LF), _ptr(UR), _ptr(SB), rperm.ctypes.data_as(C.POINTER(C.c_int64)),
                     info.ctypes.data_as(C.POINTER(C.c_int64)), None))
    worst = 0.0
    for k in range(count):
        F = Fs[k]
        lf = LF[k].T  # m x ni
        ur = UR[k].T  # ni x nb
        sb = SB[k].T
        assert info[k] == 0
        rp = rperm[k]
        assert sorted(rp.tolist()) == list(range(ni))
        Aii, Aib, Abi, Abb = F[:ni, :ni], F[:ni, ni:], F[ni:, :ni], F[ni:, ni:]
        Lm = np.tril(lf[:ni], -1) + np.eye(ni)
        Um = np.triu(lf[:ni])
        nrm = np.linalg.norm(F)
        e1 = np.linalg.norm(Lm @ Um - Aii[rp]) / nrm  # P*Aii = L*U
        e2 = np.linalg.norm(lf[ni:] @ Um - Abi) / nrm  # Lbi * U = Abi
        e3 = np.linalg.norm(Lm @ ur - Aib[rp]) / nrm  # L * Uib = P*Aib
        Sref = Abb - Abi @ np.linalg.solve(Aii, Aib) if ni else Abb
        e4 = np.linalg.norm(sb - Sref) / max(np.linalg.norm(Sref), 1e-300) if nb else 0.0
        worst = max(worst, e1, e2, e3)
        cond = np.linalg.cond(Aii) if ni else 1.0
        assert e1 < 1e-13 and e2 < 1e-13 and e3 < 1e-13, (ni, nb, cplx, kind, e1, e2, e3)
        assert e4 < 1e-13 * max(cond, 10.0), (ni, nb, cplx, kind, e4, cond)
        # pivot growth sanity: |L| <= ~1 is NOT guaranteed by tournament pivoting, but must stay modest
        assert np.max(np.abs(np.tril(lf[:ni], -1))) < 50.0
    return worst


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize(
    "count,ni,nb", [(1, 32, 0), (1, 32, 32), (3, 64, 40), (2, 33, 7), (1, 1, 1), (2, 5, 70), (1, 100, 129), (2, 128, 129), (1, 300, 200), (1, 257, 0), (1, 600, 333), (4, 96, 96)]
)
def test_front_factor(hs, count, ni, nb, cplx):
    front_check(hs, count, ni, nb, cplx, seed=ni * 7 + nb)


@pytest.mark.parametrize("cplx", [False, True])
def test_front_factor_needs_pivoting(hs, cplx):
    front_check(hs, 2, 200, 50, cplx, seed=5, kind="needs_pivot")
    front_check(hs, 1, 1030, 100, cplx, seed=6, kind="needs_pivot")  # several tournament rounds (ni > 256*4)


def test_front_factor_spd_large(hs):
    front_check(hs, 1, 1500, 700, False, seed=11, kind="spd")


def test_front_singular_flag(hs):
    ni, nb = 40, 8
    m = ni + nb
    rng = np.random.default_rng(3)
    F = rng.standard_normal((m, m))
    F[:ni, 5] = 0.0  # exactly singular interior block
    Fcol = np.ascontiguousarray(F.T)[None]
    info = np.zeros(1, dtype=np.int64)
    L = hs._lib.lib()
    hs._lib.check(L.hsk_front_factor_d(1, ni, nb, _ptr(Fcol), None, None, None, None, info.ctypes.data_as(C.POINTER(C.c_int64)), None))
    assert info[0] != 0


def test_mfma_peak_probe(hs):
    tf = hs._lib.lib().hsk_mfma_f64_peak(2, 20000)
    assert tf > 10.0, tf


def test_flow_exchange_primitive_round_trip(hs):
    """hsk_flow_pingpong_us: two workgroups of one launch answer each other through agent-scope atomic stores and polled loads -- the exchange
    primitive of the dataflow sweeps of ldiv! (kernels_solve_wide.hip).  Every poll is bounded, so the call returns whatever happens; a round trip
    is ~1 us on MI355X (DESIGN.md section 4a'), same XCD (peer 8) or not (peer 1)."""
    L = hs._lib.lib()
    for peer in (1, 8, 255):
        us = L.hsk_flow_pingpong_us(peer, 2000)
        print(f"peer {peer}: {us:.3f} us per round trip")
        assert 0.05 < us < 50.0, (peer, us)
    assert L.hsk_flow_pingpong_us(0, 10) < 0  # argument check
