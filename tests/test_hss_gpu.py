"""GPU parity tests of the device HSS module (include/hs_hss.h) against the CPU oracle (oracle/hs_hss.py).

HssMatrices.jl is absent from the reference tree: PARITY UNPINNED against the Julia package.  What is checked:
the device generators, fed to the ORACLE's matvec / expansion / elimination, reproduce the matrix to the
tolerance; the device product and the device solve agree with the oracle's on the same generators to round-off;
ranks agree with the oracle's own compression of the same matrix within a small margin."""
import numpy as np
import pytest

from oracle import hs_hss as HS
from test_hss_oracle import kernel_matrix, schur_of_separator

pytestmark = pytest.mark.gpu


def to_oracle(Hd):
    """Oracle `Hss` object holding the DEVICE generators."""
    n = Hd.shape[0]
    nodes = []
    for i in range(Hd.num_nodes):
        d = Hd.node(i)
        x = HS.HssNode(d["lo"], d["hi"], d["level"])
        x.left, x.right, x.m, x.r = d["left"], d["right"], d["m"], d["r"]
        if i != 0:
            x.p, x.T = d["p"].astype(np.int64), d["T"]
        x.D, x.B12, x.B21 = d["D"], d["B12"], d["B21"]
        nodes.append(x)
    for i, x in enumerate(nodes):
        for c in (x.left, x.right):
            if c >= 0:
                nodes[c].parent = i
    return HS.Hss(n, nodes, Hd.dtype)


def check_tree(Ho, n, leafsize):
    for i, x in enumerate(Ho.nodes):
        if x.left < 0:
            assert x.hi - x.lo <= leafsize and x.m == x.hi - x.lo
        else:
            l, r = Ho.nodes[x.left], Ho.nodes[x.right]
            assert (l.lo, r.hi, l.hi) == (x.lo, x.hi, r.lo) and x.m == l.r + r.r
        if i:
            assert sorted(x.p.tolist()) == list(range(x.m)) and 1 <= x.r <= x.m and x.T.shape == (x.m - x.r, x.r)
    assert Ho.nodes[0].lo == 0 and Ho.nodes[0].hi == n


@pytest.mark.parametrize("complex_", [False, True])
@pytest.mark.parametrize("n,leaf,tol", [(500, 40, 1e-4), (1200, 64, 1e-8)])
def test_compress_mul_ldiv_against_oracle(hs, complex_, n, leaf, tol):
    A = kernel_matrix(n, complex_)
    Hd = hs.hss.compress(A, leafsize=leaf, atol=tol, rtol=tol, kest=32)
    Ho = to_oracle(Hd)
    check_tree(Ho, n, leaf)
    Fh = HS.hss_full(Ho)
    err = np.linalg.norm(Fh - A) / np.linalg.norm(A)
    assert err < 100 * tol, err
    # ranks: the oracle's own compression of the same matrix (QR-based IDs) is the yardstick
    ro = HS.hssrank(HS.compress(A, leafsize=leaf, atol=tol, rtol=tol, kest=32, level_scale=0.5))
    # (the device detects ranks from the pivots of a sketched LU, the oracle from a pivoted QR: same slack as the low-rank
    # Gauss transforms, tests/test_compressed_gpu.py)
    assert Hd.rank == HS.hssrank(Ho) and Hd.rank <= 1.15 * ro + 4, (Hd.rank, ro)
    rng = np.random.default_rng(5)
    X = rng.standard_normal((n, 3)) + (1j * rng.standard_normal((n, 3)) if complex_ else 0)
    # product: device kernels vs the oracle's matvec on the same generators (round-off only)
    Yd = Hd @ X
    Yo = HS.hss_matvec(Ho, X)
    assert np.linalg.norm(Yd - Yo) <= 1e-12 * np.linalg.norm(Fh) * np.linalg.norm(X)
    y1 = Hd @ X[:, 0]
    assert y1.shape == (n,) and np.allclose(y1, Yd[:, 0])
    # elimination + solve: inverse of the HSS matrix (not of A)
    Zd = Hd.ldiv(X)
    Zo = HS.rs_solve(HS.rs_factor(Ho), X)
    ref = np.linalg.solve(Fh, X)
    assert np.linalg.norm(Zo - ref) / np.linalg.norm(ref) < 1e-9
    assert np.linalg.norm(Zd - ref) / np.linalg.norm(ref) < 1e-9
    assert np.linalg.norm(Zd - np.linalg.solve(A, X)) / np.linalg.norm(ref) < 1e4 * tol
    assert Hd.times()["compress_s"] > 0 and Hd.times()["factor_s"] > 0


def test_first_split_single_leaf_and_adaptivity(hs):
    A = kernel_matrix(300)
    H = hs.hss.compress(A, hs.hss.bisection_cluster((100, 300), leafsize=50), atol=1e-8, rtol=1e-8)
    Ho = to_oracle(H)
    assert (Ho.nodes[1].lo, Ho.nodes[1].hi, Ho.nodes[2].hi) == (0, 100, 300)
    assert np.linalg.norm(HS.hss_full(Ho) - A) / np.linalg.norm(A) < 1e-6
    # one leaf: H = D
    B = kernel_matrix(48)
    H1 = hs.hss.compress(B, leafsize=64)
    assert H1.num_nodes == 1 and H1.rank == 0
    x = np.arange(48.0)
    assert np.allclose(H1 @ x, B @ x) and np.allclose(H1.ldiv(x), np.linalg.solve(B, x))
    # 8 samples cannot carry the ranks at 1e-10: the sample count must have grown
    H2 = hs.hss.randcompress_adaptive(kernel_matrix(256, seed=3), leafsize=32, atol=1e-10, rtol=1e-10, kest=8)
    assert H2.samples > 8
    assert np.linalg.norm(H2.full() - kernel_matrix(256, seed=3)) / np.linalg.norm(kernel_matrix(256, seed=3)) < 1e-7


def test_schur_complement_of_a_separator(hs):
    """What `S` of a compressed front is (factorization.jl:56-57,109-110): compressible at the front tolerances, and the
    HSS solve is a good preconditioner-quality inverse."""
    S = schur_of_separator(32)
    for tol, rmax in ((1e-2, 16), (1e-6, 32)):
        H = hs.hss.compress(S, leafsize=8, atol=tol, rtol=tol, kest=16)
        assert np.linalg.norm(H.full() - S) / np.linalg.norm(S) < 100 * tol
        assert H.rank <= rmax
        b = np.ones(S.shape[0])
        x = H.ldiv(b)
        assert np.linalg.norm(S @ x - b) / np.linalg.norm(b) < 1e3 * tol


def test_large_block_timing(hs):
    """A 4096 x 4096 kernel block: records compress / factor times (printed with -s), solution checked."""
    n = 4096
    A = kernel_matrix(n)
    H = hs.hss.compress(A, leafsize=128, atol=1e-6, rtol=1e-6, kest=64)
    b = np.ones(n)
    x = H.ldiv(b)
    assert np.linalg.norm(A @ x - b) / np.linalg.norm(b) < 1e-3
    print("hss 4096: rank", H.rank, "samples", H.samples, H.times())


def test_errors(hs):
    with pytest.raises(hs.DimensionMismatch):
        hs.hss.compress(np.zeros((3, 4)))
    with pytest.raises(ValueError):
        hs.hss.compress(np.eye(8), leafsize=0)


def test_blocks_without_coupling(hs):
    """Off-diagonal blocks that are exactly (or numerically) zero: a node then keeps one nominal skeleton position with T = 0."""
    K = kernel_matrix(200)
    Z = np.zeros((200, 200))
    A = np.block([[K, Z], [Z, 2.0 * K]])
    H = hs.hss.compress(A, hs.hss.bisection_cluster((200, 400), leafsize=50), atol=1e-8, rtol=1e-8, kest=32)
    assert H._info(1)["r"] == 1 and H._info(2)["r"] == 1
    assert np.linalg.norm(H.full() - A) / np.linalg.norm(A) < 1e-6
    assert np.linalg.norm(H.expand() - H.full()) / np.linalg.norm(A) < 1e-12  # the expansion kernels against products with the identity
    b = np.arange(400.0)
    assert np.linalg.norm(H.ldiv(b) - np.linalg.solve(A, b)) / np.linalg.norm(b) < 1e-7
    D = np.diag(np.linspace(1.0, 3.0, 300))  # every sample block is exactly zero
    Hd = hs.hss.compress(D, leafsize=50, atol=1e-8, rtol=1e-8, kest=16)
    assert Hd.rank == 1
    assert np.allclose(Hd.full(), D)
    assert np.allclose(Hd.ldiv(b[:300]), b[:300] / np.diag(D))


def test_device_pointer_abi(hs):
    """`where = 1`: matrix, right-hand sides and results stay in HBM (what a device-resident caller such as the elimination uses)."""
    import ctypes as C

    import torch

    n, q = 700, 5
    A = kernel_matrix(n)
    dA = torch.from_numpy(np.asfortranarray(A).T.copy()).cuda()  # column-major bytes of A
    L = hs._lib.lib()
    o = hs._lib.hs_hss_options(64, 0, 1e-8, 1e-8, 32, 8, 123, 0.5)
    h = C.c_void_p()
    hs._lib.check(L.hs_hss_compress_ex_d(n, C.c_void_p(dA.data_ptr()), n, 1, None, C.byref(o), None, C.byref(h)))
    try:
        X = np.random.default_rng(2).standard_normal((n, q))
        dX = torch.from_numpy(X.T.copy()).cuda()
        dY = torch.zeros_like(dX)
        hs._lib.check(L.hs_hss_mul(h, C.c_void_p(dX.data_ptr()), n, C.c_void_p(dY.data_ptr()), n, q, 1))
        Y = dY.cpu().numpy().T
        assert np.linalg.norm(Y - A @ X) / np.linalg.norm(A @ X) < 1e-6
        hs._lib.check(L.hs_hss_ldiv(h, C.c_void_p(dY.data_ptr()), n, q, 1))  # in place: H^-1 (H X) = X
        assert np.linalg.norm(dY.cpu().numpy().T - X) / np.linalg.norm(X) < 1e-9
        assert L.hs_hss_size(h) == n and L.hs_hss_rank(h) > 0 and L.hs_hss_num_nodes(h) >= 3
        # argument errors come back as codes with a message, never as a crash
        assert L.hs_hss_mul(h, C.c_void_p(dX.data_ptr()), n, C.c_void_p(dX.data_ptr()), n, q, 1) == hs._lib.HS_ERR_ARGUMENT
        assert L.hs_hss_ldiv(h, None, n, q, 1) == hs._lib.HS_ERR_ARGUMENT
        assert L.hs_hss_ldiv(h, C.c_void_p(dY.data_ptr()), n, 0, 1) == 0  # no right-hand side: nothing to do
        bad = (C.c_int64 * 8)()
        assert L.hs_hss_node_info(h, 10**6, bad) == hs._lib.HS_ERR_ARGUMENT
    finally:
        L.hs_hss_free(h)
    h2 = C.c_void_p()
    assert L.hs_hss_compress_ex_d(0, C.c_void_p(dA.data_ptr()), n, 1, None, C.byref(o), None, C.byref(h2)) == hs._lib.HS_ERR_ARGUMENT
    perm = (C.c_int64 * n)(*([0] * n))  # not a permutation
    assert L.hs_hss_compress_ex_d(n, C.c_void_p(dA.data_ptr()), n, 1, perm, C.byref(o), None, C.byref(h2)) == hs._lib.HS_ERR_ARGUMENT


def test_permuted_matrix(hs):
    """H ~= A[perm, perm]: a scrambled kernel matrix compresses as well as the ordered one when the permutation is supplied."""
    n = 600
    K = kernel_matrix(n)
    q = np.random.default_rng(7).permutation(n)
    inv = np.argsort(q)
    A = K[np.ix_(inv, inv)]  # A[q][:, q] == K
    H = hs.hss.compress(A, leafsize=50, atol=1e-8, rtol=1e-8, kest=32, perm=q)
    H0 = hs.hss.compress(K, leafsize=50, atol=1e-8, rtol=1e-8, kest=32)
    assert H.rank <= H0.rank + 6
    assert np.linalg.norm(H.full() - A) / np.linalg.norm(A) < 1e-6  # products and solves keep the caller's order
    assert np.linalg.norm(H.expand() - K) / np.linalg.norm(K) < 1e-6  # `Matrix(H)` in the matrix's own order: A[q][:, q]
    b = np.arange(float(n))
    assert np.linalg.norm(H.ldiv(b) - np.linalg.solve(A, b)) / np.linalg.norm(b) < 1e-7
    Hbad = hs.hss.compress(A, leafsize=50, atol=1e-8, rtol=1e-8, kest=32)  # without it: ranks close to the block sizes
    assert Hbad.rank > 2 * H.rank


@pytest.mark.parametrize("complex_", [False, True])
def test_operator_with_low_rank_update(hs, complex_):
    """C6 on the device: the Schur complement `S = Abb - Abi*R` with a low-rank R is compressed from products and entries, never
    formed (`_sample_schur!` / `_getindex_schur`, factorization.jl:238-249).  Here B - C*M*Z with a smooth update of rank 24."""
    n, r1, r2 = 900, 24, 20
    rng = np.random.default_rng(11)
    B = kernel_matrix(n, complex_)
    x = np.linspace(0.0, 1.0, n)
    Cm = np.stack([np.cos(j * x) for j in range(r1)], axis=1) * 2.0
    Z = np.stack([np.sin((j + 1) * x) for j in range(r2)], axis=0)
    M = rng.standard_normal((r1, r2)) + (1j * rng.standard_normal((r1, r2)) if complex_ else 0)
    S = B - Cm @ M @ Z
    tol = 1e-7
    q = None if not complex_ else np.arange(n)[::-1].copy()  # also through a permutation (reversal keeps the blocks smooth)
    H = hs.hss.compress_lowrank_update(B, Cm, M, Z, leafsize=64, atol=tol, rtol=tol, kest=32, perm=q)
    assert np.linalg.norm(H.full() - S) / np.linalg.norm(S) < 200 * tol
    H0 = hs.hss.compress(S, leafsize=64, atol=tol, rtol=tol, kest=32, perm=q)  # the formed matrix: same ranks up to the draw
    assert abs(H.rank - H0.rank) <= 8, (H.rank, H0.rank)
    b = rng.standard_normal(n)
    assert np.linalg.norm(S @ H.ldiv(b) - b) / np.linalg.norm(b) < 1e4 * tol
    Ho = to_oracle(H)
    assert np.linalg.norm(HS.hss_full(Ho)[np.ix_(np.argsort(q), np.argsort(q))] - S if q is not None else HS.hss_full(Ho) - S) / np.linalg.norm(S) < 200 * tol
    with pytest.raises(hs.DimensionMismatch):
        hs.hss.compress_lowrank_update(B, Cm, M[:, :3], Z)


@pytest.mark.parametrize("complex_", [False, True])
def test_entry_access(hs, complex_):
    """`H[I, J]` on the device against the oracle's recursive getindex on the same generators, and against the matrix."""
    n = 900
    A = kernel_matrix(n, complex_, seed=4)
    H = hs.hss.compress(A, hs.hss.bisection_cluster((333, n), leafsize=48), atol=1e-9, rtol=1e-9, kest=32)
    Ho = to_oracle(H)
    rng = np.random.default_rng(9)
    for ni, nj in ((1, 1), (37, 120), (300, 5), (n, 40)):
        I = rng.integers(0, n, ni) if ni < n else rng.permutation(n)
        J = rng.integers(0, n, nj)
        G = H.getindex(I, J)
        Go = HS.hss_getindex(Ho, I, J)
        assert G.shape == (ni, nj)
        assert np.abs(G - Go).max() <= 1e-12 * np.abs(A).max() * 50, (ni, nj, np.abs(G - Go).max())
        assert np.abs(G - A[np.ix_(I, J)]).max() <= 1e-6 * np.abs(A).max()
    # through a permutation: indices are the caller's
    q = rng.permutation(n)
    inv = np.argsort(q)
    Ap = A[np.ix_(inv, inv)]
    Hp = hs.hss.compress(Ap, leafsize=48, atol=1e-9, rtol=1e-9, kest=32, perm=q)
    I, J = rng.integers(0, n, 64), rng.integers(0, n, 77)
    assert np.abs(Hp.getindex(I, J) - Ap[np.ix_(I, J)]).max() <= 1e-6 * np.abs(A).max()
    with pytest.raises(ValueError):
        H.getindex([n], [0])
    assert H.getindex([], [1, 2]).shape == (0, 2)


@pytest.mark.parametrize("complex_", [False, True])
def test_transposed_product_and_block_views(hs, complex_):
    """`H^T X`, and `H.A11` / `H.A22` as HSS matrices sharing H's generators (what a parent front reads from a child's S)."""
    n, n1 = 800, 300
    A = kernel_matrix(n, complex_, seed=6)
    H = hs.hss.compress(A, hs.hss.bisection_cluster((n1, n), leafsize=50), atol=1e-9, rtol=1e-9, kest=32)
    Fh = H.full()
    rng = np.random.default_rng(12)
    X = rng.standard_normal((n, 4)) + (1j * rng.standard_normal((n, 4)) if complex_ else 0)
    assert np.linalg.norm(H.rmatmul_t(X) - Fh.T @ X) <= 1e-12 * np.linalg.norm(Fh) * np.linalg.norm(X)
    assert np.linalg.norm(H @ X - Fh @ X) <= 1e-12 * np.linalg.norm(Fh) * np.linalg.norm(X)  # the plain product is untouched
    H11, H22 = H.block(0), H.block(1)
    assert H11.shape == (n1, n1) and H22.shape == (n - n1, n - n1)
    assert np.allclose(H11.full(), Fh[:n1, :n1], atol=1e-12 * np.abs(Fh).max())
    assert np.allclose(H22.full(), Fh[n1:, n1:], atol=1e-12 * np.abs(Fh).max())
    I, J = rng.integers(0, n - n1, 50), rng.integers(0, n - n1, 60)
    assert np.allclose(H22.getindex(I, J), Fh[n1:, n1:][np.ix_(I, J)], atol=1e-12 * np.abs(Fh).max())
    b = X[:n1]
    assert np.linalg.norm(H11.ldiv(b) - np.linalg.solve(Fh[:n1, :n1], b)) / np.linalg.norm(b) < 1e-9
    assert np.linalg.norm(H11.rmatmul_t(b) - Fh[:n1, :n1].T @ b) <= 1e-12 * np.linalg.norm(Fh) * np.linalg.norm(b)
    assert H11.rank <= H.rank and H11.block(0).shape[0] == (n1 + 1) // 2  # views nest
    del H11, H22
    assert np.allclose(H.full(), Fh)  # the parent is intact after its views are gone
    with pytest.raises(RuntimeError, match="turned into a leaf"):
        hs.hss.compress(kernel_matrix(30), leafsize=64).block(0)


@pytest.mark.parametrize("complex_", [False, True])
def test_expanded_bases_and_offdiagonal_generators(hs, complex_):
    """`generators(S.A11)`, `S.B12`, `S.B21` (factorization.jl:129-137): A12 = U1 B12 U2^T from the expanded bases, against the oracle."""
    n, n1 = 700, 260
    A = kernel_matrix(n, complex_, seed=8)
    H = hs.hss.compress(A, hs.hss.bisection_cluster((n1, n), leafsize=40), atol=1e-9, rtol=1e-9, kest=32)
    Fh = H.full()
    U1, B12, U2, B21 = H.offdiag()
    assert U1.shape == (n1, B12.shape[0]) and U2.shape == (n - n1, B12.shape[1])
    tol = 1e-12 * np.abs(Fh).max() * 50
    assert np.abs(U1 @ B12 @ U2.T - Fh[:n1, n1:]).max() < tol
    assert np.abs(U2 @ B21 @ U1.T - Fh[n1:, :n1]).max() < tol
    Ho = to_oracle(H)
    O1, _, O2, _ = HS.hss_offdiag(Ho)
    assert np.abs(U1 - O1).max() < 1e-12 * max(1.0, np.abs(O1).max()) * 50 and np.abs(U2 - O2).max() < 1e-12 * max(1.0, np.abs(O2).max()) * 50
    # a deeper node: its basis reproduces the block row against the rest through the node's skeleton rows (interpolation property)
    k = 5
    d = H._info(k)
    Uk = H.basis(k)
    assert Uk.shape == (d["hi"] - d["lo"], d["r"])
    with pytest.raises(ValueError):
        H.basis(0)


@pytest.mark.parametrize("complex_", [False, True])
def test_prune_leaves_and_equilibrate_clusters(hs, complex_):
    """`prune_leaves!` / `compatible` / `_equilibrate_clusters` (src/factorization.jl:143-168): pruning merges the deepest leaves without
    changing the matrix the generators represent; two Schur complements of different depth end up on compatible trees."""
    n = 700
    A = kernel_matrix(n, complex_)
    perm = np.random.default_rng(3).permutation(n)
    H = hs.hss.compress(A, leafsize=40, atol=1e-9, rtol=1e-9, kest=48, perm=perm)
    F0 = H.full()
    Hp = H.prune_leaves()
    assert Hp.depth == H.depth - 1 and Hp.num_nodes < H.num_nodes
    assert np.linalg.norm(Hp.full() - F0) <= 1e-12 * np.linalg.norm(F0)  # the SAME matrix, leaf blocks formed from the generators
    I, J = np.arange(0, n, 7), np.arange(3, n, 11)
    assert np.linalg.norm(Hp.getindex(I, J) - F0[np.ix_(I, J)]) <= 1e-11 * np.linalg.norm(F0)
    b = np.random.default_rng(4).standard_normal(n) + (1j if complex_ else 0)
    assert np.linalg.norm(Hp.ldiv(b) - np.linalg.solve(F0, b)) <= 1e-8 * np.linalg.norm(b)
    Hpp = Hp.prune_leaves()
    assert Hpp.depth == H.depth - 2 and np.linalg.norm(Hpp.full() - F0) <= 1e-12 * np.linalg.norm(F0)
    assert H.compatible(H) and not H.compatible(Hp)
    # two "Schur complements" with the forced [int | bnd] split whose A11 blocks have different depths
    S1 = hs.hss.compress(kernel_matrix(900, complex_), hs.hss.bisection_cluster((640, 900), leafsize=40), atol=1e-8, rtol=1e-8, kest=48)
    S2 = hs.hss.compress(kernel_matrix(500, complex_), hs.hss.bisection_cluster((160, 500), leafsize=40), atol=1e-8, rtol=1e-8, kest=48)
    assert not S1.block(0).compatible(S2.block(0))
    A1, A2 = hs.hss.equilibrate_clusters(S1, S2)
    assert A1.compatible(A2) and A1.shape == (640, 640) and A2.shape == (160, 160)
    assert np.linalg.norm(A1.full() - S1.block(0).full()) <= 1e-12 * np.linalg.norm(A1.full())
    # a matrix that is a single leaf cannot be pruned: the reference's error
    H1 = hs.hss.compress(kernel_matrix(30, complex_), leafsize=64)
    with pytest.raises(RuntimeError, match="turned into a leaf"):
        H1.prune_leaves()


@pytest.mark.parametrize("dtype", [np.float64, np.complex128])
def test_expand_equals_products_with_the_identity(hs, dtype):
    """`Matrix(H)` (hs_hss_expand: every basis once, then U_l*B12*U_r^T / U_r*B21*U_l^T per inner node) against H applied to the identity,
    for the matrix and for the block views a parent front reads (hs_hss_child)."""
    n = 700
    K = kernel_matrix(n, complex_=(dtype == np.complex128))
    H = hs.hss.compress(K, hs.hss.bisection_cluster((300, n), leafsize=60), atol=1e-9, rtol=1e-9, kest=32)
    F = H.full()
    assert np.linalg.norm(H.expand() - F) / np.linalg.norm(F) < 1e-12
    assert np.linalg.norm(F - K) / np.linalg.norm(K) < 1e-7
    for which, sl in ((0, slice(0, 300)), (1, slice(300, n))):
        V = H.block(which)
        assert np.linalg.norm(V.expand() - F[sl, sl]) / np.linalg.norm(F[sl, sl]) < 1e-12
    Hl = hs.hss.compress(K[:40, :40], leafsize=64, atol=1e-9, rtol=1e-9, kest=16)  # a single leaf
    assert np.allclose(Hl.expand(), K[:40, :40])


@pytest.mark.parametrize("complex_", [False, True])
def test_batched_compression_of_several_operators(hs, complex_):
    """hs_hss_compress_lru_multi: matrices of different sizes, one permuted, one with a low-rank update, one with a forced first split, compressed
    as ONE forest -- every result represents its own operator to the tolerance, with the rank of the one-at-a-time compression."""
    rng = np.random.default_rng(11)
    items, refs = [], []
    for b, n in enumerate((520, 700, 380)):
        K = kernel_matrix(n, complex_, seed=20 + b)
        it = dict(B=K)
        A = K
        if b == 0:  # scrambled: perm restores the order the kernel compresses in
            q = rng.permutation(n)
            inv = np.argsort(q)
            it = dict(B=K[np.ix_(inv, inv)], perm=q)
            A = it["B"]
        if b == 1:  # B - C*M*Z with smooth factors
            r = 7
            Cm = np.cos(np.outer(np.linspace(0, 1, n), np.arange(1, r + 1)))
            Z = np.sin(np.outer(np.arange(1, r + 1), np.linspace(0, 2, n)))
            M = rng.standard_normal((r, r))
            it.update(C=Cm, M=M, Z=Z)
            A = K - Cm @ M @ Z
        if b == 2:
            it["cl"] = (100, n, 48)
        items.append(it)
        refs.append(A)
    Hs = hs.hss.compress_lowrank_update_batch(items, leafsize=48, atol=1e-9, rtol=1e-9, kest=32)
    assert len(Hs) == 3
    for it, A, H in zip(items, refs, Hs):
        assert H.shape == A.shape
        assert np.linalg.norm(H.full() - A) / np.linalg.norm(A) < 1e-6
        b = rng.standard_normal(A.shape[0])
        assert np.linalg.norm(H.ldiv(b) - np.linalg.solve(A, b)) / np.linalg.norm(b) < 1e-6
        if it.get("C") is None:
            H1 = hs.hss.compress(it["B"], it.get("cl"), leafsize=48, atol=1e-9, rtol=1e-9, kest=32, perm=it.get("perm"))
        else:
            H1 = hs.hss.compress_lowrank_update(it["B"], it["C"], it["M"], it["Z"], leafsize=48, atol=1e-9, rtol=1e-9, kest=32)
        assert abs(H.rank - H1.rank) <= 6, (H.rank, H1.rank)
    # the block views a parent front reads work on a matrix of a batch too
    V = Hs[2].block(0)
    assert V.shape == (100, 100)
    assert np.linalg.norm(V.expand() - refs[2][:100, :100]) / np.linalg.norm(refs[2][:100, :100]) < 1e-6


def test_trim_releases_the_recycled_blocks(hs):
    """hs_hss_trim: the block caches hold memory between factorizations; trimming returns it and everything keeps working."""
    K = kernel_matrix(600)
    H = hs.hss.compress(K, leafsize=64, atol=1e-8, rtol=1e-8, kest=32)
    b = np.arange(600.0)
    x0 = H.ldiv(b)
    freed = hs.hss.trim()
    assert freed > 0
    assert hs.hss.trim() == 0  # nothing left to give back
    H2 = hs.hss.compress(K, leafsize=64, atol=1e-8, rtol=1e-8, kest=32)
    assert np.allclose(H2.ldiv(b), x0, rtol=1e-6, atol=1e-9)
    assert np.allclose(H.ldiv(b), x0)  # the generators a live matrix owns were never in the cache


@pytest.mark.parametrize("complex_", [False, True])
def test_pack_unpack_roundtrip(hs, complex_):
    """`hs_hss_pack` / `hs_hss_unpack`: the one-buffer form in which a Schur complement crosses ranks as an HssMatrix ("ship
    HSS generators instead of dense S"; src/factorization.jl:126-140 reads the children's S as HSS).  The unpacked matrix has the same tree and
    the same generators bit for bit: products, entries, block views, off-diagonal factors, expansion and the solve are identical."""
    rng = np.random.default_rng(5)
    n = 900
    A = kernel_matrix(n, complex_)
    perm = np.arange(n)[::-1].copy()  # a permutation travels with the matrix (this one keeps the blocks compressible)
    H = hs.hss.compress(A, leafsize=64, atol=1e-9, rtol=1e-9, kest=32, perm=perm)
    buf = H.pack()
    assert buf.dtype.itemsize == 1 and buf.numel() < 0.8 * A.nbytes  # generators, not the matrix
    G = hs.hss.HssMatrix.unpack(buf, complex_)
    del buf
    assert G.shape == H.shape and G.rank == H.rank and G.num_nodes == H.num_nodes and G.samples == H.samples
    for i in range(H.num_nodes):
        a, b = H.node(i), G.node(i)
        for k in ("lo", "hi", "level", "left", "right", "m", "r"):
            assert a[k] == b[k], (i, k)
        for k in ("p", "T", "D", "B12", "B21"):
            if a[k] is None:
                assert b[k] is None
            else:
                assert np.array_equal(a[k], b[k]), (i, k)
    X = rng.standard_normal((n, 3)) + (1j * rng.standard_normal((n, 3)) if complex_ else 0)
    assert np.array_equal(H.matmul(X), G.matmul(X))
    assert np.array_equal(H.rmatmul_t(X), G.rmatmul_t(X))
    I, J = rng.integers(0, n, 40), rng.integers(0, n, 50)
    assert np.array_equal(H.getindex(I, J), G.getindex(I, J))
    assert np.array_equal(H.full(), G.full())
    for which in (0, 1):
        assert np.array_equal(H.block(which).full(), G.block(which).full())
        Ch, Zh = H.offdiag_lowrank(which)
        Cg, Zg = G.offdiag_lowrank(which)
        assert np.array_equal(Ch, Cg) and np.array_equal(Zh, Zg)
    b = X[:, 0].copy()
    xh, xg = H.ldiv(b), G.ldiv(b)  # the factors do not travel: the receiver eliminates again, with the same arithmetic
    assert np.linalg.norm(xh - xg) <= 1e-12 * np.linalg.norm(xh)
    assert np.linalg.norm(A @ xg - b) / np.linalg.norm(b) < 1e-6
    # a truncated or foreign buffer is refused
    bad = H.pack()[:100]
    with pytest.raises(ValueError):
        hs.hss.HssMatrix.unpack(bad, complex_)
    with pytest.raises(ValueError):
        hs.hss.HssMatrix.unpack(H.pack(), not complex_)


@pytest.mark.parametrize("complex_", [False, True])
def test_residual_norm_order_is_a_pivoted_qr_and_agrees_with_the_default(hs, complex_):
    """hs_hss_qr_order(1): every window of the orthogonalisation is chosen by downdated residual norms and a pivot is accepted only while it is
    >= 0.5 x every residual outside the window -- the greedy order of `pqrfact` (src/factorization.jl:171-182).  The default (tournament order)
    must deliver the same ranks (+-3 % + 2) and the same errors (within 3x), from 1e-3 down to 1e-12; and the ranks of the greedy order are
    within 6 % + 3 of the oracle's, whose IDs are scipy's column-pivoted QR stopped at the same tolerance."""
    if complex_:  # points of a 2-D sheet (a separator) in recursive-bisection order, Helmholtz-like kernel: ranks grow like sqrt(n)
        m = 36
        gx, gy = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
        P = np.stack([gx.ravel(), gy.ravel()], 1).astype(float)

        def rb(idx):
            if len(idx) <= 16:
                return list(idx)
            c = P[idx]
            o = np.argsort(c[:, int(np.argmax(c.max(0) - c.min(0)))], kind="stable")
            h = len(idx) // 2
            return rb(idx[o[:h]]) + rb(idx[o[h:]])

        P = P[rb(np.arange(m * m))]
        R = np.sqrt(((P[:, None, :] - P[None, :, :]) ** 2).sum(-1))
        A = np.exp(1j * 0.7 * R) / (1.0 + R) + (4.0 + 1.0j) * np.eye(m * m)
    else:
        A = kernel_matrix(1500, False)
    nA = np.linalg.norm(A)
    assert hs.hss.qr_order() == "lu"
    try:
        for tol in (1e-3, 1e-6, 1e-9, 1e-12):
            out = {}
            for order in ("lu", "norm"):
                hs.hss.qr_order(order)
                H = hs.hss.compress(A, leafsize=64, atol=tol * 1e-3, rtol=tol, kest=64)
                out[order] = (H.rank, np.linalg.norm(H.full() - A) / nA)
            (r0, e0), (r1, e1) = out["lu"], out["norm"]
            print(f"complex={complex_} tol={tol:g}: lu rank {r0} err {e0:.2e} | norm rank {r1} err {e1:.2e}")
            assert abs(r0 - r1) <= 0.03 * r1 + 2, out
            assert e0 <= 3 * e1 + 1e-14 and e1 <= 3 * e0 + 1e-14, out
            assert e1 < 30 * tol + 1e-13, out
            if tol >= 1e-9:
                ro = HS.hssrank(HS.compress(A, leafsize=64, atol=tol * 1e-3, rtol=tol, kest=64, level_scale=0.5))
                assert abs(r1 - ro) <= 0.06 * ro + 3, (r1, ro)
    finally:
        hs.hss.qr_order("lu")
