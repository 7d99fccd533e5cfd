"""Operators that are never formed (include/hs_hss.h: hs_hss_blockop): two diagonal HSS blocks + sparse couplings, as
`_assemble_blocks` builds Aii / Abb from the children's HSS Schur complements (src/factorization.jl:126-140), their products,
the batched entry access behind them and the matrix-free HSS compression (`randcompress_adaptive` on the operator, :110,228-249).
HssMatrices.jl is absent from the reference tree: PARITY UNPINNED; checked against the same operator formed densely on the host."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def kernel(n, rng, cplx=False, shift=0.0):
    pts = np.sort(rng.random(n))
    K = 1.0 / (1.0 + 30.0 * np.abs(pts[:, None] - pts[None, :])) + 0.1 * n * np.eye(n)
    if cplx:
        K = K * np.exp(1j * 3.0 * np.abs(pts[:, None] - pts[None, :]))
    return K + shift


def relerr(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def make_op(hs, rng, cplx, n1, n2, views):
    """H1, H2 (optionally views of a larger HSS matrix, as a parent front sees its children's S.A11) + a sparse matrix on 3*(n1+n2) global ids."""
    if views:
        S1 = hs.hss.compress(kernel(n1 + 70, rng, cplx), hs.hss.bisection_cluster((n1, n1 + 70), leafsize=32), atol=1e-10, rtol=1e-10, kest=48)
        S2 = hs.hss.compress(kernel(n2 + 50, rng, cplx), hs.hss.bisection_cluster((n2, n2 + 50), leafsize=32), atol=1e-10, rtol=1e-10, kest=48)
        H1, H2 = S1.block(0), S2.block(0)
    else:
        p1 = rng.permutation(n1)
        H1 = hs.hss.compress(kernel(n1, rng, cplx), leafsize=32, atol=1e-10, rtol=1e-10, kest=48, perm=p1).view()  # tree order, permutation dropped
        H2 = hs.hss.compress(kernel(n2, rng, cplx), leafsize=32, atol=1e-10, rtol=1e-10, kest=48)
    n = n1 + n2
    ng = 3 * n
    A = sp.random(ng, ng, density=4.0 / ng, random_state=np.random.RandomState(5), format="csc")
    A = A + sp.diags(np.ones(ng))
    if cplx:
        A = A + 1j * sp.random(ng, ng, density=2.0 / ng, random_state=np.random.RandomState(6), format="csc")
    gid = rng.permutation(ng)[:n]
    As = hs.hss.SparseDevice(A)
    op = hs.hss.BlockOperator(H1, H2, gid, As)
    Ad = A.toarray()
    D = np.zeros((n, n), dtype=np.complex128 if cplx else np.float64)
    D[:n1, :n1] = H1.full()
    D[n1:, n1:] = H2.full()
    D[:n1, n1:] = Ad[np.ix_(gid[:n1], gid[n1:])]
    D[n1:, :n1] = Ad[np.ix_(gid[n1:], gid[:n1])]
    return op, D


@pytest.mark.parametrize("cplx", [False, True])
@pytest.mark.parametrize("views", [False, True])
def test_blockop_products(hs, cplx, views):
    rng = np.random.default_rng(11)
    op, D = make_op(hs, rng, cplx, 150, 119, views)
    X = rng.standard_normal((D.shape[0], 5)) + (1j * rng.standard_normal((D.shape[0], 5)) if cplx else 0)
    assert relerr(op.matmul(X), D @ X) < 1e-12
    assert relerr(op.matmul(X, trans=True), D.T @ X) < 1e-12
    # the scratch map is handed back clean: a second operator on the same sparse matrix works
    assert int((op.As.lpos != -1).sum()) == 0


@pytest.mark.parametrize("cplx", [False, True])
def test_blockop_compression_matches_the_formed_operator(hs, cplx):
    """H = compress(Op - C M Z) from products and entries only, with a permutation and a forced first split: expands to the formed matrix."""
    rng = np.random.default_rng(12)
    n1, n2 = 170, 131
    op, D = make_op(hs, rng, cplx, n1, n2, True)
    n = n1 + n2
    # make the sparse couplings matter: they are O(1) entries next to O(n) diagonal blocks
    r1, r2 = 7, 5
    mk = lambda *s: rng.standard_normal(s) + (1j * rng.standard_normal(s) if cplx else 0)  # noqa: E731
    Cm, M, Z = mk(n, r1), mk(r1, r2), mk(r2, n)
    perm = rng.permutation(n)
    tol = 1e-8
    for upd, ref in ((None, D), ((Cm, M, Z), D - Cm @ M @ Z)):
        H = op.compress(hs.hss.bisection_cluster((100, n), leafsize=40), atol=tol, rtol=tol, kest=64, perm=perm, update=upd)
        assert relerr(H.full(), ref) < 50 * tol
        b = mk(n)
        assert relerr(H.ldiv(b), np.linalg.solve(ref, b)) < 1e-5
        assert int((op.As.lpos != -1).sum()) == 0


def test_offdiag_lowrank_and_whole_view(hs):
    rng = np.random.default_rng(13)
    n, n1 = 260, 110
    K = kernel(n, rng)
    perm = rng.permutation(n)
    S = hs.hss.compress(K, hs.hss.bisection_cluster((n1, n), leafsize=32), atol=1e-10, rtol=1e-10, kest=48, perm=perm)
    Kp = K[np.ix_(perm, perm)]
    for which, blk in ((0, Kp[:n1, n1:]), (1, Kp[n1:, :n1])):
        Cm, Z = S.offdiag_lowrank(which)
        assert Cm.shape[1] == Z.shape[0]
        assert relerr(Cm @ Z, blk) < 1e-8
    V = S.view()  # tree order: the permutation is gone
    assert relerr(V.full(), Kp) < 1e-8
    I, J = rng.permutation(n)[:40], rng.permutation(n)[:33]
    assert relerr(V.getindex(I, J), Kp[np.ix_(I, J)]) < 1e-8
