"""Compressed fronts, phase 1 (low-rank Gauss transforms; D and S dense) on the GPU.

Reference scenario: `factor(A, nd, nd_loc; swlevel=-4, swsize=8, atol=rtol=1e-6, ...)` used as right
preconditioner of GMRES (test/rungmres.jl:23-48).  Parity unpinned (LowRankApprox / HssMatrices are not in
the reference tree): the product is compared with `oracle/hs_oracle_lr.py` through quantities both must
agree on to O(tol) -- solution error, GMRES iteration count -- and with its own dense Gauss transforms.
"""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from helpers import prepare, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hs():
    import hsamd

    return hsamd.load()


CASES = [
    (("poisson2d_p1_h64_nmax100", {}), -2, 8),
    (("poisson2d_p1_h128_nmax100", {}), -3, 8),
    (((16, 16, 16), dict(kind="poisson", nmax=64)), -2, 8),
    (("helmholtz2d_p1_h64_nmax100", {}), -2, 8),
    (((16, 16, 16), dict(kind="helmholtz", nmax=64)), -2, 8),
]


@pytest.mark.parametrize("name,swlevel,swsize", CASES)
@pytest.mark.parametrize("tol", [1e-2, 1e-6, 1e-10])
def test_compressed_ldiv_accuracy(hs, name, swlevel, swsize, tol):
    from oracle import hs_oracle as O, hs_oracle_lr as OL

    P = prepare(hs, name[0], rhs="randn", **name[1])
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=swlevel, swsize=swsize, atol=tol, rtol=tol)
    x = hs.ldiv(F, P["b"])
    xr = spla.splu(P["A"]).solve(P["b"])
    Fo = OL.factor(P["A"], P["ond"], P["ond_loc"], swlevel=swlevel, swsize=swsize, atol=tol, rtol=tol)
    xo = O.ldiv(Fo, P["b"])
    e_gpu, e_orc = relerr(x, xr), relerr(xo, xr)
    assert hs.maxrank(F) > 0  # something was compressed (factornode.jl:49-57)
    assert OL.maxrank(Fo) > 0
    # both are O(tol)-accurate preconditioners (the oracle truncates Abi/Aib by QRCP before D^-1 is applied, the
    # product truncates Abi*U^-1 / L^-1*P*Aib by a sketched LU): within 10x of each other or of the tolerance
    assert e_gpu <= max(4 * e_orc, 50 * tol), (e_gpu, e_orc)
    print(f"tol={tol:g} err(product)={e_gpu:.2e} err(oracle)={e_orc:.2e} maxrank {hs.maxrank(F)} / {OL.maxrank(Fo)}")
    # ranks: the device orders the rows by a tournament-pivoted LU of the sketch and truncates on the residual norms of their
    # orthogonalisation, the oracle on the |R_jj| of a column-pivoted QR: within 15 %
    assert hs.maxrank(F) <= 1.15 * OL.maxrank(Fo) + 4


@pytest.mark.parametrize("name,swlevel,swsize", CASES[:3])
def test_compressed_transforms_vs_dense(hs, name, swlevel, swsize):
    """Reconstructed C*Z of every compressed front is within tol of the dense Gauss transform."""
    tol = 1e-6
    P = prepare(hs, name[0], **name[1])
    Fd = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    Fc = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=swlevel, swsize=swsize, atol=tol, rtol=tol)
    ncomp = 0
    for node in range(Fc.nnodes):
        comp, rl, rr = Fc.node_ranks(node)
        if not comp:
            assert (rl, rr) == (0, 0)
            continue
        ncomp += 1
        bd, bc = Fd.node_blocks(node), Fc.node_blocks(node)
        for key, r in (("Lbi", rl), ("Uib", rr)):
            D, Cc = bd[key], bc[key]
            scale = max(np.abs(D).max(), 1e-300)
            assert np.abs(D - Cc).max() <= 200 * tol * max(scale, 1.0), (node, key, r)
            assert r <= min(D.shape)
    assert ncomp > 0


def test_compressed_gmres_iterations(hs):
    """rungmres.jl scenario: GMRES right-preconditioned by a loose-tolerance compressed factorization."""
    from oracle import hs_oracle as O, hs_oracle_lr as OL

    P = prepare(hs, (16, 16, 16), kind="poisson", nmax=64, rhs="randn")
    tol = 1e-2
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=-3, swsize=8, atol=tol, rtol=tol)
    x, ch = hs.gmres(P["A"], P["b"], Pr=F, reltol=1e-9, restart=30, maxiter=30, log=True)
    assert ch["isconverged"]
    assert relerr(P["A"] @ x, P["b"]) < 1e-8
    Fo = OL.factor(P["A"], P["ond"], P["ond_loc"], swlevel=-3, swsize=8, atol=tol, rtol=tol)
    cnt = [0]

    def prec(v):
        cnt[0] += 1
        return O.ldiv(Fo, v)

    M = spla.LinearOperator(P["A"].shape, matvec=prec, dtype=P["A"].dtype)
    xo, info = spla.gmres(P["A"], P["b"], M=M, rtol=1e-9, restart=30, maxiter=30)
    assert info == 0
    assert ch["iters"] <= cnt[0] + 3


def test_compressed_refactor_and_multirhs(hs):
    P = prepare(hs, "poisson2d_p1_h64_nmax100", rhs="randn")
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=-2, swsize=8, atol=1e-8, rtol=1e-8)
    B = np.random.default_rng(0).standard_normal((P["A"].shape[0], 3))
    X = hs.ldiv(F, B)
    Xr = spla.splu(P["A"]).solve(B)
    assert relerr(X, Xr) < 1e-5
    for j in range(3):
        assert relerr(hs.ldiv(F, B[:, j]), X[:, j]) < 1e-12


def test_refactor_reuses_ranks_and_kest(hs):
    """A second numeric factorization of the same handle starts its sketches from the ranks of the first one; `kest`
    seeds the very first sketch (factorization.jl:102-104 uses kest the same way for the Schur sampling)."""
    import torch

    P = prepare(hs, (16, 16, 16), kind="poisson", nmax=64, rhs="randn")
    xr = spla.splu(P["A"]).solve(P["b"])
    S = hs.dist.StagedSolver(P["A"], P["nd"], P["nd_loc"], device=torch.device("cuda:0"), swlevel=-2, swsize=8, atol=1e-8, rtol=1e-8, kest=32)
    errs, ranks = [], []
    for _ in range(3):
        S.numeric()
        b = torch.from_numpy(np.ascontiguousarray(P["b"])).to("cuda:0")
        S.solve(b)
        errs.append(relerr(b.cpu().numpy(), xr))
        ranks.append(int(S.backend.L.hs_maxrank(S.backend._h)))
    assert max(errs) < 1e-5, errs
    assert ranks[0] > 32  # the sketch had to grow beyond kest
    assert max(ranks) - min(ranks) <= 8, ranks  # the randomized rank decision is stable between passes


SPLIT_CASES = [
    ("poisson3d_32", {}, 3, 256),
    ("poisson3d_32", {}, 2, 512),
    ((20, 20, 20), dict(kind="helmholtz", nmax=200), 2, 256),
    ("poisson2d_p1_h128_nmax100", {}, 3, 256),  # fronts too small to split: the option must be a no-op
]


@pytest.mark.parametrize("name,kw,swlevel,split", SPLIT_CASES)
@pytest.mark.parametrize("tol", [1e-6, 1e-13])
def test_split_fronts(hs, name, kw, swlevel, split, tol):
    """hs_options.split: the interior block of a large compressed front is eliminated in slices (block LU of D with
    low-rank off-diagonal panels -- the role of the reference's 2x2 BlockFactorization, blockmatrix.jl:106-130).
    With a tolerance at round-off level the factorization is exact again; otherwise O(tol)."""
    P = prepare(hs, name, rhs="randn", **kw)
    xr = spla.splu(P["A"]).solve(P["b"])
    F0 = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=swlevel, swsize=8, atol=tol, rtol=tol)
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=swlevel, swsize=8, atol=tol, rtol=tol, split_size=split)
    e0, e = relerr(hs.ldiv(F0, P["b"]), xr), relerr(hs.ldiv(F, P["b"]), xr)
    print(f"{name} tol={tol:g} split={split}: err {e:.2e} (unsplit {e0:.2e}) maxrank {hs.maxrank(F)} (unsplit {hs.maxrank(F0)})")
    assert e <= max(30 * e0, 300 * tol, 1e-10), (e, e0)
    # node ids, sizes and levels seen through the API are those of the user's tree
    root = F.nnodes - 1
    assert F.node_info(root) == F0.node_info(root)
    comp, rl, rr = F.node_ranks(root)
    ni, nb, _ = F.node_info(root)
    if ni >= 2 * split:
        assert comp and rl > 0 and rr > 0  # the root itself has no boundary, its slices do
        with pytest.raises(hs.UnsupportedError):
            F.node_blocks(root)
    x, ch = hs.gmres(P["A"], P["b"], Pr=F, reltol=1e-10, restart=30, maxiter=30, log=True)
    assert ch["isconverged"] and ch["iters"] <= (3 if tol < 1e-10 else 8)
