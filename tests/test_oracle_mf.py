"""CPU tests of the matrix-free oracle (oracle/hs_oracle_mf.py): Schur complements handed between fronts as HSS matrices,
`D = blockfactor` over HSS blocks, Gauss transforms from the children's generators, `S` sampled from its operator
(rows C3, B2', B5, C5, C6, F2 of SURVEY.md section 8).  PARITY UNPINNED; pinned by SuperLU and by the dense-S oracles."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from helpers import prepare, relerr
from oracle import hs_hss as HS, hs_oracle_lr as OL, hs_oracle_mf as OM


@pytest.mark.parametrize("name,swlevel", [(((16, 16, 16), dict(kind="poisson", nmax=64)), 3), (((12, 12, 12), dict(kind="helmholtz", nmax=64)), 2),
                                          (("poisson2d_p1_h64_nmax100", {}), 3)])
def test_matrix_free_compressed_branch(hs, name, swlevel):
    P = prepare(hs, name[0], rhs="randn", **name[1])
    xr = spla.splu(P["A"]).solve(P["b"])
    errs = {}
    for tol in (1e-3, 1e-8):
        F = OM.factor(P["A"], P["ond"], P["ond_loc"], dexp=2, swlevel=swlevel, swsize=8, atol=tol, rtol=tol, leafsize=16)
        kinds = OM.count_kinds(F)
        assert kinds.get("mf", 0) >= 1, kinds  # at least the root assembled HSS children without densifying them
        assert F.kind == "mf" and isinstance(F.left.S, HS.Hss) and isinstance(F.right.S, HS.Hss)
        errs[tol] = relerr(OM.ldiv(F, P["b"]), xr)
        assert OM.maxrank(F) > 0
    print(name[0], errs)
    assert errs[1e-8] < 1e-5 and errs[1e-3] < 0.2 and errs[1e-8] < errs[1e-3]
    # the same preconditioner quality as the dense-S variant (hs_oracle_lr) at the loose tolerance
    Fl = OL.factor(P["A"], P["ond"], P["ond_loc"], swlevel=swlevel, swsize=8, atol=1e-3, rtol=1e-3)
    from oracle import hs_oracle as O

    e_lr = relerr(O.ldiv(Fl, P["b"]), xr)
    assert errs[1e-3] <= max(300 * e_lr, 0.2)


def test_exact_when_nothing_is_compressed(hs):
    P = prepare(hs, (10, 9, 8), rhs="randn", kind="poisson", nmax=40)
    F = OM.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)
    assert OM.count_kinds(F) == {"dense": OM.count_kinds(F)["dense"]}
    assert relerr(OM.ldiv(F, P["b"]), spla.splu(P["A"]).solve(P["b"])) < 1e-10


def test_single_hss_interior_block_matrix_free(hs):
    """`D` as ONE HSS matrix over a bisection order, compressed from the children's HSS blocks + sparse couplings without
    `A11^-1 * A12` (the device formulation of hs_options.hss_d, matrix-free): same accuracy as the reference's 2x2 blockfactor."""
    P = prepare(hs, (16, 16, 16), rhs="randn", kind="poisson", nmax=64)
    xr = spla.splu(P["A"]).solve(P["b"])
    for tol, bound in ((1e-3, 0.2), (1e-8, 1e-5)):
        kw = dict(dexp=2, swlevel=3, swsize=8, atol=tol, rtol=tol, leafsize=16)
        Fs = OM.factor(P["A"], P["ond"], P["ond_loc"], dmode="single", **kw)
        Fb = OM.factor(P["A"], P["ond"], P["ond_loc"], dmode="block", **kw)
        assert isinstance(Fs.D, OM.SingleD) and isinstance(Fb.D, OM.BlockD)
        es, eb = relerr(OM.ldiv(Fs, P["b"]), xr), relerr(OM.ldiv(Fb, P["b"]), xr)
        print(f"tol={tol:g}: single HSS D {es:.2e} (hssrank {Fs.D.hssrank}), 2x2 blockfactor {eb:.2e} (hssrank {Fb.D.hssrank})")
        assert es < bound and es <= max(30 * eb, bound)
