"""Shared helpers for the parity tests: run the reference scenario (test/rungmres.jl:15-19) on both sides."""
import numpy as np

from oracle import hs_oracle as O


def prepare(hs, name_or_shape, **kw):
    """read -> symfact! -> postorder -> permute -> permuted!  (test/rungmres.jl:15-19), for the
    product's host layer AND, independently, for the oracle."""
    A, b, nd = hs.problems.make_problem(name_or_shape, **kw)
    arrays = hs.serialize_elimtree(nd)
    # product side
    nd, nd_loc = hs.symfact(nd)
    perm = hs.postorder(nd)
    Ap = A[perm - 1][:, perm - 1].tocsc()
    nd = hs.permuted(nd, hs.invperm(perm))
    bp = b[perm - 1]
    # oracle side (its own restatement of the symbolic layer)
    ond = O.parse_elimtree(*arrays)
    ond, ond_loc = O.symfact(ond)
    operm = O.postorder(ond)
    assert np.array_equal(operm, perm)
    ond = O.permuted(ond, O.invperm(operm))
    return dict(A=Ap, b=bp, nd=nd, nd_loc=nd_loc, ond=ond, ond_loc=ond_loc, perm=perm, A0=A, b0=b)


def relerr(x, y):
    d = np.linalg.norm(np.asarray(x) - np.asarray(y))
    n = np.linalg.norm(np.asarray(y))
    return d / n if n > 0 else d
