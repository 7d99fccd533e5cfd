"""Generate the committed golden fixtures (DATA ONLY: inputs + expected outputs).

The reference (Julia) cannot run in the build container and ships no vectors of its own
(SURVEY.md section 8(c)), so these fixtures are produced by the CPU oracle (oracle/hs_oracle.py) and
pinned by its own exactness check against SuperLU, asserted below before anything is written.

    python tests/golden/make_golden.py

Each .npz holds: the CSC matrix (indptr/indices/data, 0-based SciPy fields), b, the serialized
elimination tree (7 arrays, 1-based, util/read_problem.jl layout), the solution x = A \\ b, and per
post-order node the reference's FactorNode fields D (raw interior block), L, R and S (S in the
node's OWN bnd order, i.e. before `S[perm,perm]`)."""
import os
import sys

import numpy as np
import scipy.sparse.linalg as spla

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hsamd  # noqa: E402
from oracle import hs_oracle as O  # noqa: E402

CASES = {
    "poisson2d_9x9_nmax12": ((9, 9), "poisson", 12),
    "helmholtz2d_9x9_nmax12": ((9, 9), "helmholtz", 12),
    "poisson3d_6x6x6_nmax30": ((6, 6, 6), "poisson", 30),
    "helmholtz2d_13x7_nmax10": ((13, 7), "helmholtz", 10),
}


def main():
    hs = hsamd.load()
    out = os.path.dirname(os.path.abspath(__file__))
    for name, (shape, kind, nmax) in CASES.items():
        A, b, nd = hs.problems.make_problem(shape, kind=kind, nmax=nmax, rhs="randn", seed=123)
        tree = hs.serialize_elimtree(nd)
        o = O.parse_elimtree(*tree)
        o, o_loc = O.symfact(o)
        perm = O.postorder(o)
        Ap = A[perm - 1][:, perm - 1].tocsc()
        o = O.permuted(o, O.invperm(perm))
        F = O.factor(Ap, o, o_loc, swlevel=0)
        xp = O.ldiv(F, b[perm - 1])
        x = np.empty_like(xp)
        x[perm - 1] = xp
        xs = spla.splu(A.tocsc()).solve(b)
        err = np.linalg.norm(x - xs) / np.linalg.norm(xs)
        assert err < 1e-12, (name, err)
        nodes = []

        def walk(f):
            if f.left is not None:
                walk(f.left)
            if f.right is not None:
                walk(f.right)
            nodes.append(f)

        walk(F)
        d = dict(
            indptr=A.indptr.astype(np.int64), indices=A.indices.astype(np.int64), data=A.data, shape=np.array(A.shape), b=b, x=x,
            fathers=tree[0], lsons=tree[1], rsons=tree[2], ninter=tree[3], inter=tree[4], nbound=tree[5], bound=tree[6],
            nnodes=np.array(len(nodes)),
        )
        for k, f in enumerate(nodes):
            D = f.D
            if isinstance(D, O.BlockFactorization):
                Bm = D.B
                D = np.block([[Bm.A11, Bm.A12], [Bm.A21, Bm.A22 + Bm.A21 @ np.linalg.solve(Bm.A11, Bm.A12)]])
            perm_loc = np.concatenate([f.int_loc, f.bnd_loc]) - 1
            Sp = O._dense(f.S)
            S = np.zeros_like(Sp)
            S[np.ix_(perm_loc, perm_loc)] = Sp
            d[f"D{k}"], d[f"L{k}"], d[f"R{k}"], d[f"S{k}"] = D, O._dense(f.L), O._dense(f.R), S
        np.savez_compressed(os.path.join(out, name + ".npz"), **d)
        print(name, "n =", A.shape[0], "nodes =", len(nodes), "err vs splu = %.1e" % err)


if __name__ == "__main__":
    main()
