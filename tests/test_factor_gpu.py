"""Parity tests proper: hs_factor_* / hs_ldiv_* (HIP, through the C ABI) against the CPU oracle and
against SuperLU on the same seeded inputs.

Tolerances (FP64, stated per SURVEY.md section 8(c)): the dense path is exact, so
  * solution:        ||x - x_ref|| / ||x_ref|| <= 1e-10  (oracle and splu),
  * per-node blocks: relative Frobenius error <= 1e-9 on D, L, R, S of every node vs the oracle.
Pivot orders differ (tournament pivoting here, LAPACK partial pivoting in Julia/the oracle), so
block-wise agreement is to rounding, not bitwise.
"""
import os

import numpy as np
import pytest
import scipy.sparse.linalg as spla

from helpers import prepare, relerr
from oracle import hs_oracle as O

pytestmark = pytest.mark.gpu

SOL_TOL = 1e-10
BLK_TOL = 1e-9


def _solve_check(hs, P, F, nrhs=1, seed=0):
    n = P["A"].shape[0]
    rng = np.random.default_rng(seed)
    B = rng.standard_normal((n, nrhs))
    if F.dtype.kind == "c":
        B = B + 1j * rng.standard_normal((n, nrhs))
    X = hs.ldiv(F, B if nrhs > 1 else B[:, 0])
    Xr = spla.splu(P["A"]).solve(B if nrhs > 1 else B[:, 0])
    assert relerr(X, Xr) < SOL_TOL
    res = np.linalg.norm(P["A"] @ X - (B if nrhs > 1 else B[:, 0])) / np.linalg.norm(B)
    assert res < 1e-11
    return X


@pytest.mark.parametrize("name", ["poisson2d_p1_h64_nmax100", "helmholtz2d_p1_h64_nmax100"])
def test_scenario_vs_oracle_blocks(hs, name):
    """configs[0] (and its complex twin): every node's D, L, R, S and the solution vs the oracle."""
    P = prepare(hs, name)
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0, keep_schur=True)
    Fo = O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)
    onodes = []

    def walk(f):
        if f.left is not None:
            walk(f.left)
        if f.right is not None:
            walk(f.right)
        onodes.append(f)

    walk(Fo)
    assert len(onodes) == F.nnodes
    for k, fo in enumerate(onodes):
        ni, nb, _ = F.node_info(k)
        assert (ni, nb) == (len(fo.int), len(fo.bnd))
        rb = F.reference_blocks(k, with_schur=True)
        D = fo.D.B.dense() if isinstance(fo.D, O.BlockFactorization) else fo.D
        if isinstance(fo.D, O.BlockFactorization):
            # reference stores (A11, A12, A21, S22) unfactored: rebuild Aii = [A11 A12; A21 S22 + A21 A11^-1 A12]
            Bm = fo.D.B
            D = np.block([[Bm.A11, Bm.A12], [Bm.A21, Bm.A22 + Bm.A21 @ np.linalg.solve(Bm.A11, Bm.A12)]])
        assert relerr(rb["D"], D) < BLK_TOL, k
        if nb:
            assert relerr(rb["L"], O._dense(fo.L)) < BLK_TOL, k
            assert relerr(rb["R"], O._dense(fo.R)) < BLK_TOL, k
            # oracle S is S[perm,perm] with perm = [int_loc; bnd_loc]; ours is in the node's own bnd order
            perm = np.concatenate([fo.int_loc, fo.bnd_loc]) - 1
            assert relerr(rb["S"][np.ix_(perm, perm)], O._dense(fo.S)) < BLK_TOL, k
    x = hs.ldiv(F, P["b"])
    xo = O.ldiv(Fo, P["b"])
    assert relerr(x, xo) < SOL_TOL
    assert relerr(x, spla.splu(P["A"]).solve(P["b"])) < SOL_TOL
    assert hs.maxrank(F) == O.maxrank(Fo) == 0


@pytest.mark.parametrize(
    "name", ["poisson2d_p1_h128_nmax100", "helmholtz2d_p1_h128_nmax100", "poisson3d_32", "helmholtz3d_32"]
)
def test_exact_solve_vs_splu(hs, name):
    P = prepare(hs, name, rhs="randn")
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    x = hs.ldiv(F, P["b"])
    assert relerr(x, spla.splu(P["A"]).solve(P["b"])) < SOL_TOL
    _solve_check(hs, P, F, nrhs=3, seed=2)
    st = F.stats()
    assert st["flops_factor"] == pytest.approx(O.tree_flops(P["ond"]) * (4 if F.dtype.kind == "c" else 1), rel=1e-12)


@pytest.mark.parametrize("shape,nmax,kind", [((7, 5), 6, "poisson"), ((9, 9), 12, "helmholtz"), ((6, 6, 6), 30, "poisson"), ((3, 3), 100, "poisson"), ((40, 3), 9, "poisson")])
def test_small_and_ragged_trees(hs, shape, nmax, kind):
    """tiny / ragged trees: single-leaf tree, leaves at different depths, fronts smaller than one panel."""
    P = prepare(hs, shape, kind=kind, nmax=nmax, rhs="randn")
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    Fo = O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0)
    assert relerr(hs.ldiv(F, P["b"]), O.ldiv(Fo, P["b"])) < SOL_TOL


def test_unpermuted_tree_general_index_sets(hs):
    """`factor` does not need the postorder permutation (the script applies it, test/rungmres.jl:17-19):
    scattered, non-contiguous int sets must give the same solution."""
    A, b, nd = hs.problems.make_problem((17, 13), kind="poisson", nmax=20, rhs="randn")
    nd, nd_loc = hs.symfact(nd)
    F = hs.factor(A, nd, nd_loc, swlevel=0)
    assert relerr(hs.ldiv(F, b), spla.splu(A.tocsc()).solve(b)) < SOL_TOL


def test_root_with_boundary(hs):
    """A tree whose root keeps a boundary: ldiv! then solves with the root Schur complement
    (`C[F.bnd,:] = F.S \\ C[F.bnd,:]`, factornode.jl:72)."""
    A, b, nd = hs.problems.make_problem((12, 10), kind="poisson", nmax=16, rhs="randn")
    sub = nd.left  # left subtree: its bnd is non-empty; restrict A to the subtree's DOFs
    dofs = np.sort(np.concatenate([x.int for x in hs.postorder_nodes(sub)] + [sub.bnd]))
    remap = np.zeros(A.shape[0] + 1, dtype=np.int64)
    remap[dofs] = np.arange(1, len(dofs) + 1)
    for x in hs.postorder_nodes(sub):
        x.int, x.bnd = remap[x.int], remap[x.bnd]
    As = A[dofs - 1][:, dofs - 1].tocsc()
    bs = b[dofs - 1]
    arrays = hs.serialize_elimtree(sub)
    sub, sub_loc = hs.symfact(sub)
    assert len(sub.bnd) > 0
    F = hs.factor(As, sub, sub_loc, swlevel=0)
    x = hs.ldiv(F, bs)
    assert relerr(x, spla.splu(As).solve(bs)) < SOL_TOL
    o = O.parse_elimtree(*arrays)
    o, o_loc = O.symfact(o)
    assert relerr(x, O.ldiv(O.factor(As, o, o_loc, swlevel=0), bs)) < SOL_TOL


def test_ldiv_semantics(hs):
    P = prepare(hs, (20, 20), kind="helmholtz", nmax=30, rhs="randn")
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    b = P["b"].copy()
    x2 = hs.ldiv(F, b)  # 2-arg form: returns a new array, b untouched (factornode.jl:62)
    assert np.array_equal(b, P["b"]) and x2 is not b
    c = np.empty_like(b)
    x3 = hs.ldiv(c, F, b)
    assert x3 is c and np.array_equal(c, x2)
    hs.ldiv(b, F, b)  # aliasing: true in-place
    assert np.array_equal(b, x2)
    # linearity (size-independent property)
    rng = np.random.default_rng(0)
    u = rng.standard_normal(len(b)) + 1j * rng.standard_normal(len(b))
    v = rng.standard_normal(len(b)) + 1j * rng.standard_normal(len(b))
    lhs = hs.ldiv(F, 2.0 * u - 3.0j * v)
    assert relerr(lhs, 2.0 * hs.ldiv(F, u) - 3.0j * hs.ldiv(F, v)) < 1e-12
    with pytest.raises(hs.DimensionMismatch):
        hs.ldiv(F, b[:-1])
    assert repr(F) == "FactorNode{ComplexF64}" and F.eltype is np.complex128


def test_error_paths_on_device(hs):
    P = prepare(hs, (9, 9), kind="poisson", nmax=12)
    with pytest.raises(ValueError, match="swsize"):
        hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0, swsize=0)
    with pytest.raises(ValueError, match="c_tol"):
        hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0, c_tol=0.0)
    # singular interior block -> SingularException with the node id
    A = P["A"].tolil(copy=True)
    leaf = hs.postorder_nodes(P["nd"])[0]
    i = int(leaf.int[0]) - 1
    A[:, i] = 0.0
    A[i, :] = 0.0
    with pytest.raises(hs.SingularException):
        hs.factor(A.tocsc(), P["nd"], P["nd_loc"], swlevel=0)
    # one-child node -> ErrorException (factorization.jl:25)
    nd = P["nd"]
    saved = nd.right
    nd.right = None
    loc_saved = P["nd_loc"].right
    P["nd_loc"].right = None
    with pytest.raises(RuntimeError, match="binary tree"):
        hs.factor(P["A"], nd, P["nd_loc"], swlevel=0)
    nd.right, P["nd_loc"].right = saved, loc_saved


def test_golden_fixtures_on_device(hs):
    import glob
    import os

    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert files
    from test_oracle import load_fixture

    for f in files:
        fx = load_fixture(f)
        nd = hs.parse_elimtree(*fx["tree"])
        nd, nd_loc = hs.symfact(nd)
        perm = hs.postorder(nd)
        Ap = fx["A"][perm - 1][:, perm - 1].tocsc()
        nd = hs.permuted(nd, hs.invperm(perm))
        F = hs.factor(Ap, nd, nd_loc, swlevel=0, keep_schur=True)
        x = hs.ldiv(F, fx["b"][perm - 1])
        assert relerr(x, fx["x"][perm - 1]) < SOL_TOL, f
        for k in range(F.nnodes):
            rb = F.reference_blocks(k, with_schur=True)
            assert relerr(rb["D"], fx["D"][k]) < BLK_TOL, (f, k)
            if fx["L"][k].size:
                assert relerr(rb["L"], fx["L"][k]) < BLK_TOL, (f, k)
                assert relerr(rb["R"], fx["R"][k]) < BLK_TOL, (f, k)
                assert relerr(rb["S"], fx["S"][k]) < BLK_TOL, (f, k)


def test_optimistic_pivoting_detection(hs, monkeypatch):
    """Optimistic pivoting at the kernel level (test hook): on a diagonally dominant front the growth flag stays down
    and the factors are exact; on a random front (partial pivoting leaves the diagonal block all the time) it goes up."""
    import ctypes as C

    from hierarchicalsolvers_jl_amd import _lib

    monkeypatch.setenv("HS_HOOK_OPTIMISTIC", "1")
    L = _lib.lib()
    rng = np.random.default_rng(3)
    ni, nb = 200, 70
    m = ni + nb
    for dominant in (True, False, "nan", "inf"):
        F = np.asfortranarray(rng.standard_normal((m, m)))
        if dominant:
            F[np.arange(m), np.arange(m)] += 4.0 * np.sqrt(m)
        if dominant == "nan":  # a NaN multiplier must raise the flag too (fmax() would have dropped it)
            F[40, 3] = np.nan
        if dominant == "inf":  # inf - inf inside the multiplier's dot product
            F[40, 3] = np.inf
            F[40, 4] = -np.inf
        outLF = np.zeros((m, ni), order="F")
        outUR = np.zeros((ni, nb), order="F")
        outSB = np.zeros((nb, nb), order="F")
        rp = np.zeros(ni, dtype=np.int64)
        info = np.zeros(1, dtype=np.int64)
        ms = C.c_double(0)
        p = lambda a: a.ctypes.data_as(_lib.p_f64)  # noqa: E731
        _lib.check(L.hsk_front_factor_d(1, ni, nb, p(F), p(outLF), p(outUR), p(outSB), rp.ctypes.data_as(_lib.p_i64), info.ctypes.data_as(_lib.p_i64), C.byref(ms)))
        if dominant is True:
            assert info[0] == 0
            S = F[ni:, ni:] - F[ni:, :ni] @ np.linalg.solve(F[:ni, :ni], F[:ni, ni:])
            assert relerr(outSB, S) < 1e-11
        else:
            assert info[0] == -1  # a multiplier beyond the growth bound: the caller must redo with the tournament


def test_optimistic_pivoting_redo_path():
    """The level-redo machinery of hs_numeric_levels (assemble again, eliminate with tournament pivoting), forced through
    HS_OPTIMISTIC_FORCE_REDO in a child process: exact solution, the handle stops trying afterwards."""
    import subprocess
    import sys

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, scipy.sparse.linalg as spla, hsamd\n"
        "hs = hsamd.load()\n"
        "from helpers import prepare, relerr\n"
        "P = prepare(hs, (24, 24), kind='poisson', nmax=40, rhs='randn')\n"
        "F = hs.factor(P['A'], P['nd'], P['nd_loc'], swlevel=0, verbose=True)\n"
        "print('ERR', relerr(hs.ldiv(F, P['b']), spla.splu(P['A']).solve(P['b'])))\n"
    ) % (ROOT, os.path.join(ROOT, "tests"))
    env = dict(os.environ, HS_OPTIMISTIC_FORCE_REDO="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("redoing the level with tournament pivoting") == 1  # the first level only: then the handle gives up
    err = float([ln for ln in r.stdout.splitlines() if ln.startswith("ERR")][0].split()[1])
    assert err < 1e-10


def test_arena_cache_reuse_and_trim(hs):
    """hs_free parks the big device blocks of a factorization and the next factorization of the same size takes them over (the driver needs
    4.9 s for the 139 GiB arena of Poisson 128^3 once such a block has been freed before, 1.4 s in a fresh process: the unexplained 5 s of the
    one-shot `factor` in round 2); hs_trim gives them back.  A recycled arena is NOT zero-filled: the result must not depend on that."""
    P = prepare(hs, (48, 48, 48), rhs="randn", kind="poisson", nmax=512)
    xr = None
    hs.trim()
    for rep in range(3):
        F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
        x = hs.ldiv(F, P["b"])
        if xr is None:
            xr = x
            assert np.linalg.norm(P["A"] @ x - P["b"]) <= 1e-10 * np.linalg.norm(P["b"])
        else:
            assert np.array_equal(x, xr)  # same kernels on the same data, wherever the arena came from
        fac_bytes = F.stats()["bytes_factors"]
        F.free()
    assert fac_bytes > 300 * 2**20  # large enough to be parked
    released = hs.trim()
    assert released >= 0.9 * fac_bytes, (released, fac_bytes)
    assert hs.trim() == 0
    # a factorization of ANOTHER size after the parked blocks: allocates its own
    P2 = prepare(hs, (32, 32, 32), rhs="randn", kind="helmholtz", nmax=512)
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    F.free()
    F2 = hs.factor(P2["A"], P2["nd"], P2["nd_loc"], swlevel=0)
    x2 = hs.ldiv(F2, P2["b"])
    assert np.linalg.norm(P2["A"] @ x2 - P2["b"]) <= 1e-10 * np.linalg.norm(P2["b"])
    F2.free()
    hs.trim()


@pytest.mark.parametrize("kind,shape,nmax", [("poisson", (40, 36, 33), 1400), ("helmholtz", (37, 30, 26), 900)])
def test_dataflow_sweeps_agree_with_the_launch_per_step_sweeps(hs, kind, shape, nmax, tmp_path):
    """ldiv! runs its triangular sweeps as ONE dataflow launch per level (kernels_solve_wide.hip: workgroups own 64 rows, values are exchanged
    through sentinel-armed vectors); HS_SOLVE_FLOW=0 (read once per process: a child process) selects the launch-per-256-columns sweeps.  Both
    must give the solution of SuperLU, ragged fronts (ni not a multiple of 64 or 256), several right-hand sides and repeated solves included."""
    import subprocess
    import sys

    P = prepare(hs, shape, kind=kind, nmax=nmax, rhs="randn")
    F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    rng = np.random.default_rng(2)
    B = rng.standard_normal((len(P["b"]), 3)) + (1j * rng.standard_normal((len(P["b"]), 3)) if kind == "helmholtz" else 0)
    ref = spla.splu(P["A"]).solve(B)
    for rep in range(3):  # the exchange vectors are re-armed by every sweep
        X = hs.ldiv(F, B)
        assert relerr(X, ref) < SOL_TOL
    np.save(tmp_path / "B.npy", B)
    code = f"""
import sys, numpy as np
sys.path.insert(0, {str(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))!r}); sys.path.insert(0, {str(os.path.dirname(os.path.abspath(__file__)))!r})
import hsamd
from helpers import prepare
hs = hsamd.load()
P = prepare(hs, {shape!r}, kind={kind!r}, nmax={nmax}, rhs="randn")
F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
np.save({str(tmp_path / "X0.npy")!r}, hs.ldiv(F, np.load({str(tmp_path / "B.npy")!r})))
"""
    env = dict(os.environ, HS_SOLVE_FLOW="0")
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
    X0 = np.load(tmp_path / "X0.npy")
    assert relerr(X0, ref) < SOL_TOL and relerr(X, X0) < 1e-11


def test_a_plan_that_cannot_fit_is_refused_before_any_allocation(hs):
    """Poisson 160^3 exact needs ~340 GiB of dense factors: hs_analyze says so (HS_ERR_NOMEM -> MemoryError, with the bytes and where the
    per-rank sizes of every flow can be looked up) instead of failing somewhere inside hipMalloc."""
    import time

    P = prepare(hs, (160, 160, 160), kind="poisson", nmax=4096)
    t0 = time.perf_counter()
    with pytest.raises(MemoryError) as e:
        hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
    msg = str(e.value)
    assert "GiB" in msg and "size_model" in msg, msg
    assert time.perf_counter() - t0 < 60.0
    # the library is still usable afterwards
    P2 = prepare(hs, (12, 11, 10), kind="poisson", nmax=60)
    F = hs.factor(P2["A"], P2["nd"], P2["nd_loc"], swlevel=0)
    assert relerr(hs.ldiv(F, P2["b"]), spla.splu(P2["A"]).solve(P2["b"])) < SOL_TOL
