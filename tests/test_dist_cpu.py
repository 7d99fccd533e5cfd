"""N > 1 path on the CPU: the multi-rank schedule of hierarchicalsolvers.jl_amd/dist.py (`run_numeric`,
`run_solve` -- the SAME code the GPU path runs) driven over torch.distributed/gloo with an
oracle-backed backend.  It checks that the subtree partition, the Schur-complement exchanges at the
joins and the boundary-vector exchanges of ldiv! reproduce the serial oracle solution.

The partition itself (who owns which front, what crosses ranks) is checked against an independent
restatement of the ownership rule.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _few_blas_threads_per_rank(monkeypatch):
    """2-8 rank processes share the 8 cores of the CI container: one or two BLAS threads each (spawned children inherit the environment;
    without this every rank starts 8 OpenBLAS / OpenMP threads and the world-8 cases take minutes)."""
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        monkeypatch.setenv(k, "1")


def build_plan(hs, nd, nranks):
    """Independent restatement of the ownership rule: cut at level p+1, 2^p = nranks;
    a front above the cut belongs to the first rank of its group."""
    from hierarchicalsolvers_jl_amd.dist import Plan

    nodes = hs.postorder_nodes(nd)
    ids = {id(x): k for k, x in enumerate(nodes)}
    owner, level = {}, {}

    def walk(x, lv, lo, cnt):
        owner[ids[id(x)]] = lo
        level[ids[id(x)]] = lv
        if x.left is not None:
            if cnt > 1:
                walk(x.left, lv + 1, lo, cnt // 2)
                walk(x.right, lv + 1, lo + cnt // 2, cnt // 2)
            else:
                walk(x.left, lv + 1, lo, 1)
                walk(x.right, lv + 1, lo, 1)

    walk(nd, 1, 0, nranks)
    ex = []
    for k, x in enumerate(nodes):
        for c in (x.left, x.right):
            if c is not None and owner[ids[id(c)]] != owner[k]:
                ex.append(dict(node=ids[id(c)], level=level[ids[id(c)]], src=owner[ids[id(c)]], dst=owner[k], nb=len(c.bnd), nelems=len(c.bnd) ** 2))
    p = int(np.log2(nranks))
    return Plan(max(level.values()), p + 1, ex, nranks), owner, level


class OracleBackend:
    """The backend interface of dist.py on top of the NumPy oracle: one front at a time, fronts owned by
    other ranks are never touched.  Exchange buffers are CPU torch tensors."""

    def __init__(self, A, ond, ond_loc, owner, level, rank):
        import torch

        from oracle import hs_oracle as O

        self.O, self.torch = O, torch
        self.A, self.rank = A.tocsc(), rank
        self.nodes, self.locs = [], []

        def walk(x, xl):
            if x.left is not None:
                walk(x.left, xl.left)
                walk(x.right, xl.right)
            self.nodes.append(x)
            self.locs.append(xl)

        walk(ond, ond_loc)
        self.ids = {id(x): k for k, x in enumerate(self.nodes)}
        self.owner, self.level = owner, level
        self.F = {}  # node id -> FactorNode (owned) or a stub holding a received S
        self.is_c = np.iscomplexobj(A.data)
        self.dtype = np.complex128 if self.is_c else np.float64
        self.tdtype = torch.complex128 if self.is_c else torch.float64
        self._schur = {}

    # -- numeric ------------------------------------------------------------------------------------------
    def numeric_begin(self):
        self.F = {}

    def numeric_levels(self, lv_from, lv_to):
        O = self.O
        opts = O.SolverOptions(swlevel=0)
        for lv in range(lv_from, lv_to - 1, -1):
            for k, x in enumerate(self.nodes):
                if self.level[k] != lv or self.owner[k] != self.rank:
                    continue
                if x.left is None:
                    self.F[k] = O._factor_leaf(self.A, x, self.locs[k], False, opts)
                else:
                    Fl, Fr = self.F[self.ids[id(x.left)]], self.F[self.ids[id(x.right)]]
                    self.F[k] = O._factor_branch(self.A, Fl, Fr, x, self.locs[k], False, opts)

    def numeric_end(self):
        pass

    def schur_tensor(self, node):
        """Owned child: its S (already `S[perm,perm]`, what the parent consumes); remote child: a receive buffer."""
        nb = len(self.nodes[node].bnd)
        if node not in self._schur:
            self._schur[node] = self.torch.zeros(nb * nb, dtype=self.tdtype)
        t = self._schur[node]
        if self.owner[node] == self.rank:
            t.copy_(self.torch.from_numpy(np.ascontiguousarray(self.F[node].S).reshape(-1)))
        else:  # stub whose .S is a view of the buffer (filled by recv before the parent is factored)
            stub = type("Stub", (), {})()
            stub.S = t.numpy().reshape(nb, nb)
            self.F[node] = stub
        return t

    def sync(self):
        pass

    def comm_sync(self):
        pass

    def host_sync(self):
        pass

    # -- solve -----------------------------------------------------------------------------------------------
    def _mine(self, lv_lo, lv_hi):
        return [k for k in range(len(self.nodes)) if lv_lo <= self.level[k] <= lv_hi and self.owner[k] == self.rank]

    def fwd(self, b, lv_from, lv_to):  # post-order over owned fronts, deepest level first (factornode.jl:77-82)
        O = self.O
        for lv in range(lv_from, max(lv_to, 1) - 1, -1):
            for k in self._mine(lv, lv):
                F = self.F[k]
                b[F.bnd - 1] = b[F.bnd - 1] - O._dense(F.L) @ b[F.int - 1]

    def bwd(self, b, lv_from, lv_to):  # _dsolve! + _rsolve! fused per front, root first (factornode.jl:83-99)
        O = self.O
        for lv in range(max(lv_from, 1), lv_to + 1):
            for k in self._mine(lv, lv):
                F = self.F[k]
                y = O.blockldiv_inplace(F.D, b[F.int - 1][:, None])[:, 0] if isinstance(F.D, O.BlockFactorization) else O._ldiv(F.D, b[F.int - 1])
                b[F.int - 1] = y - O._dense(F.R) @ b[F.bnd - 1]

    def pack_bnd(self, node, b):
        return self.torch.from_numpy(np.ascontiguousarray(b[self.nodes[node].bnd - 1]))

    def bnd_buffer(self, node):
        return self.torch.zeros(len(self.nodes[node].bnd), dtype=self.tdtype)

    def unpack_bnd(self, node, b, buf):
        b[self.nodes[node].bnd - 1] = buf.numpy()

    def extract_owned(self, b):
        out = np.zeros_like(b)
        for k in self._mine(1, 10**9):
            out[self.nodes[k].int - 1] = b[self.nodes[k].int - 1]
        return self.torch.from_numpy(out)

    def assign(self, b, out):
        b[:] = out.numpy()


def _worker(rank, world, port, name, q):
    try:
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hsamd

        hs = hsamd.load()
        from helpers import prepare
        from hierarchicalsolvers_jl_amd.dist import TorchComm, run_numeric, run_solve
        from oracle import hs_oracle as O

        shape, kind, nmax = name
        P = prepare(hs, shape, kind=kind, nmax=nmax, rhs="randn")
        plan, owner, level = build_plan(hs, P["nd"], world)
        be = OracleBackend(P["A"], P["ond"], P["ond_loc"], owner, level, rank)
        comm = TorchComm()
        run_numeric(be, plan, rank, comm)
        b = np.array(P["b"], dtype=be.dtype)
        run_solve(be, plan, rank, comm, b)
        xs = O.ldiv(O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0), P["b"])  # serial oracle
        err = float(np.linalg.norm(b - xs) / np.linalg.norm(xs))
        nsent = sum(1 for e in plan.exchanges if e["src"] == rank)
        q.put((rank, err, nsent, len(plan.exchanges)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), 0, 0))


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("name", [((17, 13), "poisson", 12), ((8, 8, 6), "helmholtz", 30)])
def test_subtree_partition_over_gloo(world, name):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 13 + len(name[0])) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, nsent, nex in sorted(res):
        assert not isinstance(err, str), err
        assert err < 1e-10, (rank, err)
        assert nex == world - 1  # one Schur complement crosses ranks per join above the cut
    assert sum(r[2] for r in res) == world - 1


def test_plan_matches_library(hs):
    """The C library's ownership / exchange list (hs_plan, host only -- no GPU) equals the independent restatement."""
    import ctypes as C

    from helpers import prepare

    P = prepare(hs, (20, 12), kind="poisson", nmax=10)
    L = hs._lib.lib()
    for world in (1, 2, 4, 8):
        plan, owner, level = build_plan(hs, P["nd"], world)
        h = hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=world)
        try:
            assert L.hs_nlevels(h) == plan.nlevels and L.hs_cut_level(h) == plan.cut_level
            for k, o in owner.items():
                assert L.hs_node_owner(h, k) == o
            out6 = (C.c_int64 * 6)()
            got = []
            for k in range(L.hs_num_exchanges(h)):
                hs._lib.check(L.hs_exchange_info(h, k, out6))
                got.append((out6[0], out6[1], out6[2], out6[3], out6[4]))
            want = [(e["node"], e["level"], e["src"], e["dst"], e["nb"]) for e in plan.exchanges]
            assert sorted(got) == sorted(want)
        finally:
            L.hs_free(h)
    with pytest.raises(ValueError, match="power of two"):
        hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=3)


# ---- hs_options.dist_top: fronts above the cut belong to their whole group (csrc/hs_dist.h) -----------------------------------------
def build_plan_dist(hs, nd, nranks):
    """Independent restatement of the group rule: a front above the cut is held by every rank of its group; at a join every rank of a
    child's group swaps with its partner in the sibling's group (rank +- |child group|)."""
    from hierarchicalsolvers_jl_amd.dist import Plan

    nodes = hs.postorder_nodes(nd)
    ids = {id(x): k for k, x in enumerate(nodes)}
    group, level = {}, {}

    def walk(x, lv, lo, cnt):
        group[ids[id(x)]] = (lo, cnt)
        level[ids[id(x)]] = lv
        if x.left is not None:
            half = cnt // 2 if cnt > 1 else 1
            walk(x.left, lv + 1, lo, half)
            walk(x.right, lv + 1, lo + (cnt // 2 if cnt > 1 else 0), half)

    walk(nd, 1, 0, nranks)
    ex = []
    for k, x in enumerate(nodes):
        if x.left is None or group[k][1] == 1:
            continue
        for c, sign in ((x.left, +1), (x.right, -1)):
            ck = ids[id(c)]
            lo, cnt = group[ck]
            for r in range(lo, lo + cnt):
                ex.append(dict(node=ck, level=level[ck], src=r, dst=r + sign * cnt, nb=len(c.bnd), nelems=len(c.bnd) ** 2))
    p = int(np.log2(nranks))
    return Plan(max(level.values()), p + 1, ex, nranks), group, level


class OracleBackendDist(OracleBackend):
    """Fronts above the cut are held (and, on the CPU, simply eliminated again) by every member of their group; what is checked is the
    exchange pattern of the joins and the replicated sweeps of `run_solve_dist`."""

    def __init__(self, A, ond, ond_loc, group, level, rank, plan, comm):
        owner = {k: (rank if lo <= rank < lo + cnt else lo) for k, (lo, cnt) in group.items()}  # "owner == rank" <=> member of the group
        super().__init__(A, ond, ond_loc, owner, level, rank)
        self.first = {k: lo for k, (lo, cnt) in group.items()}
        self.plan, self.comm = plan, comm

    def numeric_levels(self, lv_from, lv_to):
        cut = self.plan.cut_level
        for lv in range(lv_from, max(lv_to, 1) - 1, -1):
            if lv < cut:  # the join below this level: swap the children's Schur complements with the partner rank
                mine = [e for e in self.plan.at_child_level(lv + 1) if e["src"] == self.rank]
                theirs = [e for e in self.plan.at_child_level(lv + 1) if e["dst"] == self.rank]
                assert len(mine) == 1 and len(theirs) == 1 and mine[0]["dst"] == theirs[0]["src"]
                nb_s, nb_r = mine[0]["nb"], theirs[0]["nb"]
                send = self.torch.from_numpy(np.ascontiguousarray(self.F[mine[0]["node"]].S).reshape(-1).astype(self.dtype))
                recv = self.torch.zeros(nb_r * nb_r, dtype=self.tdtype)
                assert send.numel() == nb_s * nb_s
                self.comm.exchange(send, recv, mine[0]["dst"])
                stub = type("Stub", (), {})()
                stub.S = recv.numpy().reshape(nb_r, nb_r)
                self.F[theirs[0]["node"]] = stub
            super().numeric_levels(lv, lv)

    def extract_owned(self, b):
        out = np.zeros_like(b)
        for k in range(len(self.nodes)):
            if self.first[k] == self.rank and self.owner[k] == self.rank:
                out[self.nodes[k].int - 1] = b[self.nodes[k].int - 1]
        return self.torch.from_numpy(out)


def _worker_dist(rank, world, port, name, q):
    try:
        import torch  # noqa: F401
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hsamd

        hs = hsamd.load()
        from helpers import prepare
        from hierarchicalsolvers_jl_amd.dist import TorchComm, run_numeric_dist, run_solve_dist
        from oracle import hs_oracle as O

        shape, kind, nmax = name
        P = prepare(hs, shape, kind=kind, nmax=nmax, rhs="randn")
        plan, group, level = build_plan_dist(hs, P["nd"], world)
        comm = TorchComm()
        be = OracleBackendDist(P["A"], P["ond"], P["ond_loc"], group, level, rank, plan, comm)
        be.plan = plan
        run_numeric_dist(be)
        b = np.array(P["b"], dtype=be.dtype)
        run_solve_dist(be, plan, rank, comm, b)
        xs = O.ldiv(O.factor(P["A"], P["ond"], P["ond_loc"], swlevel=0), P["b"])  # serial oracle
        err = float(np.linalg.norm(b - xs) / np.linalg.norm(xs))
        q.put((rank, err, len(plan.exchanges)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), 0))


@pytest.mark.parametrize("world,name", [(2, ((17, 13), "poisson", 12)), (4, ((17, 13), "poisson", 12)), (2, ((8, 8, 6), "helmholtz", 30)),
                                        (4, ((8, 8, 6), "helmholtz", 30)), (8, ((33, 29), "poisson", 12)), (8, ((12, 10, 10), "helmholtz", 30))])
def test_group_fronts_over_gloo(world, name):
    """dist_top: replicated fronts above the cut, pairwise swaps at the joins, no communication in the backward sweep."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 7 + world * 29 + len(name[0]) + 101) % 2000
    procs = [ctx.Process(target=_worker_dist, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, nex in sorted(res):
        assert not isinstance(err, str), err
        assert err < 1e-10, (rank, err)
        assert nex == world * int(np.log2(world))  # every rank swaps once per level above the cut


def test_dist_top_plan_matches_library(hs):
    """hs_plan with dist_top: group membership and the symmetric exchange list equal the independent restatement."""
    import ctypes as C

    from helpers import prepare

    P = prepare(hs, (20, 12), kind="poisson", nmax=10)
    L = hs._lib.lib()
    for world in (2, 4, 8):
        plan, group, level = build_plan_dist(hs, P["nd"], world)
        for rank in (0, world - 1):
            h = hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=rank, nranks=world, dist_top=True)
            try:
                assert L.hs_cut_level(h) == plan.cut_level
                for k, (lo, cnt) in group.items():
                    assert L.hs_node_owner(h, k) == lo
                out6 = (C.c_int64 * 6)()
                got = []
                for k in range(L.hs_num_exchanges(h)):
                    hs._lib.check(L.hs_exchange_info(h, k, out6))
                    got.append((out6[0], out6[1], out6[2], out6[3], out6[4]))
                want = [(e["node"], e["level"], e["src"], e["dst"], e["nb"]) for e in plan.exchanges]
                assert sorted(got) == sorted(want)
            finally:
                L.hs_free(h)


def test_dist_top_rejects_a_tree_that_ends_above_the_cut(hs):
    """A branch that ends above the rank cut leaves its group with nothing to join: refused with the plan (dist_top = 0 still accepts it)."""
    from helpers import prepare

    P = prepare(hs, (9, 7), kind="poisson", nmax=40)  # a shallow tree
    L = hs._lib.lib()
    depth = None
    h = hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=1)
    try:
        depth = L.hs_nlevels(h)
    finally:
        L.hs_free(h)
    world = 1 << depth  # more ranks than the shortest branch has levels
    with pytest.raises(hs._lib.Unsupported if hasattr(hs._lib, "Unsupported") else Exception, match="above the rank cut"):
        hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=world, dist_top=True)
    h = hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=world)
    L.hs_free(h)


def _worker_transfer(rank, world, port, q):
    try:
        import ctypes as C

        import torch  # noqa: F401
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hsamd

        hs = hsamd.load()
        from hierarchicalsolvers_jl_amd.dist import LibComm

        # what the library hands to hs_transfer_fn: one host message per peer and direction (here: to / from every other rank, sizes differ per pair)
        peers = [r for r in range(world) if r != rank]
        sbufs = [np.full(1000 + 10 * rank + p, 16 * rank + p, dtype=np.uint8) for p in peers]
        rbufs = [np.zeros(1000 + 10 * p + rank, dtype=np.uint8) for p in peers]
        n = len(peers)
        i64 = C.c_int64
        speer = (i64 * n)(*peers)
        sbytes = (i64 * n)(*[b.size for b in sbufs])
        rbytes = (i64 * n)(*[b.size for b in rbufs])
        sptr = (C.c_void_p * n)(*[b.ctypes.data for b in sbufs])
        rptr = (C.c_void_p * n)(*[b.ctypes.data for b in rbufs])
        st = LibComm._transfer(None, n, speer, sptr, sbytes, n, speer, rptr, rbytes)
        ok = st == 0 and all(np.all(rbufs[k] == 16 * p + rank) for k, p in enumerate(peers))
        st0 = LibComm._transfer(None, 0, speer, sptr, sbytes, 0, speer, rptr, rbytes)  # nothing to move
        q.put((rank, bool(ok and st0 == 0)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_host_transfer_callback_over_gloo(world):
    """The Python half of the host-staged communicator (`LibComm._transfer`, the hs_transfer_fn the library calls with host pointers):
    every rank exchanges one message of its own size with every other rank in a single call."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 5 + world * 31 + 407) % 2000
    procs = [ctx.Process(target=_worker_transfer, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, ok in sorted(res):
        assert ok is True, (rank, ok)


# ---- hs_options.mf over ranks: the joins ship HSS generators (src/factorization.jl:78-112,126-140) ---------------------------
class OracleMfBackend(OracleBackend):
    """The backend interface of dist.py on top of oracle/hs_oracle_mf.py, one front at a time (`factor_node`): a flagged front hands its
    Schur complement on as an HSS matrix; where the parent lives on another rank the generators cross as ONE byte buffer (`hss_pack`), the
    byte count first -- the same schedule (`run_numeric`, `run_solve`) the GPU path runs with hs_schur_pack / hs_schur_unpack."""

    def __init__(self, A, ond, ond_loc, owner, level, rank, opts):
        super().__init__(A, ond, ond_loc, owner, level, rank)
        from oracle import hs_hss as HS
        from oracle import hs_oracle_mf as OM

        self.OM, self.HS, self.opts = OM, HS, opts
        self.swlevel = opts.swlevel
        self.dsc = 1e-2
        self.keep = {}
        self.moved = 0
        OM._factor.G = None  # dmode = "block"

    def numeric_levels(self, lv_from, lv_to):
        for lv in range(lv_from, lv_to - 1, -1):
            for k, x in enumerate(self.nodes):
                if self.level[k] != lv or self.owner[k] != self.rank:
                    continue
                Fl = Fr = None
                if x.left is not None:
                    Fl, Fr = self.F[self.ids[id(x.left)]], self.F[self.ids[id(x.right)]]
                self.F[k] = self.OM.factor_node(self.A, x, self.locs[k], lv, self.swlevel, self.opts, self.dsc, Fl, Fr)

    def schur_hss_pack(self, node):
        S = self.F[node].S
        assert isinstance(S, self.HS.Hss), "the plan says HSS, the front produced a dense S"
        ints, flat = self.HS.hss_pack(S)
        head = np.array([len(ints), flat.nbytes], dtype=np.int64)
        raw = np.concatenate([head.view(np.uint8), ints.view(np.uint8), flat.view(np.uint8)])
        return self.torch.from_numpy(raw.copy())

    def schur_hss_unpack(self, node, buf):
        raw = buf.numpy()
        nint, nbytes = (int(v) for v in raw[:16].view(np.int64))
        ints = raw[16:16 + 8 * nint].view(np.int64)
        is_c = bool(ints[2])
        flat = raw[16 + 8 * nint:16 + 8 * nint + nbytes].view(np.complex128 if is_c else np.float64)
        self.F[node] = self.OM.remote_child(self.HS.hss_unpack(ints, flat), self.nodes[node], self.locs[node])

    def schur_tensor(self, node):  # a join whose child is NOT flagged ships the dense block, as on the exact path
        nb = len(self.nodes[node].bnd)
        if node not in self._schur:
            self._schur[node] = self.torch.zeros(nb * nb, dtype=self.tdtype)
        t = self._schur[node]
        if self.owner[node] == self.rank:
            t.copy_(self.torch.from_numpy(np.ascontiguousarray(self.F[node].S).reshape(-1)))
        else:
            self.F[node] = self.OM.remote_child(t.numpy().reshape(nb, nb), self.nodes[node], self.locs[node])
        return t

    def size_tensor(self, value):
        return self.torch.tensor([int(value)], dtype=self.torch.int64)

    def byte_buffer(self, nbytes):
        return self.torch.empty(int(nbytes), dtype=self.torch.uint8)

    def note_transfer(self, nbytes):
        self.moved += int(nbytes)

    def fwd(self, b, lv_from, lv_to):  # hs_oracle_mf._fwd, one front at a time
        O = self.O
        for lv in range(lv_from, max(lv_to, 1) - 1, -1):
            for k in self._mine(lv, lv):
                F = self.F[k]
                i, bb = F.int - 1, F.bnd - 1
                if F.kind == "mf":
                    t = F.D.solve(b[i][:, None])[:, 0]
                    self.keep[k] = t
                    if len(bb):
                        b[bb] = b[bb] - F.L.U @ (F.L.V.conj().T @ t)
                elif len(bb):
                    b[bb] = b[bb] - O._dense(F.L) @ b[i]

    def bwd(self, b, lv_from, lv_to):
        O = self.O
        for lv in range(max(lv_from, 1), lv_to + 1):
            for k in self._mine(lv, lv):
                F = self.F[k]
                i, bb = F.int - 1, F.bnd - 1
                if F.kind == "mf":
                    b[i] = self.keep[k] - (F.R.U @ (F.R.V.conj().T @ b[bb]) if len(bb) else 0)
                else:
                    d = O.blockldiv_inplace(F.D, b[i][:, None])[:, 0] if isinstance(F.D, O.BlockFactorization) else O._ldiv(F.D, b[i])
                    b[i] = d - (O._dense(F.R) @ b[bb] if len(bb) else 0)


def _worker_mf(rank, world, port, name, q):
    try:
        os.environ.setdefault("OMP_NUM_THREADS", "2")
        os.environ.setdefault("OPENBLAS_NUM_THREADS", "2")
        import torch  # noqa: F401
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import hsamd

        hs = hsamd.load()
        from helpers import prepare
        from hierarchicalsolvers_jl_amd.dist import TorchComm, run_numeric, run_solve
        from oracle import hs_hss as HS
        from oracle import hs_oracle as O
        from oracle import hs_oracle_mf as OM

        shape, kind, nmax, swlevel, leaf, tol = name
        P = prepare(hs, shape, kind=kind, nmax=nmax, rhs="randn")
        plan, owner, level = build_plan(hs, P["nd"], world)
        opts = O.SolverOptions(swlevel=swlevel, swsize=8, atol=tol, rtol=tol, leafsize=leaf)
        be = OracleMfBackend(P["A"], P["ond"], P["ond_loc"], owner, level, rank, opts)
        # which joins ship generators: the child is flagged (factorization.jl:15) and its S is compressed (|int_loc| > 0, |bnd| > leafsize)
        for e in plan.exchanges:
            k = e["node"]
            e["hss"] = level[k] <= swlevel and len(be.nodes[k].bnd) >= opts.swsize and len(be.locs[k].int) > 0 and len(be.nodes[k].bnd) > leaf
        comm = TorchComm()
        run_numeric(be, plan, rank, comm)
        b = np.array(P["b"], dtype=be.dtype)
        run_solve(be, plan, rank, comm, b)
        Fs = OM.factor(P["A"], P["ond"], P["ond_loc"], opts=opts)  # the serial oracle: same fronts, same seeds, one process
        xs = OM.ldiv(Fs, P["b"])
        import scipy.sparse.linalg as spla

        xe = spla.splu(P["A"]).solve(P["b"])
        q.put((rank, float(np.linalg.norm(b - xs) / np.linalg.norm(xs)), float(np.linalg.norm(xs - xe) / np.linalg.norm(xe)),
               sum(1 for e in plan.exchanges if e["hss"]), len(plan.exchanges), be.moved,
               max((len(be.nodes[e["node"]].bnd) for e in plan.exchanges), default=0), OM.count_kinds(Fs), HS.__name__))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc()))


@pytest.mark.parametrize("world,name", [(2, ((16, 16, 16), "poisson", 128, 3, 32, 1e-6)), (4, ((16, 16, 16), "poisson", 128, 3, 32, 1e-6)),
                                        (2, ((12, 12, 10), "helmholtz", 64, 3, 24, 1e-5)), (2, ((32, 32, 32), "poisson", 512, 3, 128, 1e-4))])
def test_matrix_free_flow_over_gloo(world, name):
    """The reference's compressed data flow over ranks (src/factorization.jl:78-112,126-140): `run_numeric` / `run_solve` of dist.py -- the code
    the GPU path runs -- with an oracle-backed backend.  Every join above the cut ships a packed HSS matrix (byte count first), never a dense
    block, and the result is the SERIAL oracle's to round-off (same fronts, same seeds; the pack / unpack round trip is exact)."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 3 + world * 41 + len(name[1])) % 2000
    procs = [ctx.Process(target=_worker_mf, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=400) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in sorted(res):
        assert not isinstance(r[1], str), r[1]
        rank, err_serial, err_exact, nhss, nex, moved, nbmax, kinds = r[:8]
        assert err_serial < 1e-11, (rank, err_serial)
        assert err_exact < 100 * name[5]  # and the flow itself follows the tolerance
        assert nhss == nex == world - 1
        assert kinds.get("mf", 0) == 3
    moved = [r[5] for r in res]
    assert max(moved) > 0 and sum(moved) == 2 * sum(moved) // 2  # every byte that left one rank arrived at another
    print("bytes of generators per rank:", moved)


def test_mf_plan_over_ranks_and_refused_combinations(hs):
    """hs_plan (host only): with hs_options.mf and nranks > 1 the flagged joins are HSS exchanges (round 2 dropped mf silently with nranks > 1);
    combinations the library cannot honour are refused with HS_ERR_UNSUPPORTED / ArgumentError instead of falling back to another flow."""
    import ctypes as C

    from helpers import prepare

    P = prepare(hs, (16, 16, 16), kind="poisson", nmax=128)
    L = hs._lib.lib()
    kw = dict(swlevel=3, swsize=8, atol=1e-4, rtol=1e-4, leafsize=64)
    for world in (2, 4):
        for rank in range(world):
            h = hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=rank, nranks=world, mf="dense", **kw)
            try:
                n = L.hs_num_exchanges(h)
                assert n == world - 1
                out6 = (C.c_int64 * 6)()
                for k in range(n):
                    hs._lib.check(L.hs_exchange_info(h, k, out6))
                    assert L.hs_exchange_kind(h, k) == 1 and out6[5] == 0  # generators, no dense element count
                out8 = (C.c_int64 * 8)()
                hs._lib.check(L.hs_flow_info(h, out8))
                assert out8[6] == world
            finally:
                L.hs_free(h)
        # without mf the same joins ship dense blocks
        h = hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=world, **kw)
        try:
            assert all(L.hs_exchange_kind(h, k) == 0 for k in range(L.hs_num_exchanges(h)))
        finally:
            L.hs_free(h)
    with pytest.raises(hs.UnsupportedError, match="dist_top"):
        hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=2, mf="dense", dist_top=True, **kw)
    with pytest.raises(hs.UnsupportedError, match="flagged for compression"):
        hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=4, dist_top=True, **kw)  # level 2 is above the cut of 4 ranks and flagged
    hs._lib.lib().hs_free(hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=2, dist_top=True, **kw))  # 2 ranks: only the root, never flagged
    with pytest.raises(hs.UnsupportedError, match="hss_d"):
        hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=2, hss_min=1024, **kw)
    with pytest.raises(hs.UnsupportedError, match="split"):
        hs.dist.plan_only(P["A"], P["nd"], P["nd_loc"], rank=0, nranks=1, mf="dense", split_size=256, **kw)
    # hs_options.mf outside 0:3 straight through the C ABI
    opts = hs.SolverOptions(**kw).to_c()
    opts.mf = 7
    t, _keep = hs.dist._tree_struct(P["nd"], P["nd_loc"])
    hh = C.c_void_p()
    assert L.hs_plan(0, P["A"].shape[0], C.byref(t), C.byref(opts), 0, 1, C.byref(hh)) == hs._lib.HS_ERR_ARGUMENT
    assert b"mf must be in 0:3" in L.hs_last_error()
