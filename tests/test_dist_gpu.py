"""N > 1 path on real hardware, rehearsed on ONE GPU: two (and four) ranks share cuda:0, the C library
runs each rank's fronts, buffers cross ranks over gloo staged through the host (RCCL refuses two ranks
on one device).  Exercises hs_analyze(rank, nranks), hs_set_schur_buffer, hs_numeric_levels,
hs_pack_bnd/hs_unpack_bnd, hs_extract_owned and the schedule of dist.py end to end."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, name, q, fopts=None, env=None):
    try:
        os.environ.update(env or {})
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import scipy.sparse.linalg as spla

        import hsamd

        hs = hsamd.load()
        from helpers import prepare

        P = prepare(hs, name, rhs="randn")
        dev = torch.device("cuda:0")
        S = hs.dist.StagedSolver(P["A"], P["nd"], P["nd_loc"], rank=rank, nranks=world, device=dev, **(fopts or dict(swlevel=0)))
        errs = []
        # SuperLU takes minutes on the larger / complex 3-D cases: there the bar is the residual (the exact path sits at round-off level)
        xr = spla.splu(P["A"]).solve(P["b"]) if (P["A"].shape[0] <= 40000 and not np.iscomplexobj(P["A"].data)) or P["A"].shape[0] <= 5000 else None
        for rep in range(2):  # numeric twice: the second pass re-uses every buffer
            S.host_solve = rep == 1  # dist_top: ldiv! inside the library first, then the host-driven sweeps (dist.run_solve_dist)
            S.numeric()
            b = torch.from_numpy(np.ascontiguousarray(P["b"])).to(dev)
            S.solve(b)
            x = b.cpu().numpy()
            if xr is None:
                res = float(np.linalg.norm(P["A"] @ x - P["b"]) / np.linalg.norm(P["b"]))
                errs.append(res * 1e-2 if res < 1e-11 else max(res, 1.0))  # reported on the scale of the solution-error bar (1e-10)
            else:
                errs.append(float(np.linalg.norm(x - xr) / np.linalg.norm(xr)))
        nmine = sum(1 for k in range(len(hs.postorder_nodes(P["nd"]))) if S.backend.L.hs_node_owner(S.backend._h, k) == rank)
        if S.backend.libcomm is not None:
            S.backend.libcomm.selftest(1 << 16)  # ring shift through the communicator the factorization used
            assert S.backend.libcomm.bandwidth(1 << 20, 2) > 0.0
        q.put((rank, max(errs), nmine))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), 0))


def _run_ranks(world, args_of_rank, timeout=400):
    """Start one process per rank and collect one result each.  A rank that fails reports at once; its peers would wait for it in a
    receive for ever, so they are terminated as soon as the first error arrives."""
    import queue as pyqueue
    import time

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=args_of_rank(r, q)) for r in range(world)]
    for p in procs:
        p.start()
    res, t0, failed = [], time.time(), None
    while len(res) < world and failed is None:
        try:
            r = q.get(timeout=2.0)
            res.append(r)
            if isinstance(r[1], str):
                failed = r
        except pyqueue.Empty:
            if time.time() - t0 > timeout:
                failed = (-1, f"timeout after {timeout} s; got {res}", 0)
            elif any(p.exitcode not in (None, 0) for p in procs):
                failed = (-1, f"a rank died: exit codes {[p.exitcode for p in procs]}; got {res}", 0)
    if failed is not None:
        for p in procs:
            if p.is_alive():
                p.terminate()
    for p in procs:
        p.join(timeout=60)
    assert failed is None, failed
    return sorted(res)


@pytest.mark.parametrize("world,name", [(2, "poisson2d_p1_h64_nmax100"), (4, "helmholtz2d_p1_h64_nmax100"), (2, "poisson3d_32")])
def test_two_ranks_one_gpu(world, name):
    port = 29500 + (os.getpid() * 11 + world * 17 + len(name)) % 2000
    res = _run_ranks(world, lambda r, q: (r, world, port, name, q))
    total = 0
    for rank, err, nmine in res:
        assert not isinstance(err, str), err
        assert err < 1e-10, (rank, err)
        total += nmine
    assert total > 0


def test_two_ranks_compressed_fronts():
    """Compressed fronts (low-rank off-diagonal blocks) above and below the rank cut: the joins ship dense Schur
    complements exactly as on the dense path; the solution error is O(tol)."""
    world, name, tol = 2, "poisson3d_32", 1e-8
    port = 29500 + (os.getpid() * 13 + 977) % 2000
    fopts = dict(swlevel=3, swsize=8, atol=tol, rtol=tol)
    res = _run_ranks(world, lambda r, q: (r, world, port, name, q, fopts))
    for rank, err, nmine in res:
        assert not isinstance(err, str), err
        assert err < 1e3 * tol, (rank, err)


def _mf_worker(rank, world, port, name, q, fopts):
    """Matrix-free flow over ranks (hs_options.mf, dist_top = 0): the joins ship HSS generators.  Reports the solution error against SuperLU,
    the flow the library says it ran, the bytes that crossed and hs_maxrank."""
    try:
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import torch
        import torch.distributed as dist

        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        import scipy.sparse.linalg as spla

        import hsamd

        hs = hsamd.load()
        from helpers import prepare

        P = prepare(hs, name[0], rhs="randn", **name[1])
        dev = torch.device("cuda:0")
        S = hs.dist.StagedSolver(P["A"], P["nd"], P["nd_loc"], rank=rank, nranks=world, device=dev, **fopts)
        xr = spla.splu(P["A"]).solve(P["b"])
        errs = []
        for rep in range(2):  # numeric twice: every HSS object of the first pass is released and rebuilt
            S.numeric()
            b = torch.from_numpy(np.ascontiguousarray(P["b"])).to(dev)
            S.solve(b)
            errs.append(float(np.linalg.norm(b.cpu().numpy() - xr) / np.linalg.norm(xr)))
        fi = S.backend.flow_info()
        kinds = [bool(e["hss"]) for e in S.plan.exchanges]
        q.put((rank, errs, dict(flow=fi, hss_exchanges=sum(kinds), exchanges=len(kinds), moved=getattr(S.backend, "hss_bytes_moved", 0),
                               maxrank=int(S.backend.L.hs_maxrank(S.backend._h)))))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), 0))


@pytest.mark.parametrize("world,mf", [(2, "dense"), (2, "block"), (4, "dense"), (4, "hss")])
def test_matrix_free_flow_over_ranks(world, mf):
    """hs_options.mf with nranks > 1 (round 2 silently fell back to the dense-S flow): the flagged fronts above AND below the rank cut hand their
    Schur complements on as HSS matrices; where the parent lives on another rank the generators cross packed into one buffer
    (hs_schur_pack / hs_schur_unpack) instead of a dense nb x nb block (src/factorization.jl:78-112,126-140).  The error is the
    single-rank flow's (same tolerance, same algorithm; the random draws differ per front only through the seeds, which are per node)."""
    import queue as pyqueue  # noqa: F401

    import torch.multiprocessing as mp

    tol = 1e-6
    name = ((32, 32, 32), dict(kind="poisson", nmax=512))
    fopts = dict(mf=mf, swlevel=3, swsize=8, atol=tol, rtol=tol, leafsize=128)
    port = 29500 + (os.getpid() * 7 + world * 31 + len(mf)) % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_mf_worker, args=(r, world, port, name, q, fopts)) for r in range(world)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in range(world):
            r = q.get(timeout=600)
            assert not isinstance(r[1], str), r[1]
            res.append(r)
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.terminate()
    # the single-rank flow on the same problem, in this process
    import scipy.sparse.linalg as spla

    import hsamd
    from helpers import prepare, relerr

    hs = hsamd.load()
    P = prepare(hs, name[0], rhs="randn", **name[1])
    F1 = hs.factor(P["A"], P["nd"], P["nd_loc"], **fopts)
    e1 = relerr(hs.ldiv(F1, P["b"]), spla.splu(P["A"]).solve(P["b"]))
    r1 = hs.maxrank(F1)
    print(f"mf={mf} world={world}: single rank err {e1:.2e} maxrank {r1}; ranks:", [(r[0], [f'{e:.2e}' for e in r[1]], r[2]) for r in sorted(res)])
    nmf = 0
    for rank, errs, info in res:
        assert errs[0] <= 3.0 * e1 + 1e-9 and max(errs) <= 5.0 * e1 + 1e-9 and max(errs) < 50 * tol, (rank, errs, e1)
        # (the second numeric pass starts every compression from the sample count the first one ended with: another draw -- measured 0.9x and 3.6x
        # the single-rank error of the first pass on the 4-rank case, i.e. 3 and 13 times the tolerance)
        assert info["hss_exchanges"] == info["exchanges"] == world - 1  # every join above the cut ships generators, none a dense block
        assert info["flow"]["nranks"] == world and info["flow"]["mf"] == {"dense": 1, "hss": 2, "block": 3}[mf]
        nmf += info["flow"]["mf_fronts"]
    assert nmf == 3  # levels 1-2 of the tree hold the three matrix-free fronts (parents of two flagged children), wherever they live
    moved = max(info["moved"] for _, _, info in res)
    nb_top = len(P["nd"].left.bnd)
    print(f"bytes of generators moved by the busiest rank (two passes): {moved}; one dense top-level S: {nb_top * nb_top * 8}")
    assert 0 < moved < 2 * nb_top * nb_top * 8 * (world - 1)  # two passes; below the dense blocks the dense-S flow ships (ranks ~ nb/4 here)
    assert max(info["maxrank"] for _, _, info in res) <= 1.25 * r1 + 8


@pytest.mark.parametrize("world,name,nb,period", [(2, "poisson3d_32", 256, 1), (4, "poisson3d_32", 256, 1), (2, "helmholtz3d_32", 256, 1),
                                                  (4, "helmholtz2d_p1_h64_nmax100", 256, 1), (2, "poisson3d_32", 1024, 1), (2, "poisson3d_64", 1024, 1),
                                                  (2, "poisson3d_32", 256, 2), (2, "poisson3d_64", 256, 4), (2, "poisson3d_64", 512, 2), (4, "poisson3d_64", 512, 2)])
def test_group_fronts_one_gpu(world, name, nb, period):
    """hs_options.dist_top: the fronts above the rank cut are eliminated by all ranks of their group (csrc/hs_dist.h) -- block columns of
    HS_DIST_NB interior DOFs dealt round-robin, factored by their owner, fanned out, boundary-column slices gathered at the end -- through
    the host-staged communicator (gloo moves the bytes; RCCL refuses several ranks on one device).  Same bar as the rank-local path:
    the solution agrees with SuperLU to 1e-10 after two numeric passes.  `period` > 1: several consecutive block columns per owner (the
    fan-out of one overlaps the factorization of the next)."""
    port = 29500 + (os.getpid() * 11 + world * 19 + len(name) + nb + period) % 2000
    res = _run_ranks(world, lambda r, q: (r, world, port, name, q, dict(swlevel=0, dist_top=True), {"HS_DIST_NB": str(nb), "HS_DIST_PERIOD": str(period)}))
    for rank, err, nmine in res:
        assert not isinstance(err, str), err
        assert err < 1e-10, (rank, err)


def test_rccl_communicator_single_rank():
    """The RCCL transport on the one GPU of this box: librccl is opened, a communicator of one rank created from a fresh id, and a grouped
    ncclSend / ncclRecv to itself moves a byte pattern on the library's stream (what every transfer of a multi-rank run is made of)."""
    import ctypes as C

    import torch

    import hsamd

    hs = hsamd.load()
    L = hs._lib.lib()
    torch.cuda.set_device(0)
    buf = (C.c_char * 128)()
    hs._lib.check(L.hs_comm_unique_id(C.cast(buf, C.c_void_p)))
    h = C.c_void_p()
    hs._lib.check(L.hs_comm_create_rccl(C.cast(buf, C.c_void_p), 0, 1, C.byref(h)))
    try:
        assert L.hs_comm_kind(h) == b"rccl"
        hs._lib.check(L.hs_comm_selftest(h, 1 << 20))
        hs._lib.check(L.hs_comm_selftest(h, 12345))
    finally:
        L.hs_comm_free(h)
