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


def _worker(rank, world, port, name, q, fopts=None):
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

        P = prepare(hs, name, rhs="randn")
        dev = torch.device("cuda:0")
        S = hs.dist.StagedSolver(P["A"], P["nd"], P["nd_loc"], rank=rank, nranks=world, device=dev, **(fopts or dict(swlevel=0)))
        errs = []
        for rep in range(2):  # numeric twice: the second pass re-uses every buffer
            S.numeric()
            b = torch.from_numpy(np.ascontiguousarray(P["b"])).to(dev)
            S.solve(b)
            x = b.cpu().numpy()
            xr = spla.splu(P["A"]).solve(P["b"])
            errs.append(float(np.linalg.norm(x - xr) / np.linalg.norm(xr)))
        nmine = sum(1 for k in range(len(hs.postorder_nodes(P["nd"]))) if S.backend.L.hs_node_owner(S.backend._h, k) == rank)
        q.put((rank, max(errs), nmine))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "ERR " + repr(e) + traceback.format_exc(), 0))


@pytest.mark.parametrize("world,name", [(2, "poisson2d_p1_h64_nmax100"), (4, "helmholtz2d_p1_h64_nmax100"), (2, "poisson3d_32")])
def test_two_ranks_one_gpu(world, name):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 11 + world * 17 + len(name)) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    total = 0
    for rank, err, nmine in sorted(res):
        assert not isinstance(err, str), err
        assert err < 1e-10, (rank, err)
        total += nmine
    assert total > 0


def test_two_ranks_compressed_fronts():
    """Compressed fronts (low-rank off-diagonal blocks) above and below the rank cut: the joins ship dense Schur
    complements exactly as on the dense path; the solution error is O(tol)."""
    import torch.multiprocessing as mp

    world, name, tol = 2, "poisson3d_32", 1e-8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() * 13 + 977) % 2000
    fopts = dict(swlevel=3, swsize=8, atol=tol, rtol=tol)
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, fopts)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, err, nmine in sorted(res):
        assert not isinstance(err, str), err
        assert err < 1e3 * tol, (rank, err)
