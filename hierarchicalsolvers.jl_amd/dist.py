"""Staged (analyze / numeric / solve) driver with device-resident inputs, single- or multi-rank.

One process per GPU.  The elimination tree is cut at level ``p+1`` (``nranks = 2^p``): the subtrees
below go one per rank -- the reference factors them one after the other although they are independent
(``src/factorization.jl:20-21``) -- and every front above the cut is eliminated by the first rank of
its group.  Sibling Schur complements are *concatenated* into the parent front, never summed
(``src/factorization.jl:118-121``), so a join needs no reduction: the right child's owner sends its
Schur complement (``nb x nb``) to the parent's owner, point to point.  ``ldiv!`` mirrors it with
``nb``-vectors: up in the forward sweep, down in the backward sweep, then one all-reduce of the
disjoint solution pieces.

The schedule below is written against a small *backend* interface so that the same code runs

* on GPUs: :class:`HipBackend` (C ABI of ``libhs_solver.so``; ``torch.distributed`` backend ``nccl`` =
  RCCL over xGMI moves the buffers, which are torch tensors registered with the library), and
* in the CPU tests: an oracle-backed backend over ``gloo`` (tests/test_dist_cpu.py), which checks the
  partition / exchange logic without a GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import scipy.sparse as sp

from . import _lib
from .nesteddissection import flatten_tree
from .solver import SolverOptions, chkopts

__all__ = ["Plan", "HipBackend", "run_numeric", "run_solve", "StagedSolver", "plan_only", "TorchComm", "LibComm"]


class Plan:
    """What the host layer needs to know about the partition: levels, cut and the cross-rank edges."""

    def __init__(self, nlevels, cut_level, exchanges, nranks):
        self.nlevels = int(nlevels)
        self.cut_level = int(cut_level)
        self.exchanges = list(exchanges)  # dicts: node, level (of the child), src, dst, nb, nelems
        self.nranks = int(nranks)

    def at_child_level(self, lv):
        return [e for e in self.exchanges if e["level"] == lv]


class _NullComm:
    """nranks == 1: nothing crosses ranks."""

    def send(self, t, dst):
        raise RuntimeError("no peers")

    def recv(self, t, src):
        raise RuntimeError("no peers")

    def all_reduce(self, t):
        return t

    def barrier(self):
        pass


class TorchComm:
    """torch.distributed point-to-point + all-reduce (backend nccl = RCCL on ROCm; gloo on CPU).
    Complex tensors travel as their real view (same bytes; not every backend takes complex dtypes)."""

    def __init__(self):
        import torch
        import torch.distributed as dist

        self.dist, self.torch = dist, torch
        # gloo moves host memory: device tensors are staged through the host (rehearsals of the N > 1
        # path on a single GPU; production uses nccl = RCCL, device to device over xGMI)
        self.stage = dist.get_backend() == "gloo"

    def _r(self, t):
        return self.torch.view_as_real(t) if t.is_complex() else t

    def send(self, t, dst):
        t = self._r(t)
        self.dist.send(t.cpu() if (self.stage and t.is_cuda) else t, dst=dst)

    def recv(self, t, src):
        r = self._r(t)
        if self.stage and r.is_cuda:
            h = self.torch.empty(r.shape, dtype=r.dtype)
            self.dist.recv(h, src=src)
            r.copy_(h)
        else:
            self.dist.recv(r, src=src)

    def all_reduce(self, t):
        r = self._r(t)
        if self.stage and r.is_cuda:
            h = r.cpu()
            self.dist.all_reduce(h)
            r.copy_(h)
        else:
            self.dist.all_reduce(r)
        return t

    def barrier(self):
        self.dist.barrier()

    def exchange(self, send_t, recv_t, peer):
        """Pairwise swap with ``peer`` (both sides call it): one batched isend + irecv, so neither side has to go first."""
        dist, torch = self.dist, self.torch
        s, r = self._r(send_t), self._r(recv_t)
        if self.stage and s.is_cuda:
            hs_, hr = s.cpu(), torch.empty(r.shape, dtype=r.dtype)
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, hs_, peer), dist.P2POp(dist.irecv, hr, peer)]):
                q.wait()
            r.copy_(hr)
        else:
            for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, s, peer), dist.P2POp(dist.irecv, r, peer)]):
                q.wait()


class LibComm:
    """The library's own communicator (``hs_comm``, include/hs_solver.h) for ``dist_top`` factorizations: RCCL over xGMI when the
    process group is ``nccl`` (the 128-byte id travels through torch.distributed once; afterwards the library enqueues grouped
    ncclSend / ncclRecv on its own stream), host-staged over the process group otherwise (gloo: the single-GPU rehearsals)."""

    def __init__(self, rank, nranks, device=None):
        import torch
        import torch.distributed as dist

        self.L = _lib.lib()
        self._h = C.c_void_p()
        self.rank, self.nranks = int(rank), int(nranks)
        if dist.get_backend() == "nccl":
            idt = torch.zeros(128, dtype=torch.uint8, device=device if device is not None else "cuda")
            if rank == 0:
                buf = (C.c_char * 128)()
                _lib.check(self.L.hs_comm_unique_id(C.cast(buf, C.c_void_p)))
                idt.copy_(torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8))
            dist.broadcast(idt, 0)
            raw = (C.c_char * 128).from_buffer_copy(idt.cpu().numpy().tobytes())
            _lib.check(self.L.hs_comm_create_rccl(C.cast(raw, C.c_void_p), self.rank, self.nranks, C.byref(self._h)))
        else:
            self._cb = _lib.HS_TRANSFER_FN(self._transfer)  # keep the trampoline alive as long as the communicator
            _lib.check(self.L.hs_comm_create_host(self._cb, None, self.rank, self.nranks, C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def kind(self):
        return self.L.hs_comm_kind(self._h).decode()

    def selftest(self, nbytes=1 << 20):
        _lib.check(self.L.hs_comm_selftest(self._h, int(nbytes)))

    def bandwidth(self, nbytes=1 << 26, reps=4):
        """GB/s each rank sends in a ring shift (device to device through this transport; every rank calls it)."""
        g = C.c_double(0.0)
        _lib.check(self.L.hs_comm_bandwidth(self._h, int(nbytes), int(reps), C.byref(g)))
        return g.value

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.L.hs_comm_free(h)

    def __del__(self):
        self.close()

    @staticmethod
    def _transfer(user, nsend, speer, sbuf, sbytes, nrecv, rpeer, rbuf, rbytes):
        """hs_transfer_fn: one host message per peer and direction, moved over the default process group."""
        try:
            import torch
            import torch.distributed as dist

            def view(ptr, nb):
                return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * int(nb)).from_address(ptr)))

            ops = [dist.P2POp(dist.isend, view(sbuf[k], sbytes[k]), int(speer[k])) for k in range(nsend)]
            ops += [dist.P2POp(dist.irecv, view(rbuf[k], rbytes[k]), int(rpeer[k])) for k in range(nrecv)]
            if ops:
                for q in dist.batch_isend_irecv(ops):
                    q.wait()
            return 0
        except Exception:  # pragma: no cover - reported through the library's error string
            import traceback

            traceback.print_exc()
            return 1


def run_numeric_dist(backend):
    """``dist_top``: the library moves everything itself (Schur complements at the joins, block columns inside the groups)."""
    backend.numeric_begin()
    backend.numeric_levels(backend.plan.nlevels, 0)
    backend.numeric_end()


def run_solve_dist(backend, plan, rank, comm, b):
    """``ldiv!`` with the fronts above the cut replicated on their groups: the forward sweep swaps the two children's boundary values
    between partner ranks before each join, the backward sweep needs nothing (every rank already holds the values of all its
    ancestors' DOFs), one all-reduce of the disjoint owned pieces at the end."""
    L, cut = plan.nlevels, plan.cut_level
    backend.fwd(b, L, cut)
    for lv in range(cut - 1, 0, -1):
        mine = [e for e in plan.at_child_level(lv + 1) if e["src"] == rank]
        theirs = [e for e in plan.at_child_level(lv + 1) if e["dst"] == rank]
        for em, et in zip(mine, theirs):
            sbuf = backend.pack_bnd(em["node"], b)
            rbuf = backend.bnd_buffer(et["node"])
            backend.sync()
            comm.exchange(sbuf, rbuf, em["dst"])
            backend.comm_sync()
            backend.unpack_bnd(et["node"], b, rbuf)
        backend.fwd(b, lv, lv)
    backend.bwd(b, 1, L)
    out = backend.extract_owned(b)
    backend.sync()
    comm.all_reduce(out)
    backend.comm_sync()
    backend.assign(b, out)
    return b


def run_numeric(backend, plan, rank, comm, trace=None):
    """Numeric factorization: rank-local subtrees, then the fronts above the cut level by level,
    receiving the remote child's Schur complement before each join.

    ``trace``: a list to which an INSTRUMENTED run appends ``(what, seconds)`` pairs -- the host waits for the device after every stage
    (``backend.host_sync``), so the pairs are this rank's compute ("levels ..."), transfer ("send" / "recv": includes waiting for the
    peer) and nothing else; a normal run (``trace=None``) never blocks the host."""
    import time as _time

    t_last = [_time.perf_counter()]

    def lap(what):
        if trace is None:
            return
        backend.host_sync()
        now = _time.perf_counter()
        trace.append((what, now - t_last[0]))
        t_last[0] = now

    if plan.nranks > 1:
        backend.comm_sync()  # a send of the previous factorization may still read a Schur buffer this one overwrites
    backend.numeric_begin()
    L, cut = plan.nlevels, plan.cut_level
    lap("begin")
    backend.numeric_levels(L, cut)
    lap(f"levels {L}..{cut} (rank-local subtree)")
    for lv in range(cut - 1, 0, -1):
        for e in plan.at_child_level(lv + 1):
            if e.get("hss"):
                # the matrix-free flow over ranks (hs_options.mf): the child's Schur complement is an HssMatrix (src/factorization.jl:78-112,126-140);
                # its generators cross packed into one buffer -- the byte count first, it is known only after the compression
                if e["src"] == rank:
                    buf = backend.schur_hss_pack(e["node"])
                    comm.send(backend.size_tensor(buf.numel()), e["dst"])
                    comm.send(buf, e["dst"])
                    backend.comm_sync()
                    backend.note_transfer(buf.numel())
                elif e["dst"] == rank:
                    nbytes = backend.size_tensor(0)
                    comm.recv(nbytes, e["src"])
                    backend.comm_sync()
                    buf = backend.byte_buffer(int(nbytes.item()))
                    comm.recv(buf, e["src"])
                    backend.comm_sync()
                    backend.schur_hss_unpack(e["node"], buf)
                    backend.note_transfer(buf.numel())
                if rank in (e["src"], e["dst"]):
                    lap(("send" if e["src"] == rank else "recv") + f" HSS generators of node {e['node']} ({buf.numel()} bytes)")
                continue
            if e["src"] == rank:
                backend.sync()
                comm.send(backend.schur_tensor(e["node"]), e["dst"])
                backend.comm_sync()  # the communicator's stream is unknown to the library: the buffer is free again after this
                lap(f"send dense S of node {e['node']} ({e['nelems']} elements)")
            elif e["dst"] == rank:
                comm.recv(backend.schur_tensor(e["node"]), e["src"])
                backend.comm_sync()
                lap(f"recv dense S of node {e['node']} ({e['nelems']} elements)")
        backend.numeric_levels(lv, lv)
        lap(f"level {lv}")
    backend.numeric_levels(0, 0)  # pseudo-root (root with a boundary), owned by rank 0
    backend.numeric_end()
    lap("end")


def run_solve(backend, plan, rank, comm, b):
    """``ldiv!`` in place on the (replicated) device vector ``b``; on return every rank holds the solution."""
    L, cut = plan.nlevels, plan.cut_level
    backend.fwd(b, L, cut)
    for lv in range(cut - 1, 0, -1):
        for e in plan.at_child_level(lv + 1):
            if e["src"] == rank:
                buf = backend.pack_bnd(e["node"], b)
                backend.sync()
                comm.send(buf, e["dst"])
                backend.comm_sync()  # the next pack of this buffer must not overtake the send
            elif e["dst"] == rank:
                buf = backend.bnd_buffer(e["node"])
                comm.recv(buf, e["src"])
                backend.comm_sync()
                backend.unpack_bnd(e["node"], b, buf)
        backend.fwd(b, lv, lv)
    backend.fwd(b, 0, 0)
    backend.bwd(b, 0, 1)
    for lv in range(1, cut):
        for e in plan.at_child_level(lv + 1):
            if e["dst"] == rank:  # parent's owner hands the boundary values back down
                buf = backend.pack_bnd(e["node"], b)
                backend.sync()
                comm.send(buf, e["src"])
                backend.comm_sync()
            elif e["src"] == rank:
                buf = backend.bnd_buffer(e["node"])
                comm.recv(buf, e["dst"])
                backend.comm_sync()
                backend.unpack_bnd(e["node"], b, buf)
        backend.bwd(b, lv + 1, lv + 1)
    backend.bwd(b, cut + 1, L)
    if plan.nranks > 1:
        out = backend.extract_owned(b)
        backend.sync()
        comm.all_reduce(out)
        backend.comm_sync()
        backend.assign(b, out)
    return b


def _tree_struct(nd, nd_loc):
    flat = flatten_tree(nd, nd_loc)
    t = _lib.hs_tree()
    t.nnodes = flat["nnodes"]
    for k in ("left", "right", "int_ptr", "int_idx", "bnd_ptr", "bnd_idx", "iloc_ptr", "iloc_idx", "bloc_ptr", "bloc_idx"):
        flat[k] = np.ascontiguousarray(flat[k], dtype=np.int64)
        setattr(t, k, flat[k].ctypes.data_as(_lib.p_i64))
    return t, flat


def plan_only(A, nd, nd_loc, rank=0, nranks=1, opts=None, **kw):
    """Host-side plan (``hs_plan``): returns the raw handle (free it with ``hs_free``); no GPU needed."""
    opts = (opts or SolverOptions(swlevel=0)).copy(**kw)
    chkopts(opts)
    t, _keep = _tree_struct(nd, nd_loc)
    co = opts.to_c()
    h = C.c_void_p()
    _lib.check(_lib.lib().hs_plan(int(np.iscomplexobj(A.data)), A.shape[0], C.byref(t), C.byref(co), rank, nranks, C.byref(h)))
    return h


class HipBackend:
    """The C ABI behind the schedule.  Device buffers that cross ranks are torch tensors."""

    def __init__(self, A, nd, nd_loc, opts=None, rank=0, nranks=1, device=None, libcomm=None, **kw):
        import torch

        self.torch = torch
        opts = (opts or SolverOptions()).copy(**kw)
        chkopts(opts)
        A = sp.csc_matrix(A)
        A.sort_indices()
        self.n = A.shape[0]
        self.is_c = bool(np.iscomplexobj(A.data))
        self.np_dtype = np.complex128 if self.is_c else np.float64
        self.t_dtype = torch.complex128 if self.is_c else torch.float64
        self.device = torch.device(device if device is not None else "cuda:0")
        torch.cuda.set_device(self.device)
        self.rank, self.nranks = int(rank), int(nranks)
        colptr = np.ascontiguousarray(A.indptr, dtype=np.int64) + 1
        rowval = np.ascontiguousarray(A.indices, dtype=np.int64) + 1
        flat = flatten_tree(nd, nd_loc)
        t = _lib.hs_tree()
        t.nnodes = flat["nnodes"]
        for k in ("left", "right", "int_ptr", "int_idx", "bnd_ptr", "bnd_idx", "iloc_ptr", "iloc_idx", "bloc_ptr", "bloc_idx"):
            flat[k] = np.ascontiguousarray(flat[k], dtype=np.int64)
            setattr(t, k, flat[k].ctypes.data_as(_lib.p_i64))
        co = opts.to_c()
        h = C.c_void_p()
        self.L = _lib.lib()
        _lib.check(self.L.hs_analyze(int(self.is_c), self.n, colptr.ctypes.data_as(_lib.p_i64), rowval.ctypes.data_as(_lib.p_i64),
                                     C.byref(t), C.byref(co), self.rank, self.nranks, C.byref(h)))
        self._h = h
        ex = []
        out6 = (C.c_int64 * 6)()
        for k in range(self.L.hs_num_exchanges(h)):
            _lib.check(self.L.hs_exchange_info(h, k, out6))
            ex.append(dict(node=out6[0], level=out6[1], src=out6[2], dst=out6[3], nb=out6[4], nelems=out6[5], hss=int(self.L.hs_exchange_kind(h, k)) == 1))
        self.plan = Plan(self.L.hs_nlevels(h), self.L.hs_cut_level(h), ex, self.nranks)
        # exchange buffers live in torch memory and are registered with the library
        self._schur, self._bnd = {}, {}
        self.dist_top = bool(opts.dist_top) and self.nranks > 1
        self.libcomm = None
        if self.dist_top:  # the library moves the Schur complements and the block columns itself (csrc/hs_dist.h)
            self._own_libcomm = libcomm is None
            self.libcomm = libcomm if libcomm is not None else LibComm(self.rank, self.nranks, self.device)
            _lib.check(self.L.hs_set_comm(h, self.libcomm.handle))
        for e in ex:
            if self.rank in (e["src"], e["dst"]):
                if not self.dist_top and not e["hss"]:
                    s = torch.zeros(max(int(e["nelems"]), 1), dtype=self.t_dtype, device=self.device)
                    _lib.check(self.L.hs_set_schur_buffer(h, e["node"], C.c_void_p(s.data_ptr())))
                    self._schur[e["node"]] = s
                self._bnd[e["node"]] = torch.zeros(max(int(e["nb"]), 1), dtype=self.t_dtype, device=self.device)
        self.values = torch.from_numpy(np.ascontiguousarray(A.data, dtype=self.np_dtype)).to(self.device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self.L.hs_free(h)
        lc, self.libcomm = getattr(self, "libcomm", None), None
        if lc is not None and getattr(self, "_own_libcomm", False):
            lc.close()

    # -- numeric --------------------------------------------------------------------------------------------
    def set_values(self, values):
        """New values for the same sparsity pattern (a torch tensor on this device, CSC order)."""
        self.values = values

    def numeric_begin(self):
        # hs_numeric_begin copies `values` on the library's own stream, which is ordered against the legacy default stream
        # only: whatever produced `values` on the caller's current torch stream must have finished first
        self.torch.cuda.current_stream(self.device).synchronize()
        _lib.check(self.L.hs_numeric_begin(self._h, C.c_void_p(self.values.data_ptr()), 1))

    def numeric_levels(self, lv_from, lv_to):
        _lib.check(self.L.hs_numeric_levels(self._h, lv_from, lv_to))

    def numeric_end(self):
        _lib.check(self.L.hs_numeric_end(self._h))

    def schur_tensor(self, node):
        return self._schur[node]

    # -- Schur complements that cross ranks as HSS matrices (hs_options.mf) -----------------------------------------------------
    def schur_hss_pack(self, node):
        nb = C.c_int64(0)
        _lib.check(self.L.hs_schur_pack_size(self._h, node, C.byref(nb)))
        buf = self.torch.empty(int(nb.value), dtype=self.torch.uint8, device=self.device)
        _lib.check(self.L.hs_schur_pack(self._h, node, C.c_void_p(buf.data_ptr()), int(nb.value), None))  # returns when the buffer is complete
        return buf

    def schur_hss_unpack(self, node, buf):
        _lib.check(self.L.hs_schur_unpack(self._h, node, C.c_void_p(buf.data_ptr()), int(buf.numel()), None))

    def size_tensor(self, value):
        return self.torch.tensor([int(value)], dtype=self.torch.int64, device=self.device)

    def byte_buffer(self, nbytes):
        return self.torch.empty(int(nbytes), dtype=self.torch.uint8, device=self.device)

    def note_transfer(self, nbytes):
        self.hss_bytes_moved = getattr(self, "hss_bytes_moved", 0) + int(nbytes)

    def flow_info(self):
        out = (C.c_int64 * 8)()
        _lib.check(self.L.hs_flow_info(self._h, out))
        keys = ("mf", "mf_fronts", "hss_schur_fronts", "lowrank_fronts", "hss_d_fronts", "group_fronts", "nranks", "sliced_fronts")
        return dict(zip(keys, (int(v) for v in out)))

    def sync(self):
        """Library stream -> communicator: the stream the communicator works on (torch's current one) waits for everything the library has
        enqueued, so a send reads complete data.  An event, not a host synchronisation (round 2 blocked the host on the whole device here)."""
        _lib.check(self.L.hs_stream_order(self._h, self._stream(), 0))

    def comm_sync(self):
        """Communicator -> library: the library's stream waits for what the communicator enqueued (a receive into a Schur / boundary buffer,
        or a send still reading one that the next factorization overwrites)."""
        _lib.check(self.L.hs_stream_order(self._h, self._stream(), 1))

    def host_sync(self):
        """Block the host until the device is idle (instrumented runs only: `run_numeric(..., trace=[])`)."""
        self.torch.cuda.synchronize(self.device)

    # -- solve ------------------------------------------------------------------------------------------------
    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def fwd(self, b, lv_from, lv_to):
        _lib.check(self.L.hs_solve_fwd_levels(self._h, self._p(b), lv_from, lv_to, self._stream()))

    def bwd(self, b, lv_from, lv_to):
        _lib.check(self.L.hs_solve_bwd_levels(self._h, self._p(b), lv_from, lv_to, self._stream()))

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def pack_bnd(self, node, b):
        buf = self._bnd[node]
        _lib.check(self.L.hs_pack_bnd(self._h, node, self._p(b), self._p(buf), self._stream()))
        return buf

    def bnd_buffer(self, node):
        return self._bnd[node]

    def unpack_bnd(self, node, b, buf):
        _lib.check(self.L.hs_unpack_bnd(self._h, node, self._p(b), self._p(buf), self._stream()))

    def extract_owned(self, b):
        out = self.torch.zeros_like(b)
        _lib.check(self.L.hs_extract_owned(self._h, self._p(b), self._p(out), self._stream()))
        return out

    def assign(self, b, out):
        b.copy_(out)

    def ldiv_dev(self, b):
        """``ldiv!(F, b)`` inside the library on the caller's current stream (single rank, or ``dist_top`` with the library's communicator)."""
        f = self.L.hs_ldiv_dev_z if self.is_c else self.L.hs_ldiv_dev_d
        _lib.check(f(self._h, self._p(b), self.n, self._p(b), self.n, self.n, 1, self._stream()))
        return b

    def stats(self):
        st = _lib.hs_stats()
        _lib.check(self.L.hs_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in st._fields_}


class StagedSolver:
    """analyze once, then ``numeric()`` / ``solve(b)`` with everything resident in HBM."""

    def __init__(self, A, nd, nd_loc, opts=None, rank=0, nranks=1, comm=None, device=None, libcomm=None, **kw):
        self.backend = HipBackend(A, nd, nd_loc, opts, rank, nranks, device, libcomm, **kw)
        self.plan = self.backend.plan
        self.rank = rank
        self.comm = comm if comm is not None else (_NullComm() if nranks == 1 else TorchComm())
        # dist_top: ldiv! runs inside the library too (hs_ldiv_dev_*: swaps at the joins and the final gather through the library's
        # communicator); host_solve = True drives the same sweeps level by level from here (run_solve_dist, torch.distributed)
        self.host_solve = False

    def numeric(self, values=None, trace=None):
        if values is not None:
            self.backend.set_values(values)
        if self.backend.dist_top:
            run_numeric_dist(self.backend)
        else:
            run_numeric(self.backend, self.plan, self.rank, self.comm, trace)

    def solve(self, b):
        """In place on a device tensor ``b`` of length n (every rank passes the same right-hand side)."""
        if self.backend.dist_top:
            if not self.host_solve:
                return self.backend.ldiv_dev(b)
            return run_solve_dist(self.backend, self.plan, self.rank, self.comm, b)
        return run_solve(self.backend, self.plan, self.rank, self.comm, b)

    def stats(self):
        return self.backend.stats()
