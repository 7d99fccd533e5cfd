// hs_comm.h -- the one data-movement primitive of the multi-rank elimination (include/hs_solver.h, "communicator").
//
// Everything that crosses ranks -- a factored block column fanned out to the group of a distributed front, the column
// slices of L^-1*P*Aib and of the Schur complement gathered inside the group, the Schur complements two sibling groups
// swap at a join (the reference concatenates them, src/factorization.jl:118-121: no reduction anywhere) -- is a set of
// point-to-point pieces, so the library needs exactly one operation: "send these device ranges to those ranks, receive
// those ranges from these ranks", ordered on a HIP stream.
//
//  * RcclComm: ncclSend / ncclRecv inside one ncclGroupStart/End on the world communicator (librccl is opened with dlopen when the
//    communicator is created; no sub-communicators are needed).  On MI355X the xGMI fabric is a full mesh of point-to-point links,
//    so a fan-out of one block column to the 7 peers drives 7 links at once -- a ring broadcast would drive one.
//  * HostComm: the same operation staged through host memory and handed to a callback of the host layer (MPI.jl in a Julia host,
//    torch.distributed/gloo in the rehearsals of tests/test_dist_gpu.py, where several ranks share one GPU and RCCL refuses to run).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "../../include/hs_solver.h"

struct HsPiece {
  int peer;      // world rank
  void* ptr;     // device address on THIS rank
  size_t bytes;
};

struct hs_comm {
  int rank = 0, nranks = 1;
  virtual ~hs_comm() {}
  // Enqueue on `s`: every piece of `sends` goes to its peer, every piece of `recvs` is filled by its peer.  Pieces between one
  // pair of ranks are matched in list order, so both sides must list them in the same order.  Stream-ordered for RCCL; the
  // host-staged form synchronises `s`, moves the data and returns when the received pieces are in device memory.
  virtual void transfer(const std::vector<HsPiece>& sends, const std::vector<HsPiece>& recvs, hipStream_t s) = 0;
  virtual const char* kind() const = 0;
};

hs_comm* hs_comm_make_rccl(const void* id128, int rank, int nranks);
hs_comm* hs_comm_make_host(hs_transfer_fn fn, void* user, int rank, int nranks);
void hs_comm_rccl_unique_id(void* id128);
