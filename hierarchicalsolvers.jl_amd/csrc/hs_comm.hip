// hs_comm.hip -- the two transports behind hs_comm::transfer (hs_comm.h) and the communicator part of the C ABI.
#include "hs_comm.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include "hs_common.h"

#define HS_COMM_FAIL(code, ...)         \
  do {                                  \
    hs_set_error(code, 0, __VA_ARGS__); \
    throw (int)(code);                  \
  } while (0)

// ------------------------------------------------------------------------------------------------
// RCCL over xGMI: grouped ncclSend / ncclRecv on one world communicator, stream-ordered
// ------------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl_api() {
  static RcclApi api;
  if (api.so) return api;
  // the SONAME first: a host process that already holds RCCL (PyTorch bundles its own copy) keeps a single instance
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* nm : names) {
    api.so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (api.so) break;
  }
  if (!api.so) HS_COMM_FAIL(HS_ERR_DEVICE, "librccl.so could not be loaded: %s", dlerror());
  auto sym = [&](const char* nm) {
    void* p = dlsym(api.so, nm);
    if (!p) HS_COMM_FAIL(HS_ERR_DEVICE, "librccl.so lacks %s", nm);
    return p;
  };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  return api;
}

#define HS_NCCL(call)                                                                                         \
  do {                                                                                                        \
    ncclResult_t r__ = (call);                                                                                \
    if (r__ != ncclSuccess) HS_COMM_FAIL(HS_ERR_DEVICE, "%s failed: %s", #call, rccl_api().GetErrorString(r__)); \
  } while (0)

struct RcclComm : hs_comm {
  ncclComm_t comm = nullptr;
  ~RcclComm() override {
    if (comm) (void)rccl_api().CommDestroy(comm);
  }
  const char* kind() const override { return "rccl"; }
  void transfer(const std::vector<HsPiece>& sends, const std::vector<HsPiece>& recvs, hipStream_t s) override {
    RcclApi& R = rccl_api();
    bool any = false;
    for (auto& p : sends) any = any || p.bytes > 0;
    for (auto& p : recvs) any = any || p.bytes > 0;
    if (!any) return;
    HS_NCCL(R.GroupStart());
    // a failing ncclSend / ncclRecv must not leave the communicator inside an open group (every later call on it would be queued into that
    // group and never run): the group is closed before the error is raised
    ncclResult_t bad = ncclSuccess;
    const char* what = "";
    for (auto& p : sends)
      if (p.bytes > 0 && bad == ncclSuccess) {
        bad = R.Send(p.ptr, p.bytes, ncclChar, p.peer, comm, s);
        what = "ncclSend";
      }
    for (auto& p : recvs)
      if (p.bytes > 0 && bad == ncclSuccess) {
        bad = R.Recv(p.ptr, p.bytes, ncclChar, p.peer, comm, s);
        what = "ncclRecv";
      }
    const ncclResult_t endr = R.GroupEnd();
    if (bad != ncclSuccess) HS_COMM_FAIL(HS_ERR_DEVICE, "%s failed inside a grouped transfer: %s", what, R.GetErrorString(bad));
    if (endr != ncclSuccess) HS_COMM_FAIL(HS_ERR_DEVICE, "ncclGroupEnd failed: %s", R.GetErrorString(endr));
  }
};

// ------------------------------------------------------------------------------------------------
// host-staged: one message per peer and direction, packed in pinned memory, moved by the host layer's callback
// ------------------------------------------------------------------------------------------------
struct HostComm : hs_comm {
  hs_transfer_fn fn = nullptr;
  void* user = nullptr;
  void* stage[2] = {nullptr, nullptr};
  size_t cap[2] = {0, 0};
  ~HostComm() override {
    for (void* p : stage)
      if (p) (void)hipHostFree(p);
  }
  const char* kind() const override { return "host"; }
  void reserve(int which, size_t bytes) {
    if (bytes <= cap[which]) return;
    if (stage[which]) (void)hipHostFree(stage[which]);
    stage[which] = nullptr;
    cap[which] = 0;
    HS_HIP(hipHostMalloc(&stage[which], bytes, hipHostMallocDefault));
    cap[which] = bytes;
  }
  void transfer(const std::vector<HsPiece>& sends, const std::vector<HsPiece>& recvs, hipStream_t s) override {
    // per-peer messages: the pieces of one peer back to back, in list order
    std::map<int, size_t> sbytes, rbytes;
    for (auto& p : sends) sbytes[p.peer] += p.bytes;
    for (auto& p : recvs) rbytes[p.peer] += p.bytes;
    size_t stot = 0, rtot = 0;
    std::map<int, size_t> soff, roff;
    for (auto& kv : sbytes) {
      soff[kv.first] = stot;
      stot += (kv.second + 63) / 64 * 64;
    }
    for (auto& kv : rbytes) {
      roff[kv.first] = rtot;
      rtot += (kv.second + 63) / 64 * 64;
    }
    if (stot + rtot == 0) return;
    reserve(0, std::max<size_t>(stot, 64));
    reserve(1, std::max<size_t>(rtot, 64));
    char* sb = (char*)stage[0];
    char* rb = (char*)stage[1];
    {
      std::map<int, size_t> cur = soff;
      for (auto& p : sends) {
        if (p.bytes == 0) continue;
        HS_HIP(hipMemcpyAsync(sb + cur[p.peer], p.ptr, p.bytes, hipMemcpyDeviceToHost, s));
        cur[p.peer] += p.bytes;
      }
    }
    HS_HIP(hipStreamSynchronize(s));
    std::vector<int64_t> speer, sbyte, rpeer, rbyte;
    std::vector<void*> sptr, rptr;
    for (auto& kv : sbytes)
      if (kv.second > 0) {
        speer.push_back(kv.first);
        sbyte.push_back((int64_t)kv.second);
        sptr.push_back(sb + soff[kv.first]);
      }
    for (auto& kv : rbytes)
      if (kv.second > 0) {
        rpeer.push_back(kv.first);
        rbyte.push_back((int64_t)kv.second);
        rptr.push_back(rb + roff[kv.first]);
      }
    int st = fn(user, (int64_t)speer.size(), speer.data(), sptr.data(), sbyte.data(), (int64_t)rpeer.size(), rpeer.data(), rptr.data(), rbyte.data());
    if (st != 0) HS_COMM_FAIL(HS_ERR_DEVICE, "the host layer's transfer callback returned %d", st);
    {
      std::map<int, size_t> cur = roff;
      for (auto& p : recvs) {
        if (p.bytes == 0) continue;
        HS_HIP(hipMemcpyAsync(p.ptr, rb + cur[p.peer], p.bytes, hipMemcpyHostToDevice, s));
        cur[p.peer] += p.bytes;
      }
    }
    HS_HIP(hipStreamSynchronize(s));  // the staging buffer is reused by the next call
  }
};
}  // namespace

void hs_comm_rccl_unique_id(void* id128) {
  ncclUniqueId id;
  HS_NCCL(rccl_api().GetUniqueId(&id));
  memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
}

hs_comm* hs_comm_make_rccl(const void* id128, int rank, int nranks) {
  RcclApi& R = rccl_api();
  ncclUniqueId id;
  memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
  RcclComm* c = new RcclComm();
  c->rank = rank;
  c->nranks = nranks;
  ncclResult_t r = R.CommInitRank(&c->comm, nranks, id, rank);
  if (r != ncclSuccess) {
    c->comm = nullptr;
    delete c;
    HS_COMM_FAIL(HS_ERR_DEVICE, "ncclCommInitRank(rank %d of %d) failed: %s", rank, nranks, R.GetErrorString(r));
  }
  return c;
}

hs_comm* hs_comm_make_host(hs_transfer_fn fn, void* user, int rank, int nranks) {
  HostComm* c = new HostComm();
  c->fn = fn;
  c->user = user;
  c->rank = rank;
  c->nranks = nranks;
  return c;
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------
#define HS_COMM_GUARD(...)                                   \
  try {                                                      \
    __VA_ARGS__;                                             \
    return HS_OK;                                            \
  } catch (int code) {                                       \
    return code;                                             \
  } catch (const std::bad_alloc&) {                          \
    hs_set_error(HS_ERR_NOMEM, 0, "host allocation failed"); \
    return HS_ERR_NOMEM;                                     \
  }

static void check_rank(int64_t rank, int64_t nranks) {
  if (nranks < 1 || rank < 0 || rank >= nranks) HS_COMM_FAIL(HS_ERR_ARGUMENT, "ArgumentError: rank %lld of %lld", (long long)rank, (long long)nranks);
}

extern "C" int hs_comm_unique_id(void* id128) {
  HS_COMM_GUARD(if (!id128) HS_COMM_FAIL(HS_ERR_ARGUMENT, "ArgumentError: id128 == NULL"); hs_comm_rccl_unique_id(id128));
}
extern "C" int hs_comm_create_rccl(const void* id128, int64_t rank, int64_t nranks, hs_comm** out) {
  HS_COMM_GUARD(if (!id128 || !out) HS_COMM_FAIL(HS_ERR_ARGUMENT, "ArgumentError: null argument"); *out = nullptr; check_rank(rank, nranks);
                *out = hs_comm_make_rccl(id128, (int)rank, (int)nranks));
}
extern "C" int hs_comm_create_host(hs_transfer_fn fn, void* user, int64_t rank, int64_t nranks, hs_comm** out) {
  HS_COMM_GUARD(if (!fn || !out) HS_COMM_FAIL(HS_ERR_ARGUMENT, "ArgumentError: null argument"); *out = nullptr; check_rank(rank, nranks);
                *out = hs_comm_make_host(fn, user, (int)rank, (int)nranks));
}
extern "C" void hs_comm_free(hs_comm* c) { delete c; }
extern "C" const char* hs_comm_kind(const hs_comm* c) { return c ? c->kind() : ""; }

// Self-test of a communicator: rank r sends `bytes` bytes of a pattern to (r+1) % nranks and receives from (r-1+nranks) % nranks
// (to itself when nranks == 1), then checks the pattern.  Returns HS_OK when the received bytes are the sender's.
__global__ void hs_comm_fill_kernel(unsigned char* p, size_t n, unsigned seed) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) p[i] = (unsigned char)((i * 2654435761u + seed) >> 7);
}
namespace {
// device buffers, a stream and events of the self-test / bandwidth probe, released on every exit path (a transfer or a HIP call that throws used
// to leak them: bench.py runs both on every multi-rank start and carries on after a failure)
struct ProbeRes {
  std::vector<void*> bufs;
  hipStream_t s = nullptr;
  std::vector<hipEvent_t> evs;
  ~ProbeRes() {
    if (s) (void)hipStreamSynchronize(s);
    for (hipEvent_t e : evs) (void)hipEventDestroy(e);
    for (void* p : bufs) (void)hipFree(p);
    if (s) (void)hipStreamDestroy(s);
  }
  unsigned char* buf(size_t bytes) {
    void* p = nullptr;
    HS_HIP(hipMalloc(&p, bytes));
    bufs.push_back(p);
    return (unsigned char*)p;
  }
  hipEvent_t event() {
    hipEvent_t e = nullptr;
    HS_HIP(hipEventCreate(&e));
    evs.push_back(e);
    return e;
  }
};
}  // namespace

extern "C" int hs_comm_selftest(hs_comm* c, int64_t bytes) {
  HS_COMM_GUARD(
      if (!c || bytes <= 0) HS_COMM_FAIL(HS_ERR_ARGUMENT, "ArgumentError: communicator / byte count");
      ProbeRes R; unsigned char* d = R.buf((size_t)bytes); unsigned char* r = R.buf((size_t)bytes); HS_HIP(hipStreamCreate(&R.s)); hipStream_t s = R.s;
      const int next = (c->rank + 1) % c->nranks, prev = (c->rank - 1 + c->nranks) % c->nranks;
      hs_comm_fill_kernel<<<(unsigned)((bytes + 255) / 256), 256, 0, s>>>(d, (size_t)bytes, 17u * (unsigned)c->rank + 1u);
      HS_HIP(hipMemsetAsync(r, 0, bytes, s));
      c->transfer({HsPiece{next, d, (size_t)bytes}}, {HsPiece{prev, r, (size_t)bytes}}, s);
      hs_comm_fill_kernel<<<(unsigned)((bytes + 255) / 256), 256, 0, s>>>(d, (size_t)bytes, 17u * (unsigned)prev + 1u);  // what prev sent
      std::vector<unsigned char> a(bytes), b(bytes); HS_HIP(hipMemcpyAsync(a.data(), d, bytes, hipMemcpyDeviceToHost, s));
      HS_HIP(hipMemcpyAsync(b.data(), r, bytes, hipMemcpyDeviceToHost, s)); HS_HIP(hipStreamSynchronize(s));
      if (memcmp(a.data(), b.data(), bytes) != 0) HS_COMM_FAIL(HS_ERR_DEVICE, "communicator self-test: received bytes differ from the sender's"));
}

// Point-to-point rate of the transport: `reps` ring shifts of `bytes` bytes (rank r -> r+1), device to device, timed with HIP events on
// the stream the transfers are enqueued on.  *gbps = bytes sent per rank / time (GB/s; one xGMI link per direction under RCCL).
extern "C" int hs_comm_bandwidth(hs_comm* c, int64_t bytes, int64_t reps, double* gbps) {
  HS_COMM_GUARD(
      if (!c || bytes <= 0 || reps <= 0 || !gbps) HS_COMM_FAIL(HS_ERR_ARGUMENT, "ArgumentError: communicator / sizes"); *gbps = 0.0;
      if (c->nranks < 2) return HS_OK;
      ProbeRes R; unsigned char* d = R.buf((size_t)bytes); unsigned char* r = R.buf((size_t)bytes); HS_HIP(hipStreamCreate(&R.s)); hipStream_t s = R.s;
      hipEvent_t e0 = R.event(); hipEvent_t e1 = R.event();
      HS_HIP(hipMemsetAsync(d, 1, bytes, s)); const int next = (c->rank + 1) % c->nranks, prev = (c->rank - 1 + c->nranks) % c->nranks;
      c->transfer({HsPiece{next, d, (size_t)bytes}}, {HsPiece{prev, r, (size_t)bytes}}, s);  // warm-up: connections are set up on first use
      HS_HIP(hipStreamSynchronize(s)); HS_HIP(hipEventRecord(e0, s));
      for (int64_t k = 0; k < reps; ++k) c->transfer({HsPiece{next, d, (size_t)bytes}}, {HsPiece{prev, r, (size_t)bytes}}, s);
      HS_HIP(hipEventRecord(e1, s)); HS_HIP(hipStreamSynchronize(s)); float ms = 0.f; HS_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (ms > 0.f) *gbps = (double)bytes * (double)reps / (ms * 1e-3) / 1e9);
}
