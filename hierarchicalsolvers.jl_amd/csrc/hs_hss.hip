// hs_hss.hip -- HSS matrices on the device: randomized compression, product, ULV-type elimination and solve.
//
// Role in the reference (C ABI and call sites: include/hs_hss.h): the HssMatrices.jl objects behind `S` and `D` of a
// compressed front -- `compress` / `randcompress_adaptive` (src/factorization.jl:56-57,109-110), `hssrank`
// (src/factornode.jl:53), HSS `*` (src/factorization.jl:242) and HSS `\` inside `blockfactor` / `blockldiv!`
// (src/blockmatrix.jl:121-156).  HssMatrices.jl is absent from the reference tree: the arithmetic follows the
// published algorithms (Martinsson 2011 for the randomized compression with interpolative nested bases; Martinsson &
// Rokhlin 2005 / Ho & Greengard 2012 for the ID-based elimination); the GPU tests compare against a CPU restatement of
// the same algorithms (tests/test_hss_gpu.py).  PARITY UNPINNED against the Julia package.
//
// Everything heavy runs through the kernels the fronts use, one grouped launch per tree level:
//   * samples A*Omega, Psi^T*A and every generator product: the MFMA GEMM on plain problem lists (launch_gemm_probs);
//   * the row interpolative decompositions of a level: lowrank_compress_batch (tournament-pivoted LU of the samples);
//   * the elimination of a level's redundant positions: the fronts' batched recursive LU (Sched::factor_fronts) on
//     fronts [R; S] -- LF = [X_RR; X_SR], UR = X_RS, SB = X_SS -> Schur complement on the skeleton;
//   * triangular solves with blocks of right-hand sides: the fronts' TRSM-by-inverse-blocks (laswp / trsm_rec / utrsm_rec).
// New kernels here are the HBM-bound movers: row gather / scatter, indexed sub-matrix gather (also the transposes),
// the interpolation matrices T = L21 * L11^-1 from the packed LU of the samples, index composition, Gaussian fill.
#include <exception>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <atomic>
#include <vector>

#include "../../include/hs_hss.h"
#include "hs_lowrank.h"
#include "hs_sched.h"

namespace {

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__device__ inline uint64_t hmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// rows x cols block of unit Gaussians (complex: both parts), leading dimension ld (in doubles: ld * sizeof(T)/8)
__global__ __launch_bounds__(256) void hss_randn_kernel(double* out, int rows_d, int ld_d, int cols, uint64_t seed) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows_d * cols) return;
  const int r = (int)(i % rows_d), c = (int)(i / rows_d);
  const uint64_t a = hmix64(seed ^ (i * 0xD1342543DE82EF95ull)), b = hmix64(a);
  const double u1 = ((a >> 11) + 1.0) * (1.0 / 9007199254740993.0);
  const double u2 = (b >> 11) * (1.0 / 9007199254740992.0);
  out[(size_t)r + (size_t)c * ld_d] = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

__device__ inline double conj_of(double a) { return a; }
__device__ inline cplx conj_of(cplx a) { return {a.re, -a.im}; }

enum { ROW_GATHER = 0, ROW_GATHER_NEG = 1, ROW_SCATTER = 2, ROW_SCATTER_ADD = 3 };
template <class T>
struct RowJob {
  const T* src;
  int lds;
  T* dst;
  int ldd;
  const int* idx;  // rows entries (null: identity)
  int rows, cols, mode;
};
// gather: dst[i, c] = (+-) src[idx[i], c]      scatter: dst[idx[i], c] (+)= src[i, c]
template <class T>
__global__ __launch_bounds__(64) void row_move_kernel(const RowJob<T>* __restrict__ jobs) {
  const RowJob<T> j = jobs[blockIdx.z];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= j.rows) return;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, j.cols);
  const int k = j.idx ? j.idx[i] : i;
  for (int c = c0; c < c1; ++c) {
    if (j.mode == ROW_GATHER)
      j.dst[(size_t)i + (size_t)c * j.ldd] = j.src[(size_t)k + (size_t)c * j.lds];
    else if (j.mode == ROW_GATHER_NEG)
      j.dst[(size_t)i + (size_t)c * j.ldd] = -j.src[(size_t)k + (size_t)c * j.lds];
    else if (j.mode == ROW_SCATTER)
      j.dst[(size_t)k + (size_t)c * j.ldd] = j.src[(size_t)i + (size_t)c * j.lds];
    else
      j.dst[(size_t)k + (size_t)c * j.ldd] = j.dst[(size_t)k + (size_t)c * j.ldd] + j.src[(size_t)i + (size_t)c * j.lds];
  }
}

template <class T>
struct SubJob {
  const T* A;
  int lda;
  const int* ri;  // row i of the result reads row r0 + (ri ? ri[i] : i) of A
  const int* ci;
  int r0, c0, rows, cols;
  T* out;
  int ldo, trans;  // trans: the result is written transposed (out is cols x rows)
  const int* hri = nullptr;  // HOST copies of ri / ci (+ r0 / c0 already added: absolute indices), for operators that route the
  const int* hci = nullptr;  // request on the host (BlockOp::gather); null: the range r0 + i
  int blk = 0;               // batched compression: which of the matrices A belongs to (its low-rank update corrects the block)
};
template <class T>
__global__ __launch_bounds__(64) void sub_gather_kernel(const SubJob<T>* __restrict__ jobs) {
  const SubJob<T> j = jobs[blockIdx.z];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= j.rows) return;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, j.cols);
  const size_t ar = (size_t)j.r0 + (j.ri ? j.ri[i] : i);
  for (int c = c0; c < c1; ++c) {
    const size_t ac = (size_t)j.c0 + (j.ci ? j.ci[c] : c);
    T v = j.A[ar + ac * j.lda];
    if (j.trans == 2) v = conj_of(v);  // conjugate transpose (the orthonormal bases of the QR refinement)
    if (j.trans)
      j.out[(size_t)c + (size_t)i * j.ldo] = v;
    else
      j.out[(size_t)i + (size_t)c * j.ldo] = v;
  }
}

// interpolation matrix of a row ID from the packed L\U of its pivoted samples: T = L21 * L11^-1 ((m-r) x r), L11 unit lower
// triangular.  Blocked by 32 columns from the right (compress_fixed): the update with the finished columns is an MFMA GEMM,
// this kernel finishes one block -- one thread per row, back substitution inside the block (a one-kernel version with the
// whole substitution per thread was 25 % of all device time of a factorization with HSS interior blocks).
template <class T>
struct TsJob {
  const T* Lp;
  int ldp, m, r;
  T* Tm;
  int ldt;
  int j0, j1;  // columns [j0, j1) of T
};
template <class T>
__global__ __launch_bounds__(64) void tsolve_block_kernel(const TsJob<T>* __restrict__ jobs) {
  const TsJob<T> j = jobs[blockIdx.y];
  const int row = blockIdx.x * 64 + threadIdx.x;
  if (row >= j.m - j.r) return;
  for (int c = j.j1 - 1; c >= j.j0; --c) {
    T acc = j.Tm[(size_t)row + (size_t)c * j.ldt];
    for (int q = c + 1; q < j.j1; ++q) acc = Scal<T>::fnma(j.Tm[(size_t)row + (size_t)q * j.ldt], j.Lp[(size_t)q + (size_t)c * j.ldp], acc);
    j.Tm[(size_t)row + (size_t)c * j.ldt] = acc;
  }
}

// rows of a leaf's basis U = P^T [I; T] at given local positions, as E (cnt x r; trans = 0) or as E^T (r x cnt; trans = 1):
// ip[a] = index of position a in the node's order p (skeleton positions first): < r -> unit row, else row ip - r of T
template <class T>
struct BasisJob {
  const T* Tm;
  int ldt, r, cnt;
  const int* ip;
  T* out;
  int ldo, trans;
};
template <class T>
__global__ __launch_bounds__(64) void basis_rows_kernel(const BasisJob<T>* __restrict__ jobs) {
  const BasisJob<T> j = jobs[blockIdx.z];
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= j.cnt) return;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, j.r);
  const int q = j.ip[a];
  for (int c = c0; c < c1; ++c) {
    const T v = q < j.r ? (q == c ? Scal<T>::one() : Scal<T>::zero()) : j.Tm[(size_t)(q - j.r) + (size_t)c * j.ldt];
    if (j.trans)
      j.out[(size_t)c + (size_t)a * j.ldo] = v;
    else
      j.out[(size_t)a + (size_t)c * j.ldo] = v;
  }
}

struct IdxJob {
  const int* p;     // local positions
  const int* base;  // global index of every local position (null: lo + position)
  int lo, cnt;
  int* out;
};
__global__ __launch_bounds__(64) void idx_compose_kernel(const IdxJob* __restrict__ jobs) {
  const IdxJob j = jobs[blockIdx.y];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= j.cnt) return;
  const int q = j.p[i];
  j.out[i] = j.base ? j.base[q] : j.lo + q;
}

// ------------------------------------------------------------------------------------------------
// host structures
// ------------------------------------------------------------------------------------------------
// Device blocks that the module allocates over and over (hundreds of small blocks per sweep, thousands per compression: every
// hipMalloc / hipFree costs 0.1-0.3 ms and a hipFree synchronises the device) are recycled through ONE process-wide cache: size
// classes of powers of two, blocks above 64 MiB go straight back to the driver.  The cache is never destroyed (its blocks return to
// the driver with the process; a static destructor would run after the HIP runtime has shut down) and is guarded by a mutex:
// distinct handles may be used from different threads.
extern "C" int64_t hs_arena_trim(void);
struct BlockCache {
  std::multimap<size_t, void*> free_;
  std::mutex mu;
  size_t held = 0;
  static constexpr size_t limit = (size_t)24 << 30;  // bytes the cache may hold back; beyond it freed blocks go to the driver
};
static BlockCache* global_cache() {
  static BlockCache* g = new BlockCache();
  return g;
}
// Pinned host staging blocks for descriptor uploads: a copy from pinned memory is asynchronous (ordered on the stream, no host
// synchronisation), one from a pageable std::vector is not -- and the module uploads a few small descriptor lists per tree level.
struct PinnedCache {
  std::multimap<size_t, void*> free_;
  std::mutex mu;
};
static PinnedCache* pinned_cache() {
  static PinnedCache* g = new PinnedCache();
  return g;
}
struct Pool {
  std::vector<std::pair<void*, size_t>> v;
  std::vector<std::pair<void*, size_t>> hv;  // pinned host blocks (valid until clear(): the copies they feed have been waited for by then)
  BlockCache* cache = nullptr;
  void* get_pinned(size_t bytes) {
    size_t cls = 4096;
    while (cls < bytes) cls <<= 1;
    PinnedCache* pc = pinned_cache();
    {
      std::lock_guard<std::mutex> lk(pc->mu);
      auto it = pc->free_.find(cls);
      if (it != pc->free_.end()) {
        void* p = it->second;
        pc->free_.erase(it);
        hv.push_back({p, cls});
        return p;
      }
    }
    void* p = nullptr;
    if (hipHostMalloc(&p, cls, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      hs_set_error(HS_ERR_NOMEM, 0, "hipHostMalloc of %zu bytes failed (HSS module)", cls);
      throw (int)HS_ERR_NOMEM;
    }
    hv.push_back({p, cls});
    return p;
  }
  Pool() = default;
  explicit Pool(BlockCache* c) : cache(c) {}
  Pool(const Pool&) = delete;
  Pool& operator=(const Pool&) = delete;
  // the stream the pool's blocks are used on (optional).  Every user synchronises it before its pool dies on the NORMAL path; when an
  // exception unwinds past the pool, work may still be in flight on blocks that are about to be recycled -- possibly to the other
  // compression thread (mf_parallel): wait for the stream first
  hipStream_t guard = nullptr;
  ~Pool() {
    if (guard && std::uncaught_exceptions() > 0) (void)hipStreamSynchronize(guard);
    clear();
  }
  void clear() {
    if (!hv.empty()) {
      PinnedCache* pc = pinned_cache();
      std::lock_guard<std::mutex> lk(pc->mu);
      for (auto& pr : hv) pc->free_.insert({pr.second, pr.first});
      hv.clear();
    }
    if (cache) {
      std::lock_guard<std::mutex> lk(cache->mu);
      for (auto& pr : v) {
        if (pr.second <= ((size_t)64 << 20) + 4096 && cache->held + pr.second <= BlockCache::limit) {
          cache->free_.insert({pr.second, pr.first});
          cache->held += pr.second;
        } else {
          (void)hipFree(pr.first);
        }
      }
    } else {
      for (auto& pr : v) (void)hipFree(pr.first);
    }
    v.clear();
    zcur = nullptr;
    zleft = 0;
  }
  // Zero-filled scratch carved out of chunks that are cleared with ONE memset when they are taken: the per-node buffers of a grouped
  // stage (hundreds per tree level) used to cost a memset launch each -- 64,000 of the 172,000 launches of a matrix-free factorization of
  // Poisson 128^3.  Everything a Pool hands out is used on one stream, after the chunk's memset in stream order.
  char* zcur = nullptr;
  size_t zleft = 0;
  template <class U>
  U* getz(size_t count, hipStream_t s) {
    const size_t bytes = (count * sizeof(U) + 511) / 256 * 256;
    if (bytes > zleft) {
      const size_t chunk = std::max(bytes, (size_t)16 << 20);
      zcur = (char*)get<char>(chunk);
      zleft = chunk;
      if (hipMemsetAsync(zcur, 0, chunk, s) != hipSuccess) {
        hs_set_error(HS_ERR_DEVICE, 0, "hipMemsetAsync failed (HSS module)");
        throw (int)HS_ERR_DEVICE;
      }
    }
    U* p = (U*)zcur;
    zcur += bytes;
    zleft -= bytes;
    return p;
  }
  template <class U>
  U* get(size_t count) {
    size_t bytes = count * sizeof(U) + 512;
    if (cache && bytes <= ((size_t)64 << 20)) {  // larger blocks are allocated exactly and go straight back to the driver
      size_t cls = 1024;
      while (cls < bytes) cls <<= 1;
      if (cls > 4096 && cls - cls / 4 >= bytes) cls -= cls / 4;  // a 3/4 class in between: at most 1/3 of a block is slack
      bytes = cls;
      std::lock_guard<std::mutex> lk(cache->mu);
      auto it = cache->free_.find(bytes);
      if (it != cache->free_.end()) {
        void* p = it->second;
        cache->free_.erase(it);
        cache->held -= bytes;
        v.push_back({p, bytes});
        return (U*)p;
      }
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      if (cache) {  // give the cached blocks back to the driver and try once more
        std::lock_guard<std::mutex> lk(cache->mu);
        for (auto& kv : cache->free_) (void)hipFree(kv.second);
        cache->free_.clear();
        cache->held = 0;
      }
      if (hipMalloc(&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        (void)hs_arena_trim();  // the arenas hs_free parked (hs_api.hip): up to a whole factorization's worth of memory
        if (hipMalloc(&p, bytes) != hipSuccess) {
          (void)hipGetLastError();
          hs_set_error(HS_ERR_NOMEM, 0, "hipMalloc of %zu bytes failed (HSS module)", bytes);
          throw (int)HS_ERR_NOMEM;
        }
      }
    }
    v.push_back({p, bytes});
    return (U*)p;
  }
};

#define HSS_HIP(call)                                                                             \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      hs_set_error(HS_ERR_DEVICE, 0, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      throw (int)HS_ERR_DEVICE;                                                                   \
    }                                                                                             \
  } while (0)

inline int ev(int x) { return std::max(2, (x + 1) / 2 * 2); }

template <class T>
struct HNode {
  int lo = 0, hi = 0, left = -1, right = -1, parent = -1, level = 0;
  int m = 0, r = 0;
  int* p = nullptr;   // m local positions, skeleton first
  int* sk = nullptr;  // r global indices
  T* Tm = nullptr;    // (m-r) x r
  T* Tt = nullptr;    // r x (m-r)
  int ldt = 2, ldtt = 2;
  T* D = nullptr;     // leaf: m x m
  int ldd = 2;
  T *B12 = nullptr, *B21 = nullptr;  // inner node: r_l x r_r, r_r x r_l
  int ld12 = 2, ld21 = 2;
  // elimination: the front [R; S]
  bool has_front = false;
  NodeDesc<T> fd;
  int off_in_parent = 0;  // row offset of this node's skeleton inside the parent's local vectors
  std::vector<int> hinvp;  // host: inverse of p (entry access, filled on first use)
  std::vector<int> hsk;    // host copy of sk (kept by the matrix-free compression, which routes entry requests on the host)
  T* NTm = nullptr;        // -T (entry access)
  T *DTt = nullptr, *B12t = nullptr, *B21t = nullptr;  // D^T, B12^T (rr x rl), B21^T (rl x rr): transposed product, made on first use
};

template <class T>
struct HssT {
  int n = 0, k = 0, nlev = 0;
  hs_hss_options opt;
  std::vector<HNode<T>> nd;
  std::vector<std::vector<int>> lev;
  Pool keep{global_cache()};  // generators and factors (blocks recycle through the process-wide cache)
  hipStream_t s = nullptr;
  bool own_stream = false;
  int* perm = nullptr;  // device, n entries (0-based) or null: H ~= A[perm, perm]
  std::vector<int> hinvperm;  // host: position of every caller index in the tree's order (empty: identity)
  std::vector<int> hperm;     // host copy of perm (empty: identity)
  Pool permpool{global_cache()};
  bool factored = false;
  NodeDesc<T> rootfd;  // LU of the last block
  int root_m = 0;
  double t_compress = 0.0, t_factor = 0.0;
  std::shared_ptr<void> hold;  // a matrix of a batched compression: the forest that owns its generators lives as long as any of them
  ~HssT() {
    if (s && own_stream) (void)hipStreamDestroy(s);
  }
};

// asynchronous on `s` through a pinned staging block that lives as long as `pool` does (every user synchronises `s` before its pool dies)
template <class J>
J* upload(Pool& pool, const std::vector<J>& v, hipStream_t s) {
  J* d = pool.get<J>(std::max<size_t>(v.size(), 1));
  if (v.empty()) return d;
  const size_t bytes = sizeof(J) * v.size();
  void* h = pool.get_pinned(bytes);
  memcpy(h, v.data(), bytes);
  HSS_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s));
  return d;
}
template <class J>
J* upload(Pool& pool, const std::vector<J>& v) {
  J* d = pool.get<J>(std::max<size_t>(v.size(), 1));
  if (!v.empty()) HSS_HIP(hipMemcpy(d, v.data(), sizeof(J) * v.size(), hipMemcpyHostToDevice));
  return d;
}

template <class T>
void run_rows(Pool& tmp, std::vector<RowJob<T>>& jobs, hipStream_t s) {
  std::vector<RowJob<T>> live;
  int mr = 0, mc = 0;
  for (auto& j : jobs)
    if (j.rows > 0 && j.cols > 0) {
      live.push_back(j);
      mr = std::max(mr, j.rows);
      mc = std::max(mc, j.cols);
    }
  jobs.clear();
  if (live.empty()) return;
  RowJob<T>* d = upload(tmp, live, s);
  for (size_t b = 0; b < live.size(); b += 32768) {
    const unsigned cnt = (unsigned)std::min<size_t>(32768, live.size() - b);
    hipLaunchKernelGGL(row_move_kernel<T>, dim3((mr + 63) / 64, (mc + 15) / 16, cnt), dim3(64), 0, s, (const RowJob<T>*)(d + b));
  }
}
template <class T>
void run_subs(Pool& tmp, std::vector<SubJob<T>>& jobs, hipStream_t s) {
  std::vector<SubJob<T>> live;
  int mr = 0, mc = 0;
  for (auto& j : jobs)
    if (j.rows > 0 && j.cols > 0) {
      live.push_back(j);
      mr = std::max(mr, j.rows);
      mc = std::max(mc, j.cols);
    }
  jobs.clear();
  if (live.empty()) return;
  SubJob<T>* d = upload(tmp, live, s);
  for (size_t b = 0; b < live.size(); b += 32768) {
    const unsigned cnt = (unsigned)std::min<size_t>(32768, live.size() - b);
    hipLaunchKernelGGL(sub_gather_kernel<T>, dim3((mr + 63) / 64, (mc + 15) / 16, cnt), dim3(64), 0, s, (const SubJob<T>*)(d + b));
  }
}
// C (+/-)= sum of `parts` partial products stored one after the other (stride `pstride` elements), column by column
template <class T>
struct SplitKJob {
  const T* part;
  size_t pstride;
  int parts, rows, cols, ldp;
  T* C;
  int ldc, minus;
};
template <class T>
__global__ __launch_bounds__(64) void splitk_reduce_kernel(const SplitKJob<T>* __restrict__ jobs) {
  const SplitKJob<T> j = jobs[blockIdx.z];
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= j.rows) return;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, j.cols);
  for (int c = c0; c < c1; ++c) {
    T acc = Scal<T>::zero();
    for (int p = 0; p < j.parts; ++p) acc = acc + j.part[(size_t)p * j.pstride + (size_t)i + (size_t)c * j.ldp];
    T* dst = j.C + (size_t)i + (size_t)c * j.ldc;
    *dst = j.minus ? (*dst - acc) : acc;
  }
}

// C = A*B (minus = 0; C must not alias) or C -= A*B (minus = 1); problems with an empty dimension are dropped
// (the callers zero-fill the results of C = A*B beforehand, so K = 0 leaves zeros).
// Skinny products with a long inner dimension (a 64-row window against 8,192 sample columns: ONE 128 x 128 tile walking all of K while
// the chip idles) are split along K into up to 16 partial products that run as separate tiles, then summed by a small kernel.
template <class T>
void run_gemms(Pool& tmp, std::vector<GemmProb<T>>& probs, int minus, hipStream_t s) {
  std::vector<GemmProb<T>> live;
  std::vector<SplitKJob<T>> red;
  int mM = 0, mN = 0;
  static const bool splitk = !(getenv("HS_SPLITK") && getenv("HS_SPLITK")[0] == '0');
  size_t tiles_total = 0;
  for (auto& p : probs)
    if (p.M > 0 && p.N > 0 && p.K > 0) tiles_total += (size_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
  for (auto& p : probs)
    if (p.M > 0 && p.N > 0 && p.K > 0) {
      const size_t tiles = (size_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
      if (splitk && tiles_total < 256 && p.K >= 2048) {
        int parts = (int)std::min<size_t>(16, std::max<size_t>(2, 512 / std::max<size_t>(tiles_total, 1)));
        parts = std::min(parts, p.K / 512);
        if (parts >= 2) {
          const int ldp = ev(p.M);
          const size_t pstride = (size_t)ldp * p.N;
          T* part = tmp.get<T>(pstride * parts);
          const int kc = ((p.K + parts - 1) / parts + 15) / 16 * 16;
          int used = 0;
          for (int q = 0; q * kc < p.K; ++q, ++used) {
            const int k0 = q * kc, kw = std::min(kc, p.K - k0);
            live.push_back(GemmProb<T>{p.A + (size_t)k0 * p.lda, p.B + k0, part + (size_t)q * pstride, p.M, p.N, kw, p.lda, p.ldb, ldp});
          }
          red.push_back(SplitKJob<T>{part, pstride, used, p.M, p.N, ldp, p.C, p.ldc, minus});
          mM = std::max(mM, p.M);
          mN = std::max(mN, p.N);
          (void)tiles;
          continue;
        }
      }
      live.push_back(p);
      mM = std::max(mM, p.M);
      mN = std::max(mN, p.N);
    }
  probs.clear();
  if (live.empty()) return;
  if (!red.empty()) {
    // the partial products are plain C = A*B; the direct problems keep the caller's mode: two launches when both kinds are present
    std::vector<GemmProb<T>> direct, parts;
    for (auto& p : live) {
      bool is_part = false;
      for (auto& r : red)
        if (p.C >= r.part && p.C < r.part + r.pstride * r.parts) is_part = true;
      (is_part ? parts : direct).push_back(p);
    }
    GemmProb<T>* dp = upload(tmp, parts, s);
    for (size_t b = 0; b < parts.size(); b += 32768) launch_gemm_probs<T>(dp + b, (int)std::min<size_t>(32768, parts.size() - b), mM, mN, 0, s);
    if (!direct.empty()) {
      GemmProb<T>* dd = upload(tmp, direct, s);
      for (size_t b = 0; b < direct.size(); b += 32768) launch_gemm_probs<T>(dd + b, (int)std::min<size_t>(32768, direct.size() - b), mM, mN, minus, s);
    }
    SplitKJob<T>* dr = upload(tmp, red, s);
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((mM + 63) / 64, (mN + 15) / 16, (unsigned)red.size()), dim3(64), 0, s, (const SplitKJob<T>*)dr);
    return;
  }
  GemmProb<T>* d = upload(tmp, live, s);
  static const bool glog = getenv("HS_HSS_GEMM_LOG") != nullptr;  // diagnostics: wall time of every grouped product (adds synchronisations)
  std::chrono::steady_clock::time_point t0;
  if (glog) {
    (void)hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
  }
  for (size_t b = 0; b < live.size(); b += 32768) {
    const int cnt = (int)std::min<size_t>(32768, live.size() - b);
    launch_gemm_probs<T>(d + b, cnt, mM, mN, minus, s);
  }
  if (glog) {
    (void)hipStreamSynchronize(s);
    int mK = 0;
    double fl = 0;
    for (auto& q : live) {
      mK = std::max(mK, q.K);
      fl += 2.0 * q.M * q.N * q.K;
    }
    fprintf(stderr, "[hs gemm] %zu problems maxM %d maxN %d maxK %d flops %.3g  %.1f us\n", live.size(), mM, mN, mK, fl, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
  }
}

template <class T>
void fill_randn(T* out, int rows, int ld, int cols, uint64_t seed, hipStream_t s) {
  const int f = (int)(sizeof(T) / 8);
  const size_t total = (size_t)rows * f * cols;
  if (total == 0) return;
  hipLaunchKernelGGL(hss_randn_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (double*)out, rows * f, ld * f, cols, seed);
}

// Largest row 2-norm of a block of samples: the scale of what the products summed up BEFORE a node's own block is subtracted from its rows --
// round-off of that scale is what the residual norms of the orthogonalisation bottom out at, whatever tolerance is asked for.
template <class T>
__global__ __launch_bounds__(256) void rownorm_max_kernel(const T* __restrict__ Y, int ld, int rows, int cols, unsigned long long* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  double acc = 0.0;
  if (i < rows)
    for (int c = 0; c < cols; ++c) {
      const double a = Scal<T>::abs1(Y[(size_t)i + (size_t)c * ld]);
      acc += a * a;
    }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax(out, (unsigned long long)__double_as_longlong(sqrt(red[0])));  // non-negative doubles order like their bit patterns
}
// HS_NOISE_REL * (that scale): candidates whose residual norm is below it are round-off, not rank (abs1 of a complex entry is |re| + |im|: within sqrt(2))
#define HS_NOISE_REL 1e-13
static double noise_rel() {
  static const double v = getenv("HS_NOISE_REL") ? atof(getenv("HS_NOISE_REL")) : HS_NOISE_REL;  // diagnostics
  return v;
}

template <class T>
void build_tree(HssT<T>& H, int n, int leafsize, int first_split) {
  H.nd.clear();
  H.nd.emplace_back();
  H.nd[0].lo = 0;
  H.nd[0].hi = n;
  std::vector<int> q{0};
  for (size_t qi = 0; qi < q.size(); ++qi) {
    const int i = q[qi];
    const int lo = H.nd[i].lo, hi = H.nd[i].hi, sz = hi - lo, lv = H.nd[i].level;
    const bool forced = i == 0 && first_split > 0 && first_split < n;
    if (sz <= leafsize && !forced) continue;
    const int mid = forced ? first_split : lo + (sz + 1) / 2;
    HNode<T> a, b;
    a.lo = lo; a.hi = mid; a.level = lv + 1; a.parent = i;
    b.lo = mid; b.hi = hi; b.level = lv + 1; b.parent = i;
    const int li = (int)H.nd.size();
    H.nd.push_back(a);
    H.nd.push_back(b);
    H.nd[i].left = li;
    H.nd[i].right = li + 1;
    q.push_back(li);
    q.push_back(li + 1);
  }
  H.nlev = 0;
  for (auto& x : H.nd) H.nlev = std::max(H.nlev, x.level + 1);
  H.lev.assign(H.nlev, {});
  for (int i = 0; i < (int)H.nd.size(); ++i) H.lev[H.nd[i].level].push_back(i);
}

// Optional low-rank update of the matrix being compressed: A = B - C*M*Z with thin C (n x r1), small M (r1 x r2), thin Z (r2 x n).
// The Schur complement of a compressed front is such an operator -- `S = Abb - Abi*R` with a low-rank `R`, seen through products
// (`_sample_schur!`, src/factorization.jl:238-244) and entries (`_getindex_schur`, :246-249) -- and is compressed WITHOUT being formed:
// the samples take three thin GEMMs more, every gathered block B[I, J] is corrected by (C[I, :]*M)*Z[:, J].
template <class T>
struct Lru {
  const T* C = nullptr;
  const T* M = nullptr;
  const T* Z = nullptr;
  int ldc = 0, ldm = 0, ldz = 0, r1 = 0, r2 = 0;
  bool on() const { return C && Z && r1 > 0 && r2 > 0; }  // M == nullptr: the identity (r1 == r2)
};

// ------------------------------------------------------------------------------------------------
// compression with k samples per side; returns false when some rank came too close to k (caller doubles k)
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Rank and interpolation matrix of a row ID from an orthogonalisation of the rows in PIVOT ORDER ("QR in the order a tournament-pivoted
// LU chose": the pivot rows of the LU are as good a skeleton as a column-pivoted QR's, but |u_jj| overestimates the residual norm
// |R_jj| by the growth of the elimination and T = L21*L11^-1 interpolates a few sketch columns exactly instead of all of them in the
// least-squares sense -- measured: ranks 1.4-2x a pivoted QR's, and at the upper levels of a nested compression the children's
// truncation noise, sitting just below the threshold, was counted as rank).
//
// Block classical Gram-Schmidt with re-orthogonalisation over blocks of 32 rows, every block finished by a Cholesky-QR of the 32
// residual rows (one wave per block): M[p, :] = L * Q with orthonormal rows of Q; d_j = L_jj is the distance of row j from the span of
// the rows before it -- the |R_jj| of the pivoted QR of M^H in that order.  Rank = rows with d_j > tau; the remaining rows are
// interpolated in the least-squares sense: M[p_R, :] ~= (M[p_R, :] Q_S^H) Q_S = T * M[p_S, :] with T = L_RS * L_SS^-1.
// Everything heavy is a grouped MFMA GEMM over the blocks of a tree level; the role of `pqrfact` (rank-revealing QR, tolerance-stopped)
// inside the HSS compression of the reference (src/factorization.jl:110).
// ------------------------------------------------------------------------------------------------
template <class T>
struct CholJob {
  T* G;          // b x b Gram matrix of the candidate rows (ld 32), destroyed
  T* Lout;       // -> L[done : done+b', done : done+b'] (ld ldl)
  int ldl;
  T* Linv;       // b' x b': inverse of the accepted Cholesky factor (ld 32) -- the triangular solves of the interpolation matrix
  T* LinvP;      // b' x b : the same times the block's pivoting, Q_new = LinvP * W with W in candidate order (ld 32)
  double* d;     // accepted diagonal entries (b' of them)
  double* top;   // in/out: d_0 of the job (written by the first block)
  int* p;        // the job's candidate order at this window (b entries), permuted in place: accepted rows first
  int* lperm;    // out: local order (b entries): new position i holds old candidate lperm[i]
  int* nacc;     // out: b'
  double atol, rtol, scale_floor;
  int b, first;
  double floor_abs;  // nothing below this is accepted: the round-off level of the block's samples (0: none)
  double cond, noise_rel;  // HS_CHOL_COND, HS_NOISE_REL (run-time copies: diagnostics may override them)
  const double* next2 = nullptr;  // QrJob::norm_select: squared residual norm of the largest row OUTSIDE the window (an upper bound: it is
  double theta2 = 0.0;            // downdated afterwards); a pivot below theta * that is not accepted -- the greedy order of a pivoted QR, relaxed by theta
};
template <class T>
__device__ inline double real_of(T a);
template <>
__device__ inline double real_of<double>(double a) { return a; }
template <>
__device__ inline double real_of<cplx>(cplx a) { return a.re; }
template <class T>
__device__ inline T from_real(double a);
template <>
__device__ inline double from_real<double>(double a) { return a; }
template <>
__device__ inline cplx from_real<cplx>(double a) { return {a, 0.0}; }

// Diagonally pivoted Cholesky of the Gram matrix of <= 32 candidate rows (one wave): the candidates are re-ordered by residual norm,
// and only a well-conditioned prefix is ACCEPTED -- d_k above the truncation threshold and above 1e-5 of the block's first pivot (what a
// Gram matrix resolves reliably in double precision is d_k / |w_k| >~ 1e-8).  Rejected candidates stay next in line: the following
// window orthogonalises them against the rows accepted here before judging them again.
#define HS_CHOL_COND 1e-2
#define HS_QW 64  // candidates per window (rows orthogonalised per step): one wave factors their HS_QW x HS_QW Gram matrix in LDS
template <class T>
__global__ __launch_bounds__(64) void chol_block_kernel(const CholJob<T>* __restrict__ jobs) {
  const CholJob<T> j = jobs[blockIdx.x];
  constexpr int W = HS_QW, LD = HS_QW + 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char chol_smem[];
  T* g = reinterpret_cast<T*>(chol_smem);  // W x LD
  T* li = g + W * LD;                      // W x LD
  int* perm = reinterpret_cast<int*>(li + W * LD);
  int* oldp = perm + W;
  __shared__ int s_piv;
  __shared__ double s_dk;
  __shared__ int s_nacc;
  const int t = threadIdx.x, b = j.b;
  for (int c = 0; c < W; ++c) g[t * LD + c] = (t < b && c < b) ? j.G[(size_t)t + (size_t)c * W] : Scal<T>::zero();
  perm[t] = t;
  if (t == 0) s_nacc = 0;
  __syncthreads();
  double top = j.first ? 0.0 : *j.top, d0 = 0.0, tau = 0.0;
  const double out2 = j.next2 ? j.theta2 * *j.next2 : 0.0;
  for (int k = 0; k < b; ++k) {
    if (t == 0) {  // largest remaining diagonal entry
      int best = k;
      double bv = real_of(g[k * LD + k]);
      for (int i = k + 1; i < b; ++i) {
        const double v = real_of(g[i * LD + i]);
        if (v > bv) { bv = v; best = i; }
      }
      s_piv = best;
      s_dk = bv > 0.0 ? sqrt(bv) : 0.0;
    }
    __syncthreads();
    const int pv = s_piv;
    const double dk = s_dk;
    if (k == 0) {
      d0 = dk;
      if (j.first) top = dk;
      tau = fmax(fmax(j.atol, j.rtol * fmax(top, j.scale_floor)), fmax(j.floor_abs, j.noise_rel * top));
    }
    if (!(dk > tau) || !(dk > j.cond * d0) || !(dk > 0.0)) break;  // uniform: every thread sees the same dk
    if (k > 0 && dk * dk < out2) break;  // a row outside the window is (possibly) further from the span: it goes first
    // symmetric swap k <-> pv of the full square copy (both triangles are maintained) and of the finished columns
    if (pv != k && t < b) {
      if (t == 0) { const int q = perm[k]; perm[k] = perm[pv]; perm[pv] = q; }
      const T a = g[t * LD + k], c = g[t * LD + pv];
      g[t * LD + k] = c;
      g[t * LD + pv] = a;
    }
    __syncthreads();
    if (pv != k && t < b) {
      const T a = g[k * LD + t], c = g[pv * LD + t];
      g[k * LD + t] = c;
      g[pv * LD + t] = a;
    }
    __syncthreads();
    if (t < b && t >= k) g[t * LD + k] = t == k ? from_real<T>(dk) : g[t * LD + k] / from_real<T>(dk);
    __syncthreads();
    if (t < b && t > k)
      for (int c = k + 1; c < b; ++c) g[t * LD + c] = Scal<T>::fnma(g[t * LD + k], conj_of(g[c * LD + k]), g[t * LD + c]);  // Hermitian trailing update
    if (t == 0) s_nacc = k + 1;
    __syncthreads();
  }
  __syncthreads();
  const int na = s_nacc;
  // inverse of the accepted lower-triangular factor: thread c solves L x = e_c
  for (int i = 0; i < W; ++i) li[i * LD + t] = Scal<T>::zero();
  __syncthreads();
  if (t < na) {
    li[t * LD + t] = Scal<T>::one() / g[t * LD + t];
    for (int i = t + 1; i < na; ++i) {
      T acc = Scal<T>::zero();
      for (int k = t; k < i; ++k) acc = Scal<T>::fma(g[i * LD + k], li[k * LD + t], acc);
      li[i * LD + t] = (Scal<T>::zero() - acc) / g[i * LD + i];
    }
  }
  __syncthreads();
  for (int c = 0; c < W; ++c) {
    const bool in = t < na && c < na;
    if (in) j.Lout[(size_t)t + (size_t)c * j.ldl] = c <= t ? g[t * LD + c] : Scal<T>::zero();
    j.Linv[(size_t)t + (size_t)c * W] = in ? li[t * LD + c] : Scal<T>::zero();
    j.LinvP[(size_t)t + (size_t)c * W] = Scal<T>::zero();
  }
  __syncthreads();
  // LinvP[:, perm[c]] = Linv[:, c]  (Q_new = Linv * W[perm, :] = LinvP * W)
  for (int c = 0; c < na; ++c) j.LinvP[(size_t)t + (size_t)perm[c] * W] = t < na ? li[t * LD + c] : Scal<T>::zero();
  if (t < na) j.d[t] = real_of(g[t * LD + t]);
  if (t < b) oldp[t] = j.p[t];
  __syncthreads();
  if (t < b) {
    j.p[t] = oldp[perm[t]];
    j.lperm[t] = perm[t];
  }
  if (t == 0) {
    *j.nacc = na;
    if (j.first) *j.top = top;
  }
}
template <class T>
static void launch_chol(const CholJob<T>* d, unsigned n, hipStream_t s) {
  constexpr int lds = (int)(2 * HS_QW * (HS_QW + 1) * sizeof(T) + 2 * HS_QW * sizeof(int));
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KiB of LDS per workgroup needs the opt-in (complex: 131 KiB)
    (void)hipFuncSetAttribute((const void*)chol_block_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(chol_block_kernel<T>, dim3(n), dim3(64), lds, s, d);
}

// ------------------------------------------------------------------------------------------------
// Window selection by DOWNDATED RESIDUAL NORMS (QrJob::norm_select): the step from "tournament order + windowed Cholesky-QR" to a blocked
// column-pivoted QR (`pqrfact`, src/factorization.jl:171-182, picks the column of largest residual norm at every step).  res2[i] holds the
// squared residual norm of EVERY row of the block against the rows accepted so far: after a window accepted Q_new, C = M * Q_new^H (one
// grouped product over all rows -- it is also the block column of L_RS the interpolation needs, so the product at the end disappears) and
// res2[i] -= |C[i, :]|^2.  The next window is the <= 64 unaccepted rows of largest res2 (within 2^-7 of the largest in norm: what the window's
// pivoted Cholesky can accept anyway).  A downdated squared norm has lost its digits once it fell by ~1e10: the norms are then recomputed from
// the explicit residual M - L_RS * Q (LAPACK's xGEQP3 does the same per column).  No pivoted LU of a sketch is needed for the order.
// ------------------------------------------------------------------------------------------------
template <class T>
struct NormJob {
  const T* M;
  int ldm, m, q;
  double* res2;
  int minus;  // 0: res2 = |row|^2; 1: res2 = max(res2 - |row|^2, 0)  (M = the new coefficient columns)
  int* p;     // != null with minus == 0: also p[i] = i (the initial candidate list)
};
template <class T>
__global__ __launch_bounds__(256) void rownorm2_kernel(const NormJob<T>* __restrict__ jobs) {
  const NormJob<T> j = jobs[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= j.m) return;
  double acc = 0.0;
  for (int c = 0; c < j.q; ++c) {
    const T v = j.M[(size_t)i + (size_t)c * j.ldm];
    if constexpr (sizeof(T) == 16) acc += v.re * v.re + v.im * v.im; else acc += v * v;
  }
  if (j.minus) {
    j.res2[i] = fmax(j.res2[i] - acc, 0.0);
  } else {
    j.res2[i] = acc;
    if (j.p) j.p[i] = i;
  }
}
struct SelJob {
  int* p;            // candidate list: p[0 : done) accepted, p[done : m) the rest, re-ordered here: the selected rows first
  int* tmp;          // m ints
  const double* res2;
  double* outmax;    // [0] largest res2 among the unaccepted rows, [1] largest res2 among those NOT selected for the window
  int done, m, want;
};
// one workgroup per job: (1) largest res2 of the tail, (2) histogram of log2(max / res2), (3) stable partition of the tail: the first `want`
// rows of the buckets up to the cutoff in front (the cutoff is the smallest bucket that fills the window, at most 14: 2^-7 in norm)
__global__ __launch_bounds__(256) void qr_select_kernel(const SelJob* __restrict__ jobs) {
  const SelJob j = jobs[blockIdx.x];
  const int t = threadIdx.x, nt = j.m - j.done;
  __shared__ double s_red[256];
  __shared__ int s_hist[16], s_cut, s_scan[256], s_base[2];
  if (nt <= 0) {
    if (t == 0) j.outmax[0] = j.outmax[1] = 0.0;
    return;
  }
  int* tail = j.p + j.done;
  double mx = 0.0;
  for (int e = t; e < nt; e += 256) mx = fmax(mx, j.res2[tail[e]]);
  s_red[t] = mx;
  if (t < 16) s_hist[t] = 0;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) s_red[t] = fmax(s_red[t], s_red[t + w]);
    __syncthreads();
  }
  mx = s_red[0];
  if (t == 0) {
    j.outmax[0] = mx;
    j.outmax[1] = 0.0;
  }
  if (!(mx > 0.0) || nt <= j.want) return;  // nothing to order: every remaining row is in the window (or all are zero)
  auto bucket = [&](double v) {
    if (!(v > 0.0)) return 15;
    int e;
    (void)frexp(mx / v, &e);  // mx / v in [2^(e-1), 2^e)
    return min(max(e - 1, 0), 15);
  };
  for (int e = t; e < nt; e += 256) atomicAdd(&s_hist[bucket(j.res2[tail[e]])], 1);
  __syncthreads();
  if (t == 0) {
    int cum = 0, cut = 14;
    for (int b = 0; b <= 14; ++b) {
      cum += s_hist[b];
      if (cum >= j.want) { cut = b; break; }
    }
    s_cut = cut;
    s_base[0] = 0;  // selected so far
    s_base[1] = 0;  // unselected so far
  }
  __syncthreads();
  const int cut = s_cut;
  // pass A: how many rows qualify (capped at `want`): the rest of the tail starts behind them
  int nsel_total = 0;
  for (int b = 0; b <= cut; ++b) nsel_total += s_hist[b];
  nsel_total = min(nsel_total, j.want);
  for (int c0 = 0; c0 < nt; c0 += 256) {
    const int e = c0 + t;
    const int row = e < nt ? tail[e] : -1;
    int f = (row >= 0 && bucket(j.res2[row]) <= cut) ? 1 : 0;
    s_scan[t] = f;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // inclusive scan of the flags
      const int v = t >= off ? s_scan[t - off] : 0;
      __syncthreads();
      s_scan[t] += v;
      __syncthreads();
    }
    const int incl = s_scan[t], tot = s_scan[255];
    const int sel_before = s_base[0], uns_before = s_base[1];
    __syncthreads();
    if (row >= 0) {
      const int rank_sel = sel_before + incl - f;  // selected rows before this one
      const bool take = f && rank_sel < j.want;
      // position: selected rows in front; everything else behind them in the original order
      const int taken_before = min(rank_sel, j.want);
      const int untaken_before = (e - taken_before);  // rows before this one that are not taken = index - taken
      j.tmp[take ? taken_before : nsel_total + untaken_before] = row;
    }
    if (t == 0) {
      s_base[0] = sel_before + tot;
      s_base[1] = uns_before + (min(256, nt - c0) - tot);
    }
    __syncthreads();
  }
  double mo = 0.0;
  for (int e = t; e < nt; e += 256) {
    const int row = j.tmp[e];
    tail[e] = row;
    if (e >= nsel_total) mo = fmax(mo, j.res2[row]);
  }
  __syncthreads();
  s_red[t] = mo;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (t < w) s_red[t] = fmax(s_red[t], s_red[t + w]);
    __syncthreads();
  }
  if (t == 0) j.outmax[1] = s_red[0];
}

static std::atomic<int> g_qr_order{-1};  // -1: not read yet (HS_QR_ORDER=norm in the environment), 0: the tournament order, 1: residual norms
static bool qr_norm_order() {
  int v = g_qr_order.load();
  if (v < 0) {
    v = (getenv("HS_QR_ORDER") && getenv("HS_QR_ORDER")[0] == 'n') ? 1 : 0;
    g_qr_order.store(v);
  }
  return v == 1;
}
extern "C" int hs_hss_qr_order(int mode) {
  const int prev = qr_norm_order() ? 1 : 0;
  if (mode >= 0) g_qr_order.store(mode ? 1 : 0);
  return prev;
}
template <class T>
struct QrJob {
  const T* M;     // m x q block whose rows are interpolated (ld ldm), left untouched
  int ldm, m, q;
  int* p;         // device: candidate order of the rows (m entries; the pivot order of a tournament-pivoted LU), re-ordered in place
  int rmax;       // candidates: the first rmax rows of p
  double atol_scale = 1.0;  // the absolute tolerance is multiplied by this (rows of an un-normalised Gaussian sketch are sqrt(k) times longer)
  double floor_abs = 0.0;   // round-off level of the rows of M (absolute): candidates below it are noise whatever the tolerance says
  bool norm_select = false; // windows chosen by downdated residual norms of ALL rows (qr_select_kernel) instead of taken from the order `p` arrives in;
                            // `p` is then an output only (qr_refine fills it with 0 .. m-1 first)
  // results
  int r = 0;
  double top = 0.0;      // d_0
  T* Tm = nullptr;       // (m - r) x r, ld ldt, allocated from `out_pool` by qr_refine
  int ldt = 2;
};

// rank = number of accepted rows: d_j > max(atol, rtol * max(d_0, scale_floor)) in a windowed-pivoted order (window: 32 candidates)
template <class T>
void qr_refine(Pool& tmp, Pool& out_pool, std::vector<QrJob<T>>& jobs, double atol, double rtol, double scale_floor, hipStream_t s) {
  const int nj = (int)jobs.size();
  if (nj == 0) return;
  struct St {
    T *Q = nullptr, *Qh = nullptr, *L = nullptr, *Linv = nullptr, *LinvP = nullptr, *W = nullptr, *Wh = nullptr, *C1 = nullptr, *C2 = nullptr, *G = nullptr;
    double *d = nullptr, *top = nullptr;
    int *lperm = nullptr, *nacc = nullptr;
    T* slab = nullptr;  // 32 x 32 slots for the blocks' inverse factors
    int nslab = 0, used = 0;
    int ldq = 2, ldqh = 2, ldl = 2, done = 0, active = 1, maxblk = 0, stall = 0;
    std::vector<std::pair<int, int>> blocks;  // (offset, width) of the accepted blocks
    // QrJob::norm_select
    double* res2 = nullptr;  // squared residual norm of every row
    T* LRS = nullptr;        // m x rmax: coefficients of every row against the accepted rows (the interpolation's L_RS, block column by block column)
    int ldr = 2;
    int* ptmp = nullptr;
    double* dmax = nullptr;  // device: largest res2 of the unaccepted rows at the last selection
    double refmax = 0.0;     // the same when the norms were last exact
  };
  std::vector<St> st(nj);
  for (int a = 0; a < nj; ++a) {
    QrJob<T>& J = jobs[a];
    St& S = st[a];
    J.rmax = std::max(0, std::min(J.rmax, std::min(J.m, J.q)));
    J.r = 0;
    if (J.rmax == 0) {
      S.active = 0;
      continue;
    }
    S.maxblk = J.rmax + 2;  // every step accepts at least one row or ends the job
    S.ldq = ev(J.rmax);
    S.ldqh = ev(J.q);
    S.ldl = ev(J.rmax);
    S.Q = tmp.get<T>((size_t)S.ldq * J.q);
    S.Qh = tmp.get<T>((size_t)S.ldqh * J.rmax);
    S.L = tmp.getz<T>((size_t)S.ldl * J.rmax, s);
    S.W = tmp.get<T>((size_t)HS_QW * J.q);
    S.Wh = tmp.get<T>((size_t)S.ldqh * HS_QW);
    S.C1 = tmp.get<T>((size_t)HS_QW * J.rmax);
    S.C2 = tmp.get<T>((size_t)HS_QW * J.rmax);
    S.G = tmp.get<T>(HS_QW * HS_QW);
    S.d = tmp.get<double>((size_t)J.rmax + HS_QW);
    S.top = tmp.get<double>(4);
    S.lperm = tmp.get<int>(HS_QW);
    S.nslab = J.rmax / (HS_QW / 2) + 8;
    S.slab = tmp.get<T>((size_t)S.nslab * 2 * HS_QW * HS_QW);
    if (J.norm_select) {
      S.res2 = tmp.get<double>((size_t)J.m);
      S.ldr = ev(J.m);
      S.LRS = tmp.get<T>((size_t)S.ldr * J.rmax);
      S.ptmp = tmp.get<int>((size_t)J.m);
      S.dmax = tmp.get<double>(2);
    }
  }
  std::vector<NormJob<T>> normjobs;
  auto run_norms = [&]() {
    if (normjobs.empty()) return;
    int mm = 0;
    for (auto& q : normjobs) mm = std::max(mm, q.m);
    NormJob<T>* d = upload(tmp, normjobs, s);
    hipLaunchKernelGGL(rownorm2_kernel<T>, dim3((unsigned)((mm + 255) / 256), (unsigned)normjobs.size()), dim3(256), 0, s, (const NormJob<T>*)d);
    normjobs.clear();
  };
  for (int a = 0; a < nj; ++a)
    if (jobs[a].norm_select && st[a].active) normjobs.push_back(NormJob<T>{jobs[a].M, jobs[a].ldm, jobs[a].m, jobs[a].q, st[a].res2, 0, jobs[a].p});
  run_norms();
  std::vector<double> hmax(2 * (size_t)nj, 0.0);
  double* dmaxall = tmp.get<double>(2 * (size_t)nj);
  static const double theta = getenv("HS_QR_THETA") ? atof(getenv("HS_QR_THETA")) : 0.5;  // relaxation of the greedy order (1: strict)
  int* dnacc = tmp.get<int>((size_t)nj);
  for (int a = 0; a < nj; ++a) st[a].nacc = dnacc + a;
  std::vector<int> hacc(nj, 0);
  std::vector<RowJob<T>> rows;
  std::vector<SubJob<T>> subs;
  std::vector<GemmProb<T>> g;
  std::vector<std::vector<T*>> linv(nj), linvp(nj);  // one 32 x 32 slot per accepted block
  static const double chol_cond = getenv("HS_CHOL_COND") ? atof(getenv("HS_CHOL_COND")) : HS_CHOL_COND;  // diagnostics
  static const bool qtime = getenv("HS_QR_TIMING") != nullptr;  // diagnostics: wall time of the phases (adds nothing: every step syncs anyway)
  auto tq0 = std::chrono::steady_clock::now();
  int nsteps = 0;
  for (;;) {
    ++nsteps;
    auto each = [&](auto&& f) {
      for (int a = 0; a < nj; ++a)
        if (st[a].active) f(a, jobs[a], st[a], std::min(HS_QW, jobs[a].rmax - st[a].done));
    };
    {  // norm_select: the window's candidates = the unaccepted rows of largest residual norm, moved to the front of the rest of the list
      std::vector<SelJob> sj;
      each([&](int a, QrJob<T>& J, St& S, int b) {
        if (J.norm_select) sj.push_back(SelJob{J.p, S.ptmp, S.res2, dmaxall + 2 * a, S.done, J.m, b});
      });
      if (!sj.empty()) {
        SelJob* dsj = upload(tmp, sj, s);
        hipLaunchKernelGGL(qr_select_kernel, dim3((unsigned)sj.size()), dim3(256), 0, s, (const SelJob*)dsj);
      }
    }
    // W = M[p[done : done+b], :]
    each([&](int, QrJob<T>& J, St& S, int b) { rows.push_back(RowJob<T>{J.M, J.ldm, S.W, HS_QW, J.p + S.done, b, J.q, ROW_GATHER}); });
    if (rows.empty()) break;
    run_rows(tmp, rows, s);
    for (int pass = 0; pass < 2; ++pass) {  // classical Gram-Schmidt against the accepted rows, twice
      each([&](int, QrJob<T>& J, St& S, int b) {
        if (S.done == 0) return;
        T* Cx = pass == 0 ? S.C1 : S.C2;
        // (C = A*B overwrites the b x done block that is read: no memset per window)
        g.push_back(GemmProb<T>{S.W, S.Qh, Cx, b, S.done, J.q, HS_QW, S.ldqh, HS_QW});
      });
      run_gemms(tmp, g, 0, s);
      each([&](int, QrJob<T>& J, St& S, int b) {
        if (S.done > 0) g.push_back(GemmProb<T>{pass == 0 ? S.C1 : S.C2, S.Q, S.W, b, J.q, S.done, HS_QW, S.ldq, HS_QW});
      });
      run_gemms(tmp, g, 1, s);
    }
    each([&](int, QrJob<T>&, St& S, int b) {  // C1 += C2: the coefficients of the candidates against the accepted rows
      if (S.done > 0) rows.push_back(RowJob<T>{S.C2, HS_QW, S.C1, HS_QW, nullptr, b, S.done, ROW_SCATTER_ADD});
    });
    run_rows(tmp, rows, s);
    // G = W * W^H, pivoted Cholesky of the window
    each([&](int, QrJob<T>& J, St& S, int b) { subs.push_back(SubJob<T>{S.W, HS_QW, nullptr, nullptr, 0, 0, b, J.q, S.Wh, S.ldqh, 2}); });
    run_subs(tmp, subs, s);
    std::vector<CholJob<T>> cj;
    each([&](int a, QrJob<T>& J, St& S, int b) {
      g.push_back(GemmProb<T>{S.W, S.Wh, S.G, b, b, J.q, HS_QW, S.ldqh, HS_QW});
      T* slot = S.used < S.nslab ? S.slab + (size_t)(S.used++) * 2 * HS_QW * HS_QW : tmp.get<T>(2 * HS_QW * HS_QW);
      linv[a].push_back(slot);
      linvp[a].push_back(slot + HS_QW * HS_QW);
      cj.push_back(CholJob<T>{S.G, S.L + S.done + (size_t)S.done * S.ldl, S.ldl, slot, slot + HS_QW * HS_QW, S.d + S.done, S.top, J.p + S.done, S.lperm, S.nacc, atol * J.atol_scale,
                              rtol, scale_floor, b, S.done == 0 ? 1 : 0, J.floor_abs, chol_cond, noise_rel(), J.norm_select ? dmaxall + 2 * a + 1 : nullptr, theta * theta});
    });
    run_gemms(tmp, g, 0, s);
    CholJob<T>* dcj = upload(tmp, cj, s);
    launch_chol<T>(dcj, (unsigned)cj.size(), s);
    // how many candidates each job accepted
    HSS_HIP(hipMemcpyAsync(hacc.data(), dnacc, sizeof(int) * (size_t)nj, hipMemcpyDeviceToHost, s));
    HSS_HIP(hipMemcpyAsync(hmax.data(), dmaxall, sizeof(double) * 2 * (size_t)nj, hipMemcpyDeviceToHost, s));
    HSS_HIP(hipStreamSynchronize(s));
    // Q[done : done+b', :] = LinvP * W, its conjugate transpose, and the rows of L against the earlier blocks (in accepted order)
    each([&](int a, QrJob<T>& J, St& S, int b) {
      const int na = hacc[a];
      if (na <= 0) return;
      g.push_back(GemmProb<T>{linvp[a].back(), S.W, S.Q + S.done, na, J.q, b, HS_QW, HS_QW, S.ldq});
      if (S.done > 0) rows.push_back(RowJob<T>{S.C1, HS_QW, S.L + S.done, S.ldl, S.lperm, na, S.done, ROW_GATHER});
    });
    run_gemms(tmp, g, 0, s);
    run_rows(tmp, rows, s);
    each([&](int a, QrJob<T>& J, St& S, int) {
      const int na = hacc[a];
      if (na > 0) subs.push_back(SubJob<T>{S.Q + S.done, S.ldq, nullptr, nullptr, 0, 0, na, J.q, S.Qh + (size_t)S.done * S.ldqh, S.ldqh, 2});
    });
    run_subs(tmp, subs, s);
    {  // norm_select: coefficients of EVERY row against the rows accepted in this window, and the downdate of the residual norms
      each([&](int a, QrJob<T>& J, St& S, int) {
        const int na = hacc[a];
        if (!J.norm_select || na <= 0) return;
        g.push_back(GemmProb<T>{J.M, S.Qh + (size_t)S.done * S.ldqh, S.LRS + (size_t)S.done * S.ldr, J.m, na, J.q, J.ldm, S.ldqh, S.ldr});
      });
      run_gemms(tmp, g, 0, s);
      each([&](int a, QrJob<T>& J, St& S, int) {
        const int na = hacc[a];
        if (!J.norm_select || na <= 0) return;
        normjobs.push_back(NormJob<T>{S.LRS + (size_t)S.done * S.ldr, S.ldr, J.m, na, S.res2, 1, nullptr});
      });
      run_norms();
    }
    HSS_HIP(hipStreamSynchronize(s));  // lperm / nacc are rewritten by the next window
    for (int a = 0; a < nj; ++a) {
      St& S = st[a];
      if (!S.active) continue;
      const int b = std::min(HS_QW, jobs[a].rmax - S.done), na = hacc[a];
      if (na > 0) {
        S.blocks.push_back({S.done, na});
        S.done += na;
      } else {
        linv[a].pop_back();
        linvp[a].pop_back();
      }
      // fewer accepted than offered: the first rejected candidate was below the truncation threshold (the job is finished) or badly
      // conditioned against the rows accepted in this window (it is looked at again, orthogonalised against them); a window that
      // accepts nothing ends the job
      if (S.done >= jobs[a].rmax || na == 0) S.active = 0;
      (void)b;
    }
    {  // norm_select: once the largest downdated norm^2 has fallen by 1e10 since the norms were last exact it has lost its digits: recompute
       // every row's residual from M - L_RS * Q (one product with K = rows accepted so far; twice or so in a compression at 1e-12)
      std::vector<int> fresh;
      for (int a = 0; a < nj; ++a) {
        St& S = st[a];
        if (!S.active || !jobs[a].norm_select) continue;
        if (S.refmax == 0.0) S.refmax = hmax[2 * a];
        if (hmax[2 * a] < 1e-10 * S.refmax && S.done > 0) fresh.push_back(a);
      }
      if (!fresh.empty()) {
        std::vector<T*> Rt(fresh.size(), nullptr);
        std::vector<SubJob<T>> cp;
        for (size_t f = 0; f < fresh.size(); ++f) {
          const int a = fresh[f];
          Rt[f] = tmp.get<T>((size_t)ev(jobs[a].m) * jobs[a].q);
          cp.push_back(SubJob<T>{jobs[a].M, jobs[a].ldm, nullptr, nullptr, 0, 0, jobs[a].m, jobs[a].q, Rt[f], ev(jobs[a].m), 0});
        }
        run_subs(tmp, cp, s);
        for (size_t f = 0; f < fresh.size(); ++f) {
          const int a = fresh[f];
          g.push_back(GemmProb<T>{st[a].LRS, st[a].Q, Rt[f], jobs[a].m, jobs[a].q, st[a].done, st[a].ldr, st[a].ldq, ev(jobs[a].m)});
        }
        run_gemms(tmp, g, 1, s);
        for (size_t f = 0; f < fresh.size(); ++f) {
          const int a = fresh[f];
          normjobs.push_back(NormJob<T>{Rt[f], ev(jobs[a].m), jobs[a].m, jobs[a].q, st[a].res2, 0, nullptr});
          st[a].refmax = hmax[2 * a];
        }
        run_norms();
      }
    }
    // a rejected candidate above the threshold keeps the job going; one below it ends the job -- told apart by the next window: its first pivot is then <= tau
  }
  auto tq1 = std::chrono::steady_clock::now();
  // ranks and d_0
  int maxblocks = 0;
  for (int a = 0; a < nj; ++a) {
    QrJob<T>& J = jobs[a];
    St& S = st[a];
    J.r = S.done;
    J.ldt = ev(J.m - J.r);
    J.Tm = out_pool.get<T>((size_t)J.ldt * std::max(J.r, 1));
    maxblocks = std::max(maxblocks, (int)S.blocks.size());
    if (S.top) HSS_HIP(hipMemcpyAsync(&J.top, S.top, sizeof(double), hipMemcpyDeviceToHost, s));
  }
  HSS_HIP(hipStreamSynchronize(s));
  for (int a = 0; a < nj; ++a)
    if (jobs[a].r == 0) jobs[a].top = std::max(jobs[a].top, 0.0);
  static const bool qdebug = getenv("HS_QR_DEBUG") != nullptr;  // diagnostics: the accepted d_j of the job of largest rank, window by window
  if (qdebug) {
    int am = 0;
    for (int a = 1; a < nj; ++a)
      if (jobs[a].r > jobs[am].r) am = a;
    const QrJob<T>& J = jobs[am];
    std::vector<double> hd((size_t)std::max(J.r, 1));
    if (J.r > 0) HSS_HIP(hipMemcpy(hd.data(), st[am].d, sizeof(double) * (size_t)J.r, hipMemcpyDeviceToHost));
    fprintf(stderr, "[hs qr debug] %d jobs; job %d: m %d q %d rmax %d -> r %d, top %.3e, tau_abs %.3e tau_rel %.3e floor %.3e; windows (offset:width d_first..d_last):", nj, am, J.m, J.q, J.rmax, J.r, J.top,
            atol * J.atol_scale, rtol * J.top, J.floor_abs);
    for (auto& b : st[am].blocks) fprintf(stderr, " %d:%d %.2e..%.2e", b.first, b.second, hd[b.first], hd[b.first + b.second - 1]);
    fprintf(stderr, "\n");
  }
  // T = (M[p_R, :] * Q_S^H) * L_SS^-1, block columns from the right
  std::vector<T*> YR(nj, nullptr), T2(nj, nullptr);
  for (int a = 0; a < nj; ++a) {
    QrJob<T>& J = jobs[a];
    St& S = st[a];
    const int nR = J.m - J.r;
    if (nR <= 0 || J.r <= 0) continue;
    T2[a] = tmp.get<T>((size_t)ev(nR) * HS_QW);
    if (J.norm_select) {  // L_RS is already there, block column by block column: the rows of the redundant candidates
      rows.push_back(RowJob<T>{S.LRS, S.ldr, J.Tm, J.ldt, J.p + J.r, nR, J.r, ROW_GATHER});
      continue;
    }
    YR[a] = tmp.get<T>((size_t)ev(nR) * J.q);
    rows.push_back(RowJob<T>{J.M, J.ldm, YR[a], ev(nR), J.p + J.r, nR, J.q, ROW_GATHER});
    HSS_HIP(hipMemsetAsync(J.Tm, 0, sizeof(T) * (size_t)J.ldt * J.r, s));
    g.push_back(GemmProb<T>{YR[a], S.Qh, J.Tm, nR, J.r, J.q, ev(nR), S.ldqh, J.ldt});
  }
  run_rows(tmp, rows, s);
  run_gemms(tmp, g, 0, s);  // Tm <- L_RS
  for (int step = 0; step < maxblocks; ++step) {
    std::vector<GemmProb<T>> g1, g2;
    std::vector<SubJob<T>> back;
    for (int a = 0; a < nj; ++a) {
      QrJob<T>& J = jobs[a];
      St& S = st[a];
      const int nR = J.m - J.r, r = J.r, nb_ = (int)S.blocks.size();
      if (nR <= 0 || step >= nb_) continue;
      const int bi = nb_ - 1 - step, j0 = S.blocks[bi].first, w = S.blocks[bi].second, j1 = j0 + w;
      if (r > j1) g1.push_back(GemmProb<T>{J.Tm + (size_t)j1 * J.ldt, S.L + j1 + (size_t)j0 * S.ldl, J.Tm + (size_t)j0 * J.ldt, nR, w, r - j1, J.ldt, S.ldl, J.ldt});
      g2.push_back(GemmProb<T>{J.Tm + (size_t)j0 * J.ldt, linv[a][bi], T2[a], nR, w, w, J.ldt, HS_QW, ev(nR)});
      back.push_back(SubJob<T>{T2[a], ev(nR), nullptr, nullptr, 0, 0, nR, w, J.Tm + (size_t)j0 * J.ldt, J.ldt, 0});
    }
    run_gemms(tmp, g1, 1, s);
    run_gemms(tmp, g2, 0, s);
    run_subs(tmp, back, s);
  }
  HSS_HIP(hipStreamSynchronize(s));
  if (qtime) {
    auto tq2 = std::chrono::steady_clock::now();
    int mr = 0, mm = 0, mq = 0;
    for (auto& J : jobs) { mr = std::max(mr, J.r); mm = std::max(mm, J.m); mq = std::max(mq, J.q); }
    fprintf(stderr, "[hs qr] %d jobs (m <= %d, q <= %d, rank <= %d): %d windows %.2f ms, interpolation (%d block steps) %.2f ms\n", nj, mm, mq, mr, nsteps,
            std::chrono::duration<double, std::milli>(tq1 - tq0).count(), maxblocks, std::chrono::duration<double, std::milli>(tq2 - tq1).count());
  }
}

template <class T>
struct BlockOp;  // hs_hss_op.h: [H1 A12; A21 H2] of two HSS blocks and sparse couplings, never formed
// One matrix of a BATCHED compression: the fronts of a tree level hand over Schur complements of the same kind, and compressed one by one
// each is a chain of ~10,000 small dependent launches (110 ms for n = 12,097 whatever the chip could do besides).  compress_fixed
// therefore works on a FOREST: the cluster trees of all the matrices side by side in one node array (roots at level 0, the index space
// concatenated), every stage of a tree level one group of launches for all of them.  One matrix = a forest of one tree.
template <class T>
struct CBlock {
  const T* A = nullptr;  // the matrix (device), or null with an operator
  int lda = 0, n = 0, off = 0;        // its rows are [off, off + n) of the concatenated index space
  const int* perm = nullptr;          // device: H ~= A[perm, perm] (values local to the block), or null
  const int* hperm = nullptr;         // the same on the host
  Lru<T> lru;
  int leafsize = 64, first_split = 0;
  uint64_t seed = 0;
  int root = 0;                       // node of its root (set by compress_fixed)
  T* MT = nullptr;                    // M^T of its update (work)
  double gscale = 0.0, gscale_q = 0.0;  // largest sample pivot / row norm seen so far: the scale of ITS off-diagonal part
};
template <class T>
bool compress_fixed(HssT<T>& H, std::vector<CBlock<T>>& cb, int k, BlockOp<T>* bop = nullptr) {
  const int n = H.n;
  const int F = (int)cb.size();
  hipStream_t s = H.s;
  Pool tmp(global_cache());  // samples and everything else that dies with this attempt
  tmp.guard = s;
  static const bool vtime = getenv("HS_HSS_VERBOSE") != nullptr;  // diagnostics: wall time of every phase (adds synchronisations)
  auto vt0 = std::chrono::steady_clock::now();
  auto vlap = [&](const char* what, int lv) {
    if (!vtime) return;
    (void)hipStreamSynchronize(s);
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[hs hss]      n=%d k=%d level %d: %-34s %8.3f ms\n", n, k, lv, what, std::chrono::duration<double, std::milli>(now - vt0).count());
    vt0 = now;
  };
  H.keep.clear();
  std::vector<int> nblk;  // node -> matrix
  if (F == 1) {
    build_tree(H, n, cb[0].leafsize, cb[0].first_split);
    cb[0].root = 0;
    cb[0].off = 0;
    nblk.assign(H.nd.size(), 0);
  } else {  // the trees side by side: node ids and index ranges shifted, every root at level 0
    H.nd.clear();
    int off = 0;
    for (int b = 0; b < F; ++b) {
      HssT<T> one;
      build_tree(one, cb[b].n, cb[b].leafsize, cb[b].first_split);
      const int base = (int)H.nd.size();
      cb[b].root = base;
      cb[b].off = off;
      for (HNode<T> y : one.nd) {
        y.lo += off;
        y.hi += off;
        if (y.parent >= 0) y.parent += base;
        if (y.left >= 0) {
          y.left += base;
          y.right += base;
        }
        H.nd.push_back(y);
        nblk.push_back(b);
      }
      off += cb[b].n;
    }
    H.nlev = 0;
    for (auto& x : H.nd) H.nlev = std::max(H.nlev, x.level + 1);
    H.lev.assign(H.nlev, {});
    for (int i = 0; i < (int)H.nd.size(); ++i) H.lev[H.nd[i].level].push_back(i);
  }
  H.k = k;
  const int k2 = 2 * k, ldn = ev(n), ldk = ev(k);
  auto& nd = H.nd;
  std::vector<SubJob<T>> subs;
  std::vector<RowJob<T>> rows;
  std::vector<GemmProb<T>> gemms;
  {  // M^T for the transposed gathers
    std::vector<SubJob<T>> tr;
    for (auto& B : cb)
      if (B.lru.on() && B.lru.M) {
        B.MT = tmp.get<T>((size_t)ev(B.lru.r2) * B.lru.r1);
        tr.push_back(SubJob<T>{B.lru.M, B.lru.ldm, nullptr, nullptr, 0, 0, B.lru.r1, B.lru.r2, B.MT, ev(B.lru.r2), 1});
      }
    run_subs(tmp, tr, s);
  }
  // blocks of A by index lists (the jobs in `subs` all read A): B's entries, minus the low-rank update's
  auto gather_A = [&]() {
    std::vector<SubJob<T>> jobs = subs;
    if (bop) {
      bop->gather(tmp, subs, s);
      subs.clear();
    } else {
      run_subs(tmp, subs, s);
    }
    std::vector<SubJob<T>> pieces;
    std::vector<GemmProb<T>> g1, g2;
    for (const SubJob<T>& j : jobs) {
      const Lru<T>& lru = cb[j.blk].lru;
      T* MT = cb[j.blk].MT;
      if (!lru.on()) continue;
      if (j.rows <= 0 || j.cols <= 0) continue;
      const int r1 = lru.r1, r2 = lru.r2;
      if (!j.trans) {
        T* Cg = tmp.get<T>((size_t)ev(j.rows) * r1);
        T* Zg = tmp.get<T>((size_t)ev(r2) * j.cols);
        T* T1 = lru.M ? tmp.getz<T>((size_t)ev(j.rows) * r2, s) : Cg;
        pieces.push_back(SubJob<T>{lru.C, lru.ldc, j.ri, nullptr, j.r0, 0, j.rows, r1, Cg, ev(j.rows), 0});
        pieces.push_back(SubJob<T>{lru.Z, lru.ldz, nullptr, j.ci, 0, j.c0, r2, j.cols, Zg, ev(r2), 0});
        if (lru.M) g1.push_back(GemmProb<T>{Cg, lru.M, T1, j.rows, r2, r1, ev(j.rows), lru.ldm, ev(j.rows)});
        g2.push_back(GemmProb<T>{T1, Zg, j.out, j.rows, j.cols, r2, ev(j.rows), ev(r2), j.ldo});
      } else {  // out is cols x rows:  out -= Z[:, J]^T * (M^T * C[I, :]^T)
        T* CgT = tmp.get<T>((size_t)ev(r1) * j.rows);
        T* ZgT = tmp.get<T>((size_t)ev(j.cols) * r2);
        T* T1T = lru.M ? tmp.getz<T>((size_t)ev(r2) * j.rows, s) : CgT;
        pieces.push_back(SubJob<T>{lru.C, lru.ldc, j.ri, nullptr, j.r0, 0, j.rows, r1, CgT, ev(r1), 1});
        pieces.push_back(SubJob<T>{lru.Z, lru.ldz, nullptr, j.ci, 0, j.c0, r2, j.cols, ZgT, ev(j.cols), 1});
        if (lru.M) g1.push_back(GemmProb<T>{MT, CgT, T1T, r2, j.rows, r1, ev(r2), ev(r1), ev(r2)});
        g2.push_back(GemmProb<T>{ZgT, T1T, j.out, j.cols, j.rows, r2, ev(j.cols), ev(r2), j.ldo});
      }
    }
    run_subs(tmp, pieces, s);
    run_gemms(tmp, g1, 0, s);
    run_gemms(tmp, g2, 1, s);
  };
  if (F > 1) {
    for (auto& B : cb)
      if (nd[B.root].left < 0 || bop) {
        hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: a batched compression needs matrices (no operators) of at least two leaves each");
        throw (int)HS_ERR_ARGUMENT;
      }
  } else if (nd[0].left < 0) {  // a single leaf: H = D
    HNode<T>& x = nd[0];
    x.m = n;
    x.ldd = ev(n);
    x.D = H.keep.template get<T>((size_t)x.ldd * n);
    subs.push_back(SubJob<T>{cb[0].A, cb[0].lda, cb[0].perm, cb[0].perm, 0, 0, n, n, x.D, x.ldd, 0, cb[0].hperm, cb[0].hperm});
    gather_A();
    HSS_HIP(hipStreamSynchronize(s));
    return true;
  }
  // test matrices OP = [Omega | Psi] (n x 2k) and samples Y = [B*Omega | B^T*Psi] of B = A[perm, perm]:
  // B*Omega = (A * Omega')[perm] with Omega'[perm[i]] = Omega[i]; B^T*Psi likewise from Psi'^T * A
  T* OP = tmp.get<T>((size_t)ldn * k2);
  T* Y = tmp.get<T>((size_t)ldn * k2);
  T* PsT = tmp.get<T>((size_t)ldk * n);
  T* W = tmp.get<T>((size_t)ldk * n);
  fill_randn<T>(OP, n, ldn, k2, (uint64_t)cb[0].seed * 0x9E3779B97F4A7C15ull + 17 * (uint64_t)k, s);
  T *OPs = OP, *Ys = Y;
  bool any_perm = false;
  for (auto& B : cb) any_perm = any_perm || B.perm != nullptr;
  if (any_perm) {
    OPs = tmp.get<T>((size_t)ldn * k2);
    Ys = tmp.get<T>((size_t)ldn * k2);
    for (auto& B : cb) rows.push_back(RowJob<T>{OP + B.off, ldn, OPs + B.off, ldn, B.perm, B.n, k2, B.perm ? ROW_SCATTER : ROW_GATHER});
    run_rows(tmp, rows, s);
  }
  subs.push_back(SubJob<T>{OPs + (size_t)ldn * k, ldn, nullptr, nullptr, 0, 0, n, k, PsT, ldk, 1});
  run_subs(tmp, subs, s);
  if (bop) {  // products with the operator and its transpose: Ys[:, :k] = Op*Omega', W = (Op^T*Psi')^T
    T* Y2 = tmp.get<T>((size_t)ldn * k);
    bop->mul(OPs, ldn, Ys, ldn, k, false, s);
    bop->mul(OPs + (size_t)ldn * k, ldn, Y2, ldn, k, true, s);
    subs.push_back(SubJob<T>{Y2, ldn, nullptr, nullptr, 0, 0, n, k, W, ldk, 1});
    run_subs(tmp, subs, s);
  } else {
    for (auto& B : cb) {
      gemms.push_back(GemmProb<T>{B.A, OPs + B.off, Ys + B.off, B.n, k, B.n, B.lda, ldn, ldn});
      gemms.push_back(GemmProb<T>{PsT + (size_t)B.off * ldk, B.A, W + (size_t)B.off * ldk, k, B.n, B.n, ldk, B.lda, ldk});
    }
    run_gemms(tmp, gemms, 0, s);
  }
  // round-off level of every matrix's samples: HS_NOISE_REL * its largest row norm, before and after the low-rank update is taken off
  unsigned long long* dynorm = reinterpret_cast<unsigned long long*>(tmp.getz<double>((size_t)F, s));
  auto sample_scale = [&](const T* Yx, int cols) {
    for (int b = 0; b < F; ++b)
      hipLaunchKernelGGL(rownorm_max_kernel<T>, dim3((unsigned)((cb[b].n + 255) / 256)), dim3(256), 0, s, Yx + cb[b].off, ldn, cb[b].n, cols, dynorm + b);
  };
  sample_scale(Ys, k);
  {  // Ys -= C*(M*(Z*OPs)),  W -= ((PsT*C)*M)*Z for the matrices that carry a low-rank update
    std::vector<GemmProb<T>> ga, gb, gc;
    for (auto& B : cb) {
      const Lru<T>& lru = B.lru;
      if (!lru.on()) continue;
      const int r1 = lru.r1, r2 = lru.r2;
      T* t1 = tmp.getz<T>((size_t)ev(r2) * k, s);
      T* u1 = tmp.getz<T>((size_t)ldk * r1, s);
      T* t2 = lru.M ? tmp.getz<T>((size_t)ev(r1) * k, s) : t1;  // M == nullptr: the identity (r1 == r2)
      T* u2 = lru.M ? tmp.getz<T>((size_t)ldk * r2, s) : u1;
      ga.push_back(GemmProb<T>{lru.Z, OPs + B.off, t1, r2, k, B.n, lru.ldz, ldn, ev(r2)});
      ga.push_back(GemmProb<T>{PsT + (size_t)B.off * ldk, lru.C, u1, k, r1, B.n, ldk, lru.ldc, ldk});
      if (lru.M) {
        gb.push_back(GemmProb<T>{lru.M, t1, t2, r1, k, r2, lru.ldm, ev(r2), ev(r1)});
        gb.push_back(GemmProb<T>{u1, lru.M, u2, k, r2, r1, ldk, lru.ldm, ldk});
      }
      gc.push_back(GemmProb<T>{lru.C, t2, Ys + B.off, B.n, k, r1, lru.ldc, ev(r1), ldn});
      gc.push_back(GemmProb<T>{u2, lru.Z, W + (size_t)B.off * ldk, k, B.n, r2, ldk, lru.ldz, ldk});
    }
    run_gemms(tmp, ga, 0, s);
    run_gemms(tmp, gb, 0, s);
    run_gemms(tmp, gc, 1, s);
  }
  subs.push_back(SubJob<T>{W, ldk, nullptr, nullptr, 0, 0, k, n, Ys + (size_t)ldn * k, ldn, 1});
  run_subs(tmp, subs, s);
  if (any_perm) {
    for (auto& B : cb) rows.push_back(RowJob<T>{Ys + B.off, ldn, Y + B.off, ldn, B.perm, B.n, k2, ROW_GATHER});
    run_rows(tmp, rows, s);
  }

  sample_scale(Y, k2);
  std::vector<double> noise(F, 0.0);
  HSS_HIP(hipMemcpyAsync(noise.data(), dynorm, sizeof(double) * (size_t)F, hipMemcpyDeviceToHost, s));
  HSS_HIP(hipStreamSynchronize(s));
  double noise_max = 0.0;
  for (double& x : noise) {
    x *= noise_rel();
    noise_max = std::max(noise_max, x);
  }
  vlap(bop ? "samples Op*Omega, Op^T*Psi (matrix-free)" : "samples A*Omega, Psi^T*A", 0);
  const int N = (int)nd.size();
  // local sample / test blocks of every node (leaf: rows lo:hi of Y / OP), global index of every local position
  std::vector<T*> Yl(N, nullptr), Ol(N, nullptr);
  std::vector<int> ldl(N, 0);
  std::vector<int*> Jidx(N, nullptr);
  for (int i = 0; i < N; ++i)
    if (nd[i].left < 0) {
      nd[i].m = nd[i].hi - nd[i].lo;
      Yl[i] = Y + nd[i].lo;
      Ol[i] = OP + nd[i].lo;
      ldl[i] = ldn;
    }
  for (auto& B : cb) B.gscale = B.gscale_q = 0.0;  // largest sample pivot / row norm seen so far (deeper levels), per matrix
  for (int lv = H.nlev - 1; lv >= 1; --lv) {
    const std::vector<int>& L = H.lev[lv];
    const int nj = (int)L.size();
    // ---- a. remove the node's own diagonal block from its samples ----------------------------------------------------
    std::vector<T*> DT(N, nullptr), B12T(N, nullptr), B21T(N, nullptr);
    for (int i : L) {
      HNode<T>& x = nd[i];
      const int m = x.m;
      const int bi = nblk[i];
      const CBlock<T>& B = cb[bi];
      const T* A = B.A;
      const int lda = B.lda;
      if (x.left < 0) {
        x.ldd = ev(m);
        x.D = H.keep.template get<T>((size_t)x.ldd * m);
        DT[i] = tmp.get<T>((size_t)x.ldd * m);
        const int lo_loc = x.lo - B.off;
        const int* pl = B.perm ? B.perm + lo_loc : nullptr;  // rows / columns of A behind the leaf's positions
        const int o0 = B.perm ? 0 : lo_loc;
        const int* hpl = B.hperm ? B.hperm + lo_loc : nullptr;
        subs.push_back(SubJob<T>{A, lda, pl, pl, o0, o0, m, m, x.D, x.ldd, 0, hpl, hpl, bi});
        subs.push_back(SubJob<T>{A, lda, pl, pl, o0, o0, m, m, DT[i], x.ldd, 1, hpl, hpl, bi});
        gemms.push_back(GemmProb<T>{x.D, Ol[i], Yl[i], m, k, m, x.ldd, ldl[i], ldl[i]});
        gemms.push_back(GemmProb<T>{DT[i], Ol[i] + (size_t)ldl[i] * k, Yl[i] + (size_t)ldl[i] * k, m, k, m, x.ldd, ldl[i], ldl[i]});
      } else {
        const HNode<T>&l = nd[x.left], &r = nd[x.right];
        const int rl = l.r, rr = r.r, ld = ldl[i];
        x.ld12 = ev(rl);
        x.ld21 = ev(rr);
        x.B12 = H.keep.template get<T>((size_t)x.ld12 * rr);
        x.B21 = H.keep.template get<T>((size_t)x.ld21 * rl);
        B12T[i] = tmp.get<T>((size_t)x.ld21 * rl);  // rr x rl
        B21T[i] = tmp.get<T>((size_t)x.ld12 * rr);  // rl x rr
        subs.push_back(SubJob<T>{A, lda, l.sk, r.sk, 0, 0, rl, rr, x.B12, x.ld12, 0, l.hsk.data(), r.hsk.data(), bi});
        subs.push_back(SubJob<T>{A, lda, r.sk, l.sk, 0, 0, rr, rl, x.B21, x.ld21, 0, r.hsk.data(), l.hsk.data(), bi});
        subs.push_back(SubJob<T>{A, lda, l.sk, r.sk, 0, 0, rl, rr, B12T[i], x.ld21, 1, l.hsk.data(), r.hsk.data(), bi});
        subs.push_back(SubJob<T>{A, lda, r.sk, l.sk, 0, 0, rr, rl, B21T[i], x.ld12, 1, r.hsk.data(), l.hsk.data(), bi});
        T *Yi = Yl[i], *Oi = Ol[i];
        const size_t ck = (size_t)ld * k;
        gemms.push_back(GemmProb<T>{x.B12, Oi + rl, Yi, rl, k, rr, x.ld12, ld, ld});              // Sr_l -= B12 * Om~_r
        gemms.push_back(GemmProb<T>{x.B21, Oi, Yi + rl, rr, k, rl, x.ld21, ld, ld});              // Sr_r -= B21 * Om~_l
        gemms.push_back(GemmProb<T>{B21T[i], Oi + rl + ck, Yi + ck, rl, k, rr, x.ld12, ld, ld});  // Sc_l -= B21^T * Ps~_r
        gemms.push_back(GemmProb<T>{B12T[i], Oi + ck, Yi + rl + ck, rr, k, rl, x.ld21, ld, ld});  // Sc_r -= B12^T * Ps~_l
      }
    }
    gather_A();
    vlap("entry blocks (D / B12, B21)", lv);
    run_gemms(tmp, gemms, 1, s);
    // ---- b. row IDs of the sample blocks [Sr | Sc] (read in place: without Z the ID leaves its input alone) -----------------
    std::vector<LowRank<T>> lr(nj);
    std::vector<LowRankJob<T>> jobs(nj);
    for (int a = 0; a < nj; ++a) {
      const int i = L[a];
      // sketch width k: the adaptive rule below keeps every rank under k - pad (a block that gets closer is redone at 2k by the ID itself)
      jobs[a] = LowRankJob<T>{Yl[i], ldl[i], nd[i].m, k2, k, cb[nblk[i]].seed + 7919ull * (uint64_t)(i - cb[nblk[i]].root), &lr[a], 0};
    }
    run_rows(tmp, rows, s);
    // deeper levels are truncated more tightly: the error they hand up must stay below the threshold of the levels above
    // (with one tolerance everywhere the sample blocks of the upper levels sit on a noise plateau AT the threshold and the
    // pivoted-LU rank detection reads it as rank: measured 45 instead of 18 at 1e-8 on the test kernel)
    const double lsc = std::pow(H.opt.level_scale, lv - 1);
    // the relative tolerance refers to the block's own largest pivot, but never to less than the largest one met below: a block
    // that couples weakly (or not at all) is noise of the children's truncation, and relative to ITSELF noise has full rank
    double gscale = 0.0;  // (the tolerance only matters to the LU-based rank rule, HS_HSS_QR=0: the largest scale of the batch)
    for (auto& B : cb) gscale = std::max(gscale, B.gscale);
    const bool norm_order = qr_norm_order();  // HS_QR_ORDER=norm: no pivoted LU of a sketch; qr_refine picks its windows by residual norms
    std::vector<int*> porder(nj, nullptr);
    int st = 0;
    if (norm_order) {
      for (int a = 0; a < nj; ++a) porder[a] = tmp.get<int>((size_t)std::max(nd[L[a]].m, 1));
    } else {
      st = lowrank_compress_batch<T>(jobs.data(), nj, std::max(std::max(H.opt.atol, H.opt.rtol * gscale) * lsc, noise_max), H.opt.rtol * lsc, s, false);
      for (int a = 0; a < nj; ++a) porder[a] = lr[a].rperm;
    }
    for (int a = 0; a < nj; ++a) cb[nblk[L[a]]].gscale = std::max(cb[nblk[L[a]]].gscale, lr[a].top);
    vlap("pivot order (tournament LU of sketches)", lv);
    auto free_lr = [&]() {
      for (auto& q : lr) lowrank_free(q);
    };
    if (st != 0) {
      free_lr();
      throw st;
    }
    // ---- c. skeletons, interpolation matrices: rank and least-squares T from the orthogonalisation of the rows in pivot order -------
    static const bool qr_on = !(getenv("HS_HSS_QR") && getenv("HS_HSS_QR")[0] == '0');  // diagnostics: 0 = the LU-based rule and T = L21*L11^-1
    std::vector<QrJob<T>> qj;
    if (qr_on) {
      qj.resize(nj);
      for (int a = 0; a < nj; ++a) {
        const int i = L[a];
        qj[a].M = Yl[i];
        qj[a].ldm = ldl[i];
        qj[a].m = nd[i].m;
        qj[a].q = k2;
        qj[a].p = porder[a];
        qj[a].norm_select = norm_order;
        qj[a].rmax = norm_order ? std::min(nd[i].m, k) : lr[a].k;  // the pivoted LU ordered the first k (sketch width) rows; the orthogonalisation stops by itself once a block of 32 rows is below the threshold
        qj[a].atol_scale = std::max(H.opt.atol, H.opt.rtol * cb[nblk[i]].gscale_q);  // absolute threshold of ITS matrix (times lsc below)
        qj[a].floor_abs = noise[nblk[i]];
      }
      try {
        qr_refine<T>(tmp, H.keep, qj, lsc, H.opt.rtol * lsc, 0.0, s);
      } catch (...) {
        free_lr();
        throw;
      }
      for (int a = 0; a < nj; ++a) cb[nblk[L[a]]].gscale_q = std::max(cb[nblk[L[a]]].gscale_q, qj[a].top);
      vlap("rank + interpolation (qr_refine)", lv);
    }
    bool enough = true;
    for (int a = 0; a < nj; ++a) {
      const int m = nd[L[a]].m;
      int r = qr_on ? qj[a].r : lr[a].r;
      if (r < 1 && m > 0) r = 1;  // keep one skeleton position: every later shape stays non-empty
      // a sketch of k samples is trusted up to rank 0.8*k - pad: the spectra of separator blocks decay slowly, and a rank within a few
      // per cent of k means the tail beyond the sketch was never seen (measured on the 32,768 root of Poisson 128^3: rank 2,016 of 2,048
      // samples left a residual of 0.5 where 4,096 samples give 2e-3)
      if (r > (int)(0.8 * k) - (int)H.opt.pad && r < m && k < cb[nblk[L[a]]].n) enough = false;
      nd[L[a]].r = r;
    }
    static const bool verbose = getenv("HS_HSS_VERBOSE") != nullptr;
    if (verbose) {
      int mr = 0, mm = 0, full = 0, mlu = 0;
      long long sr = 0;
      for (int a = 0; a < nj; ++a) {
        const int i = L[a];
        mr = std::max(mr, nd[i].r);
        mm = std::max(mm, nd[i].m);
        mlu = std::max(mlu, lr[a].r);
        sr += nd[i].r;
        full += nd[i].r == nd[i].m;
      }
      fprintf(stderr, "[hs hss] n=%d k=%d level %d: %d nodes, local size <= %d, rank max %d mean %.1f (pivoted-LU estimate <= %d), %d nodes of full rank%s\n", n, k, lv,
              nj, mm, mr, (double)sr / nj, mlu, full, enough ? "" : "  -> more samples");
    }
    if (!enough) {
      HSS_HIP(hipStreamSynchronize(s));
      free_lr();
      return false;
    }
    std::vector<IdxJob> ij;
    int maxR = 0, maxr = 0;
    try {
      for (int a = 0; a < nj; ++a) {
        const int i = L[a];
        HNode<T>& x = nd[i];
        const int m = x.m, r = x.r, nR = m - r;
        x.p = H.keep.template get<int>(m);
        HSS_HIP(hipMemcpyAsync(x.p, porder[a], sizeof(int) * m, hipMemcpyDeviceToDevice, s));
        x.sk = H.keep.template get<int>(r);
        x.ldt = ev(nR);
        x.ldtt = ev(r);
        x.Tt = H.keep.template get<T>((size_t)x.ldtt * std::max(nR, 1));
        if (qr_on && qj[a].r >= 1) {
          x.Tm = qj[a].Tm;  // least-squares interpolation from qr_refine (allocated from H.keep, ld = ev(nR))
        } else {
          x.Tm = H.keep.template get<T>((size_t)x.ldt * r);
          if (!qr_on && lr[a].r >= 1)
            subs.push_back(SubJob<T>{lr[a].Lp, lr[a].ldp, nullptr, nullptr, r, 0, nR, r, x.Tm, x.ldt, 0});  // T <- L21
          else  // the sample block is zero (the node does not couple to the rest at all): one nominal skeleton position, T = 0 --
            HSS_HIP(hipMemsetAsync(x.Tm, 0, sizeof(T) * (size_t)x.ldt * r, s));  // the L\U of a zero sketch holds nothing usable
        }
        {
          const CBlock<T>& B = cb[nblk[i]];
          const int lo_loc = x.lo - B.off;  // skeleton indices are local to the node's matrix
          ij.push_back(IdxJob{x.p, (x.left < 0 && B.perm) ? B.perm + lo_loc : Jidx[i], lo_loc, r, x.sk});
        }
        maxR = std::max(maxR, nR);
        maxr = std::max(maxr, r);
      }
      run_subs(tmp, subs, s);
      if (maxR > 0 && !qr_on) {
        // T = L21 * L11^-1, 32 columns at a time from the right; the descriptors of every step are uploaded once
        std::vector<GemmProb<T>> gp;
        std::vector<TsJob<T>> ts;
        struct Step {
          size_t g0, gn, t0, tn;
          int mN;
        };
        std::vector<Step> steps;
        for (int j0 = (maxr - 1) / 32 * 32; j0 >= 0; j0 -= 32) {
          Step st{gp.size(), 0, ts.size(), 0, 0};
          for (int a = 0; a < nj; ++a) {
            const HNode<T>& x = nd[L[a]];
            const int r = x.r, nR = x.m - r;
            if (nR <= 0 || r <= j0 || lr[a].r < 1) continue;
            const int j1 = std::min(j0 + 32, r);
            if (r > j1) {
              gp.push_back(GemmProb<T>{x.Tm + (size_t)j1 * x.ldt, lr[a].Lp + j1 + (size_t)j0 * lr[a].ldp, x.Tm + (size_t)j0 * x.ldt, nR, j1 - j0, r - j1, x.ldt,
                                       lr[a].ldp, x.ldt});
              st.mN = std::max(st.mN, j1 - j0);
            }
            ts.push_back(TsJob<T>{lr[a].Lp, lr[a].ldp, x.m, r, x.Tm, x.ldt, j0, j1});
          }
          st.gn = gp.size() - st.g0;
          st.tn = ts.size() - st.t0;
          steps.push_back(st);
        }
        GemmProb<T>* dgp = upload(tmp, gp, s);
        TsJob<T>* dts = upload(tmp, ts, s);
        for (const Step& st : steps) {
          if (st.gn > 0) launch_gemm_probs<T>(dgp + st.g0, (int)st.gn, maxR, st.mN, 1, s);
          if (st.tn > 0) hipLaunchKernelGGL(tsolve_block_kernel<T>, dim3((maxR + 63) / 64, (unsigned)st.tn), dim3(64), 0, s, (const TsJob<T>*)(dts + st.t0));
        }
      }
      if (maxR > 0) {
        for (int a = 0; a < nj; ++a) {
          const HNode<T>& x = nd[L[a]];
          subs.push_back(SubJob<T>{x.Tm, x.ldt, nullptr, nullptr, 0, 0, x.m - x.r, x.r, x.Tt, x.ldtt, 1});  // T^T
        }
        run_subs(tmp, subs, s);
      }
      if (verbose) {  // growth of the interpolation matrices (an LU-based ID bounds |L21| <= 1, not |L21 * L11^-1|)
        HSS_HIP(hipStreamSynchronize(s));
        double tmax = 0.0;
        for (int i : L) {
          const HNode<T>& x = nd[i];
          const int nR = x.m - x.r;
          if (nR <= 0 || x.r <= 0) continue;
          std::vector<T> ht((size_t)x.ldt * x.r);
          HSS_HIP(hipMemcpy(ht.data(), x.Tm, sizeof(T) * ht.size(), hipMemcpyDeviceToHost));
          for (int c = 0; c < x.r; ++c)
            for (int rr = 0; rr < nR; ++rr) tmax = std::max(tmax, Scal<T>::abs1(ht[(size_t)rr + (size_t)c * x.ldt]));
        }
        fprintf(stderr, "[hs hss]      level %d: max |T_ij| = %.3g\n", lv, tmax);
      }
      IdxJob* dij = upload(tmp, ij, s);
      hipLaunchKernelGGL(idx_compose_kernel, dim3((maxr + 63) / 64, (unsigned)ij.size()), dim3(64), 0, s, (const IdxJob*)dij);
      if (bop) {  // the matrix-free operator routes entry requests on the host: it needs the skeleton indices there
        for (int i : L) {
          nd[i].hsk.resize((size_t)nd[i].r);
          HSS_HIP(hipMemcpyAsync(nd[i].hsk.data(), nd[i].sk, sizeof(int) * (size_t)nd[i].r, hipMemcpyDeviceToHost, s));
        }
        HSS_HIP(hipStreamSynchronize(s));
      }
      // ---- d. hand the skeleton rows of the samples and the compressed test matrices to the parents ------------------------
      if (lv > 1) {
        std::vector<T*> GR(N, nullptr);
        for (int i : L) {
          const HNode<T>& x = nd[i];
          const int par = x.parent;
          if (nd[par].left == i) {  // allocate the parent's local blocks once both ranks are known
            const int mp = x.r + nd[nd[par].right].r;
            nd[par].m = mp;
            ldl[par] = ev(mp);
            Yl[par] = tmp.get<T>((size_t)ldl[par] * k2);
            Ol[par] = tmp.get<T>((size_t)ldl[par] * k2);
            Jidx[par] = tmp.get<int>(mp);
          }
        }
        for (int i : L) {
          HNode<T>& x = nd[i];
          const int par = x.parent, off = nd[par].left == i ? 0 : nd[nd[par].left].r;
          x.off_in_parent = off;
          const int m = x.m, r = x.r, nR = m - r;
          rows.push_back(RowJob<T>{Yl[i], ldl[i], Yl[par] + off, ldl[par], x.p, r, k2, ROW_GATHER});
          rows.push_back(RowJob<T>{Ol[i], ldl[i], Ol[par] + off, ldl[par], x.p, r, k2, ROW_GATHER});
          if (nR > 0) {
            GR[i] = tmp.get<T>((size_t)ev(nR) * k2);
            rows.push_back(RowJob<T>{Ol[i], ldl[i], GR[i], ev(nR), x.p + r, nR, k2, ROW_GATHER_NEG});
            gemms.push_back(GemmProb<T>{x.Tt, GR[i], Ol[par] + off, r, k2, nR, x.ldtt, ev(nR), ldl[par]});  // += T^T * (rows p_R)
          }
          HSS_HIP(hipMemcpyAsync(Jidx[par] + off, x.sk, sizeof(int) * r, hipMemcpyDeviceToDevice, s));
        }
        run_rows(tmp, rows, s);
        run_gemms(tmp, gemms, 1, s);
      } else {
        for (int i : L) nd[i].off_in_parent = nd[nd[i].parent].left == i ? 0 : nd[nd[nd[i].parent].left].r;
      }
      HSS_HIP(hipStreamSynchronize(s));
    } catch (...) {
      (void)hipStreamSynchronize(s);
      free_lr();
      throw;
    }
    free_lr();
    vlap("skeletons handed to the parents", lv);
  }
  // roots: couplings of their two children
  for (int b = 0; b < F; ++b) {
    HNode<T>& x = nd[cb[b].root];
    const HNode<T>&l = nd[x.left], &r = nd[x.right];
    x.m = l.r + r.r;
    x.ld12 = ev(l.r);
    x.ld21 = ev(r.r);
    x.B12 = H.keep.template get<T>((size_t)x.ld12 * r.r);
    x.B21 = H.keep.template get<T>((size_t)x.ld21 * l.r);
    subs.push_back(SubJob<T>{cb[b].A, cb[b].lda, l.sk, r.sk, 0, 0, l.r, r.r, x.B12, x.ld12, 0, l.hsk.data(), r.hsk.data(), b});
    subs.push_back(SubJob<T>{cb[b].A, cb[b].lda, r.sk, l.sk, 0, 0, r.r, l.r, x.B21, x.ld21, 0, r.hsk.data(), l.hsk.data(), b});
  }
  gather_A();
  HSS_HIP(hipStreamSynchronize(s));
  return true;
}

// ------------------------------------------------------------------------------------------------
// Y = H * X
// ------------------------------------------------------------------------------------------------
template <class T>
void hss_mul_p(HssT<T>& H, const T* X, int ldx, T* Y, int ldy, int q, bool trans);
// Y = H * X (trans: H^T * X, plain transpose) in the caller's index order: the tree works on the permuted vectors
template <class T>
void hss_mul(HssT<T>& H, const T* X, int ldx, T* Y, int ldy, int q, bool trans = false) {
  if (!H.perm) {
    hss_mul_p(H, X, ldx, Y, ldy, q, trans);
    return;
  }
  Pool tmp(global_cache());
  const int ld = ev(H.n);
  T* Xp = tmp.get<T>((size_t)ld * q);
  T* Yp = tmp.get<T>((size_t)ld * q);
  std::vector<RowJob<T>> rows{RowJob<T>{X, ldx, Xp, ld, H.perm, H.n, q, ROW_GATHER}};
  run_rows(tmp, rows, H.s);
  hss_mul_p(H, Xp, ld, Yp, ld, q, trans);
  rows.push_back(RowJob<T>{Yp, ld, Y, ldy, H.perm, H.n, q, ROW_SCATTER});
  run_rows(tmp, rows, H.s);
  HSS_HIP(hipStreamSynchronize(H.s));
}
template <class T>
void hss_mul_p(HssT<T>& H, const T* X, int ldx, T* Y, int ldy, int q, bool trans) {
  hipStream_t s = H.s;
  auto& nd = H.nd;
  const int N = (int)nd.size();
  Pool tmp(global_cache());
  std::vector<RowJob<T>> rows;
  std::vector<GemmProb<T>> gemms;
  if (trans) {  // H^T has the same bases (U = V) and the generators D^T, B12 <- B21^T, B21 <- B12^T: transposed copies on first use
    std::vector<SubJob<T>> tj;
    for (auto& x : nd) {
      if (x.left < 0) {
        if (!x.DTt && x.D) {
          x.DTt = H.keep.template get<T>((size_t)x.ldd * x.m);
          tj.push_back(SubJob<T>{x.D, x.ldd, nullptr, nullptr, 0, 0, x.m, x.m, x.DTt, x.ldd, 1});
        }
      } else if (!x.B12t) {
        const int rl = nd[x.left].r, rr = nd[x.right].r;
        x.B12t = H.keep.template get<T>((size_t)ev(rr) * rl);
        x.B21t = H.keep.template get<T>((size_t)ev(rl) * rr);
        tj.push_back(SubJob<T>{x.B12, x.ld12, nullptr, nullptr, 0, 0, rl, rr, x.B12t, ev(rr), 1});
        tj.push_back(SubJob<T>{x.B21, x.ld21, nullptr, nullptr, 0, 0, rr, rl, x.B21t, ev(rl), 1});
      }
    }
    run_subs(tmp, tj, s);
  }
  if (nd[0].left < 0) {
    HSS_HIP(hipMemset2DAsync(Y, sizeof(T) * ldy, 0, sizeof(T) * H.n, q, s));
    gemms.push_back(GemmProb<T>{trans ? nd[0].DTt : nd[0].D, X, Y, H.n, q, H.n, nd[0].ldd, ldx, ldy});
    run_gemms(tmp, gemms, 0, s);
    HSS_HIP(hipStreamSynchronize(s));
    return;
  }
  // upward pass: XT[i] = [x~_left; x~_right] of every inner node
  std::vector<T*> XT(N, nullptr), G(N, nullptr), TD(N, nullptr);
  std::vector<int> ldx_(N, 0);
  {  // everything that must start from zero comes out of ONE block cleared by ONE memset: the products of a tree level are grouped
     // launches, but a memset per node made an application of the matrix hundreds of launches (expanding a 32,768 block from its
     // generators, 32 applications: 50,000 memsets)
    size_t zel = 0;
    for (int i = 0; i < N; ++i) {
      if (nd[i].left >= 0) zel += (size_t)ev(nd[i].m) * q;
      if (i != 0 && nd[i].m - nd[i].r > 0) zel += (size_t)ev(nd[i].m - nd[i].r) * q;
    }
    T* z = tmp.get<T>(zel + 2);
    HSS_HIP(hipMemsetAsync(z, 0, sizeof(T) * zel, s));
    for (int i = 0; i < N; ++i) {
      if (nd[i].left >= 0) {
        ldx_[i] = ev(nd[i].m);
        XT[i] = tmp.get<T>((size_t)ldx_[i] * q);
        G[i] = z;
        z += (size_t)ldx_[i] * q;
      }
      if (i != 0 && nd[i].m - nd[i].r > 0) {
        TD[i] = z;
        z += (size_t)ev(nd[i].m - nd[i].r) * q;
      }
    }
    HSS_HIP(hipMemset2DAsync(Y, sizeof(T) * ldy, 0, sizeof(T) * H.n, q, s));  // the leaves' output blocks tile Y
  }
  for (int lv = H.nlev - 1; lv >= 1; --lv) {
    for (int i : H.lev[lv]) {
      const HNode<T>& x = nd[i];
      const int par = x.parent, r = x.r, nR = x.m - r;
      const T* src = x.left < 0 ? X + x.lo : XT[i];
      const int lds = x.left < 0 ? ldx : ldx_[i];
      T* slot = XT[par] + x.off_in_parent;
      rows.push_back(RowJob<T>{src, lds, slot, ldx_[par], x.p, r, q, ROW_GATHER});
      if (nR > 0) {
        T* t = tmp.get<T>((size_t)ev(nR) * q);
        rows.push_back(RowJob<T>{src, lds, t, ev(nR), x.p + r, nR, q, ROW_GATHER_NEG});
        gemms.push_back(GemmProb<T>{x.Tt, t, slot, r, q, nR, x.ldtt, ev(nR), ldx_[par]});
      }
    }
    run_rows(tmp, rows, s);
    run_gemms(tmp, gemms, 1, s);
  }
  // downward pass
  for (int lv = 0; lv < H.nlev; ++lv) {
    std::vector<RowJob<T>> adds;
    for (int i : H.lev[lv]) {
      const HNode<T>& x = nd[i];
      T* blk;  // the node's local output block (m rows)
      int ldb;
      if (x.left < 0) {
        blk = Y + x.lo;
        ldb = ldy;
        gemms.push_back(GemmProb<T>{trans ? x.DTt : x.D, X + x.lo, blk, x.m, q, x.m, x.ldd, ldx, ldy});
      } else {
        blk = G[i];
        ldb = ldx_[i];
        const int rl = nd[x.left].r, rr = nd[x.right].r;
        if (!trans) {
          gemms.push_back(GemmProb<T>{x.B12, XT[i] + rl, blk, rl, q, rr, x.ld12, ldx_[i], ldb});
          gemms.push_back(GemmProb<T>{x.B21, XT[i], blk + rl, rr, q, rl, x.ld21, ldx_[i], ldb});
        } else {
          gemms.push_back(GemmProb<T>{x.B21t, XT[i] + rl, blk, rl, q, rr, ev(rl), ldx_[i], ldb});
          gemms.push_back(GemmProb<T>{x.B12t, XT[i], blk + rl, rr, q, rl, ev(rr), ldx_[i], ldb});
        }
      }
      if (i != 0) {  // the contribution that arrives from above: U_i * g
        const int par = x.parent, r = x.r, nR = x.m - r;
        const T* g = G[par] + x.off_in_parent;
        adds.push_back(RowJob<T>{g, ldx_[par], blk, ldb, x.p, r, q, ROW_SCATTER_ADD});
        if (nR > 0) {
          T* t = TD[i];
          gemms.push_back(GemmProb<T>{x.Tm, g, t, nR, q, r, x.ldt, ldx_[par], ev(nR)});
          adds.push_back(RowJob<T>{t, ev(nR), blk, ldb, x.p + r, nR, q, ROW_SCATTER_ADD});
        }
      }
    }
    run_gemms(tmp, gemms, 0, s);
    run_rows(tmp, adds, s);
  }
  HSS_HIP(hipStreamSynchronize(s));
}

#include "hs_hss_op.h"

// ------------------------------------------------------------------------------------------------
// entries: out = H[I, J] for index lists (the `getindex` an operator assembled from HSS blocks is asked for,
// src/factorization.jl:129-137,246-249).  O((|I| + |J|) * r) per tree level: the basis rows of the requested
// indices are carried up the tree -- as rows E (I side) and as columns F = E^T (J side) -- and every inner node
// contributes E_l * B12 * F_r and E_r * B21 * F_l to the block of its two index ranges; leaves contribute D[I, J].
// One pair here; hss_getindex_batch (hs_hss_op.h) serves many pairs with one group of launches per tree level.
// ------------------------------------------------------------------------------------------------
template <class T>
void hss_getindex(HssT<T>& H, const int64_t* I, int ni, const int64_t* J, int nj, T* out, int ldo) {
  if (ni <= 0 || nj <= 0) return;
  std::vector<int> hi((size_t)ni), hj((size_t)nj);
  for (int a = 0; a < ni; ++a) {
    if (I[a] < 0 || I[a] >= H.n) {
      hs_set_error(HS_ERR_ARGUMENT, a, "BoundsError: index %lld outside 0:%d", (long long)I[a], H.n - 1);
      throw (int)HS_ERR_ARGUMENT;
    }
    hi[(size_t)a] = (int)I[a];
  }
  for (int a = 0; a < nj; ++a) {
    if (J[a] < 0 || J[a] >= H.n) {
      hs_set_error(HS_ERR_ARGUMENT, a, "BoundsError: index %lld outside 0:%d", (long long)J[a], H.n - 1);
      throw (int)HS_ERR_ARGUMENT;
    }
    hj[(size_t)a] = (int)J[a];
  }
  std::vector<GiJob<T>> one{GiJob<T>{hi.data(), ni, hj.data(), nj, out, ldo}};
  hss_getindex_batch<T>(H, one);
}

// ------------------------------------------------------------------------------------------------
// expanded basis of one node: out (size(node) x r) = U_big, A(I_node, far) ~= U_big * A(sk_node, far) -- the `generators(S.A11)` /
// `U*B12`, `V` factors a parent front builds its low-rank couplings from (src/factorization.jl:129-137).  Bottom-up over the subtree:
// leaf U = P^T [I; T]; inner node U = blkdiag(U_left, U_right) * P^T [I; T].
// ------------------------------------------------------------------------------------------------
// expanded bases E[i] (rows of node i x r_i) of every node of the subtree under `node` (node == 0: of every node but the root), level by
// level from the leaves with one group of launches per level; the basis of `node` itself is written to out when node > 0
template <class T>
static void hss_bases_impl(HssT<T>& H, Pool& tmp, int node, T* out, int ldo, std::vector<T*>& E, std::vector<int>& lde) {
  hipStream_t s = H.s;
  auto& nd = H.nd;
  const int N = (int)nd.size();
  // the subtree, by level
  std::vector<std::vector<int>> lv(H.nlev);
  std::vector<int> cur{node};
  while (!cur.empty()) {
    std::vector<int> nxt;
    for (int i : cur) {
      if (i != 0) lv[nd[i].level].push_back(i);
      if (nd[i].left >= 0) {
        nxt.push_back(nd[i].left);
        nxt.push_back(nd[i].right);
      }
    }
    cur.swap(nxt);
  }
  E.assign(N, nullptr);
  lde.assign(N, 0);
  auto target = [&](int i) {  // the requested node writes straight into `out`
    if (i == node && node > 0) {
      E[i] = out;
      lde[i] = ldo;
    } else {
      lde[i] = ev(nd[i].hi - nd[i].lo);
      E[i] = tmp.get<T>((size_t)lde[i] * std::max(nd[i].r, 1));
    }
  };
  for (int l = H.nlev - 1; l >= std::max(nd[node].level, 1); --l) {
    std::vector<BasisJob<T>> bj;
    std::vector<SubJob<T>> blocks, cg;
    std::vector<RowJob<T>> neg;
    std::vector<GemmProb<T>> ge;
    int maxcnt = 0, maxr = 0;
    for (int i : lv[l]) {
      HNode<T>& x = nd[i];
      const int cnt = x.hi - x.lo, rk = x.r, nR = x.m - rk;
      target(i);
      if (x.left < 0) {
        if (x.hinvp.empty()) {
          std::vector<int> hp(x.m);
          HSS_HIP(hipMemcpyAsync(hp.data(), x.p, sizeof(int) * x.m, hipMemcpyDeviceToHost, s));
          HSS_HIP(hipStreamSynchronize(s));
          x.hinvp.assign(x.m, 0);
          for (int a = 0; a < x.m; ++a) x.hinvp[hp[a]] = a;
        }
        int* dip = upload(tmp, x.hinvp, s);
        bj.push_back(BasisJob<T>{x.Tm, x.ldt, rk, cnt, dip, E[i], lde[i], 0});
        maxcnt = std::max(maxcnt, cnt);
        maxr = std::max(maxr, rk);
        continue;
      }
      const int cl = nd[x.left].hi - nd[x.left].lo, cr = cnt - cl, rl = nd[x.left].r, rr = nd[x.right].r;
      const int ldw = ev(cnt);
      T* W = tmp.getz<T>((size_t)ldw * x.m, s);
      blocks.push_back(SubJob<T>{E[x.left], lde[x.left], nullptr, nullptr, 0, 0, cl, rl, W, ldw, 0});
      blocks.push_back(SubJob<T>{E[x.right], lde[x.right], nullptr, nullptr, 0, 0, cr, rr, W + cl + (size_t)ldw * rl, ldw, 0});
      cg.push_back(SubJob<T>{W, ldw, nullptr, x.p, 0, 0, cnt, rk, E[i], lde[i], 0});
      if (nR > 0) {
        if (!x.NTm) {
          x.NTm = H.keep.template get<T>((size_t)x.ldt * rk);
          neg.push_back(RowJob<T>{x.Tm, x.ldt, x.NTm, x.ldt, nullptr, nR, rk, ROW_GATHER_NEG});
        }
        T* Wr = tmp.get<T>((size_t)ldw * nR);
        cg.push_back(SubJob<T>{W, ldw, nullptr, x.p + rk, 0, 0, cnt, nR, Wr, ldw, 0});
        ge.push_back(GemmProb<T>{Wr, x.NTm, E[i], cnt, rk, nR, ldw, x.ldt, lde[i]});  // E += W[:, p_R] * T
      }
    }
    if (!bj.empty()) {
      BasisJob<T>* dj = upload(tmp, bj, s);
      hipLaunchKernelGGL(basis_rows_kernel<T>, dim3((maxcnt + 63) / 64, (maxr + 15) / 16, (unsigned)bj.size()), dim3(64), 0, s, (const BasisJob<T>*)dj);
    }
    run_rows(tmp, neg, s);
    run_subs(tmp, blocks, s);
    run_subs(tmp, cg, s);
    run_gemms(tmp, ge, 1, s);
  }
}
template <class T>
void hss_basis(HssT<T>& H, int node, T* out, int ldo) {
  const int N = (int)H.nd.size();
  if (node <= 0 || node >= N) {
    hs_set_error(HS_ERR_ARGUMENT, node, "ArgumentError: HSS node %d has no basis (the root has none)", node);
    throw (int)HS_ERR_ARGUMENT;
  }
  Pool tmp(global_cache());
  std::vector<T*> E;
  std::vector<int> lde;
  hss_bases_impl<T>(H, tmp, node, out, ldo, E, lde);
  HSS_HIP(hipStreamSynchronize(H.s));
}

// `Matrix(H)` / `full(H)` of HssMatrices.jl: out (n x n, device) = the matrix H represents, in H's own index order (for a compressed matrix
// A[perm, perm]; the block views of hs_hss_child carry no permutation).  Every node's expanded basis once (O(n r^2) per level), then per
// inner node the two products U_l*B12*U_r^T and U_r*B21*U_l^T written into their place and the leaves' D blocks copied: 2 n^2 r flops,
// a tenth of applying H to the identity.
template <class T>
void hss_expand(HssT<T>& H, T* out, int ldo) {
  hipStream_t s = H.s;
  auto& nd = H.nd;
  const int N = (int)nd.size();
  Pool tmp(global_cache());
  HSS_HIP(hipMemset2DAsync(out, sizeof(T) * ldo, 0, sizeof(T) * H.n, H.n, s));
  std::vector<SubJob<T>> copies, tr;
  std::vector<GemmProb<T>> g1, g2;
  if (nd[0].left < 0) {
    copies.push_back(SubJob<T>{nd[0].D, nd[0].ldd, nullptr, nullptr, 0, 0, H.n, H.n, out, ldo, 0});
    run_subs(tmp, copies, s);
    HSS_HIP(hipStreamSynchronize(s));
    return;
  }
  std::vector<T*> E;
  std::vector<int> lde;
  hss_bases_impl<T>(H, tmp, 0, nullptr, 0, E, lde);
  std::vector<T*> ET(N, nullptr);  // E_i^T (r_i x rows): the right factor of a sibling's block
  for (int i = 1; i < N; ++i) {
    const int cnt = nd[i].hi - nd[i].lo, r = nd[i].r;
    if (r <= 0) continue;
    ET[i] = tmp.get<T>((size_t)ev(r) * cnt);
    tr.push_back(SubJob<T>{E[i], lde[i], nullptr, nullptr, 0, 0, cnt, r, ET[i], ev(r), 1});
  }
  run_subs(tmp, tr, s);
  for (int i = 0; i < N; ++i) {
    const HNode<T>& x = nd[i];
    if (x.left < 0) {
      copies.push_back(SubJob<T>{x.D, x.ldd, nullptr, nullptr, 0, 0, x.m, x.m, out + x.lo + (size_t)x.lo * ldo, ldo, 0});
      continue;
    }
    const HNode<T>&l = nd[x.left], &r = nd[x.right];
    const int cl = l.hi - l.lo, cr = r.hi - r.lo, rl = l.r, rr = r.r;
    if (rl <= 0 || rr <= 0) continue;
    T* t12 = tmp.getz<T>((size_t)ev(cl) * rr, s);
    T* t21 = tmp.getz<T>((size_t)ev(cr) * rl, s);
    g1.push_back(GemmProb<T>{E[x.left], x.B12, t12, cl, rr, rl, lde[x.left], x.ld12, ev(cl)});
    g1.push_back(GemmProb<T>{E[x.right], x.B21, t21, cr, rl, rr, lde[x.right], x.ld21, ev(cr)});
    g2.push_back(GemmProb<T>{t12, ET[x.right], out + l.lo + (size_t)r.lo * ldo, cl, cr, rr, ev(cl), ev(rr), ldo});
    g2.push_back(GemmProb<T>{t21, ET[x.left], out + r.lo + (size_t)l.lo * ldo, cr, cl, rl, ev(cr), ev(rl), ldo});
  }
  run_subs(tmp, copies, s);
  run_gemms(tmp, g1, 0, s);
  run_gemms(tmp, g2, 0, s);
  HSS_HIP(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------------------------------------
// elimination
// ------------------------------------------------------------------------------------------------
template <class T>
void alloc_front(Pool& pool, NodeDesc<T>& d, int ni, int nb, int node, hipStream_t s) {
  memset(&d, 0, sizeof d);
  const int m = ni + nb, nblk = (ni + HS_PB - 1) / HS_PB, ncand = ((ni + 127) / 128 + 1) * HS_PB;
  d.ldl = ev(m);
  d.ldu = ev(ni);
  d.lds = ev(nb);
  d.LF = pool.get<T>((size_t)d.ldl * std::max(ni, 1));
  d.UR = pool.get<T>((size_t)d.ldu * std::max(nb, 1));
  d.SB = pool.get<T>((size_t)d.lds * std::max(nb, 1));
  d.invL = pool.get<T>((size_t)2 * std::max(nblk, 1) * HS_PB * HS_PB);
  d.invU = d.invL + (size_t)std::max(nblk, 1) * HS_PB * HS_PB;
  int* ints = pool.get<int>((size_t)2 * ni + 2 * ncand + HS_PB + 8);
  // (on the caller's stream: a synchronous memset runs on the null stream, which waits for -- and holds up -- every blocking stream of the process,
  // i.e. the other fronts that are being compressed on their own host threads)
  HSS_HIP(hipMemsetAsync(ints, 0, sizeof(int) * ((size_t)2 * ni + 2 * ncand + HS_PB + 8), s));
  d.ipiv = ints;
  d.rperm = d.ipiv + ni;
  d.cand0 = d.rperm + ni;
  d.cand1 = d.cand0 + ncand;
  d.pivlist = d.cand1 + ncand;
  d.info = d.pivlist + HS_PB;
  d.ni = ni; d.nb = nb; d.m = m;
  d.ni1 = ni; d.nb1 = nb; d.isleaf = 1; d.node = node;
  d.finalize();
}

template <class T>
void factor_batch(Pool& tmp, std::vector<NodeDesc<T>>& hd, hipStream_t s) {
  if (hd.empty()) return;
  NodeDesc<T>* dn = upload(tmp, hd, s);
  int maxni = 0, maxnb = 0, maxm = 0;
  for (auto& d : hd) {
    maxni = std::max(maxni, d.ni);
    maxnb = std::max(maxnb, d.nb);
    maxm = std::max(maxm, d.m);
  }
  Profiler prof;
  launch_init_fronts<T>(dn, (int)hd.size(), maxni, s);
  Sched<T> sch{dn, (int)hd.size(), maxni, maxnb, maxm, s, &prof, nullptr, nullptr};
  sch.factor_fronts();
}

template <class T>
void hss_factor(HssT<T>& H) {
  if (H.factored) return;
  hipStream_t s = H.s;
  auto& nd = H.nd;
  const int N = (int)nd.size();
  Pool tmp(global_cache());
  std::vector<SubJob<T>> subs;
  std::vector<GemmProb<T>> gemms;
  auto t0 = std::chrono::steady_clock::now();
  auto finish_root = [&](const T* M, int ldm, int m) {
    alloc_front(H.keep, H.rootfd, m, 0, 0, s);
    subs.push_back(SubJob<T>{M, ldm, nullptr, nullptr, 0, 0, m, m, H.rootfd.LF, H.rootfd.ldl, 0});
    run_subs(tmp, subs, s);
    std::vector<NodeDesc<T>> one{H.rootfd};
    factor_batch(tmp, one, s);
    H.root_m = m;
  };
  if (nd[0].left < 0) {
    finish_root(nd[0].D, nd[0].ldd, H.n);
  } else {
    std::vector<T*> M(N, nullptr);
    std::vector<int> ldm(N, 0);
    // local matrix of an inner node: [S^_l B12; B21 S^_r]
    auto build_M = [&](int i) {
      const HNode<T>& x = nd[i];
      const HNode<T>&l = nd[x.left], &r = nd[x.right];
      ldm[i] = ev(x.m);
      M[i] = tmp.get<T>((size_t)ldm[i] * x.m);
      subs.push_back(SubJob<T>{l.fd.SB, l.fd.lds, nullptr, nullptr, 0, 0, l.r, l.r, M[i], ldm[i], 0});
      subs.push_back(SubJob<T>{r.fd.SB, r.fd.lds, nullptr, nullptr, 0, 0, r.r, r.r, M[i] + l.r + (size_t)ldm[i] * l.r, ldm[i], 0});
      subs.push_back(SubJob<T>{x.B12, x.ld12, nullptr, nullptr, 0, 0, l.r, r.r, M[i] + (size_t)ldm[i] * l.r, ldm[i], 0});
      subs.push_back(SubJob<T>{x.B21, x.ld21, nullptr, nullptr, 0, 0, r.r, l.r, M[i] + l.r, ldm[i], 0});
    };
    for (int lv = H.nlev - 1; lv >= 1; --lv) {
      const std::vector<int>& L = H.lev[lv];
      for (int i : L)
        if (nd[i].left >= 0) build_M(i);
      run_subs(tmp, subs, s);
      std::vector<NodeDesc<T>> batch;
      std::vector<GemmProb<T>> g2;
      for (int i : L) {
        HNode<T>& x = nd[i];
        const int m = x.m, r = x.r, nR = m - r;
        alloc_front(H.keep, x.fd, nR, r, i, s);
        x.has_front = nR > 0;
        const T* Ms = x.left < 0 ? x.D : M[i];
        const int ld = x.left < 0 ? x.ldd : ldm[i];
        const int *pS = x.p, *pR = x.p + r;
        NodeDesc<T>& d = x.fd;
        subs.push_back(SubJob<T>{Ms, ld, pR, pR, 0, 0, nR, nR, d.LF, d.ldl, 0});
        subs.push_back(SubJob<T>{Ms, ld, pS, pR, 0, 0, r, nR, d.LF + nR, d.ldl, 0});
        subs.push_back(SubJob<T>{Ms, ld, pR, pS, 0, 0, nR, r, d.UR, d.ldu, 0});
        subs.push_back(SubJob<T>{Ms, ld, pS, pS, 0, 0, r, r, d.SB, d.lds, 0});
        if (nR > 0) {
          // X = E M F with E = [I -T; 0 I], F = [I 0; -T^T I]  (front order [R; S])
          gemms.push_back(GemmProb<T>{x.Tm, d.LF + nR, d.LF, nR, nR, r, x.ldt, d.ldl, d.ldl});  // M_RR - T M_SR
          gemms.push_back(GemmProb<T>{x.Tm, d.SB, d.UR, nR, r, r, x.ldt, d.lds, d.ldu});        // X_RS = M_RS - T M_SS
          g2.push_back(GemmProb<T>{d.UR, x.Tt, d.LF, nR, nR, r, d.ldu, x.ldtt, d.ldl});         // X_RR = ... - X_RS T^T
          g2.push_back(GemmProb<T>{d.SB, x.Tt, d.LF + nR, r, nR, r, d.lds, x.ldtt, d.ldl});     // X_SR = M_SR - M_SS T^T
          batch.push_back(d);
        }
      }
      run_subs(tmp, subs, s);
      run_gemms(tmp, gemms, 1, s);
      run_gemms(tmp, g2, 1, s);
      factor_batch(tmp, batch, s);
    }
    build_M(0);
    run_subs(tmp, subs, s);
    finish_root(M[0], ldm[0], nd[0].m);
  }
  HSS_HIP(hipStreamSynchronize(s));
  // exactly singular pivots (the reference's `\` would throw SingularException)
  std::vector<const NodeDesc<T>*> all{&H.rootfd};
  for (auto& x : nd)
    if (x.has_front) all.push_back(&x.fd);
  std::vector<int> infos(all.size(), 0);
  for (size_t a = 0; a < all.size(); ++a) HSS_HIP(hipMemcpyAsync(&infos[a], all[a]->info, sizeof(int), hipMemcpyDeviceToHost, s));
  HSS_HIP(hipStreamSynchronize(s));
  for (size_t a = 0; a < all.size(); ++a)
    if (infos[a] != 0) {
      hs_set_error(HS_ERR_SINGULAR, all[a]->node, "SingularException: HSS node %d hit an exactly zero pivot", all[a]->node);
      throw (int)HS_ERR_SINGULAR;
    }
  H.t_factor = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  H.factored = true;
}

// laswp + forward (and optionally backward) triangular solves of a batch of factored fronts on blocks of right-hand
// sides: the descriptors are the fronts' with UR redirected to the block
template <class T>
void tri_solve_batch(Pool& tmp, std::vector<NodeDesc<T>>& hd, int q, bool lower, bool upper, hipStream_t s) {
  if (hd.empty()) return;
  int maxni = 0;
  for (auto& d : hd) maxni = std::max(maxni, d.ni);
  NodeDesc<T>* dn = upload(tmp, hd, s);
  Profiler prof;
  Sched<T> sch{dn, (int)hd.size(), maxni, q, maxni, s, &prof, nullptr, nullptr};
  int P2 = HS_PB;
  while (P2 < maxni) P2 *= 2;
  if (lower) {
    sch.laswp(HS_MAT_UR, 0, HS_BIG, 0, P2);
    sch.trsm_rec(HS_MAT_UR, 0, P2, 0, HS_BIG);
  }
  if (upper) sch.utrsm_rec(HS_MAT_UR, 0, P2, 0, HS_BIG);
  hd.clear();
}
template <class T>
NodeDesc<T> rhs_desc(const NodeDesc<T>& f, T* B, int ldb, int q) {
  NodeDesc<T> d = f;
  d.UR = B;
  d.ldu = ldb;
  d.nb = q;
  d.nb1 = q;
  d.m = f.ni;
  d.finalize();
  return d;
}

template <class T>
void hss_ldiv_p(HssT<T>& H, T* B, int ldb, int q);
template <class T>
void hss_ldiv(HssT<T>& H, T* B, int ldb, int q) {
  if (!H.perm) {
    hss_ldiv_p(H, B, ldb, q);
    return;
  }
  Pool tmp(global_cache());
  const int ld = ev(H.n);
  T* Bp = tmp.get<T>((size_t)ld * q);
  std::vector<RowJob<T>> rows{RowJob<T>{B, ldb, Bp, ld, H.perm, H.n, q, ROW_GATHER}};
  run_rows(tmp, rows, H.s);
  hss_ldiv_p(H, Bp, ld, q);
  rows.push_back(RowJob<T>{Bp, ld, B, ldb, H.perm, H.n, q, ROW_SCATTER});
  run_rows(tmp, rows, H.s);
  HSS_HIP(hipStreamSynchronize(H.s));
}
template <class T>
void hss_ldiv_p(HssT<T>& H, T* B, int ldb, int q) {
  hipStream_t s = H.s;
  auto& nd = H.nd;
  const int N = (int)nd.size();
  Pool tmp(global_cache());
  std::vector<RowJob<T>> rows;
  std::vector<GemmProb<T>> gemms;
  std::vector<NodeDesc<T>> descs;
  if (nd[0].left < 0) {
    descs.push_back(rhs_desc(H.rootfd, B, ldb, q));
    tri_solve_batch(tmp, descs, q, true, true, s);
    HSS_HIP(hipStreamSynchronize(s));
    return;
  }
  std::vector<T*> BH(N, nullptr), YR(N, nullptr);
  std::vector<int> ldh(N, 0), ldy(N, 0);
  for (int i = 0; i < N; ++i) {
    if (nd[i].left >= 0) {
      ldh[i] = ev(nd[i].m);
      BH[i] = tmp.get<T>((size_t)ldh[i] * q);
    }
    if (i != 0) {
      ldy[i] = ev(nd[i].m - nd[i].r);
      YR[i] = tmp.get<T>((size_t)ldy[i] * q);
    }
  }
  // forward: leaves to root
  for (int lv = H.nlev - 1; lv >= 1; --lv) {
    std::vector<GemmProb<T>> g2;
    for (int i : H.lev[lv]) {
      const HNode<T>& x = nd[i];
      const int par = x.parent, r = x.r, nR = x.m - r;
      const T* src = x.left < 0 ? B + x.lo : BH[i];
      const int lds = x.left < 0 ? ldb : ldh[i];
      T* slot = BH[par] + x.off_in_parent;
      rows.push_back(RowJob<T>{src, lds, slot, ldh[par], x.p, r, q, ROW_GATHER});
      if (nR > 0) {
        rows.push_back(RowJob<T>{src, lds, YR[i], ldy[i], x.p + r, nR, q, ROW_GATHER});
        gemms.push_back(GemmProb<T>{x.Tm, slot, YR[i], nR, q, r, x.ldt, ldh[par], ldy[i]});  // b_R -= T b_S
        descs.push_back(rhs_desc(x.fd, YR[i], ldy[i], q));                                    // y = L^-1 P b_R
        g2.push_back(GemmProb<T>{x.fd.LF + nR, YR[i], slot, r, q, nR, x.fd.ldl, ldy[i], ldh[par]});  // b_S -= (X_SR U^-1) y
      }
    }
    run_rows(tmp, rows, s);
    run_gemms(tmp, gemms, 1, s);
    tri_solve_batch(tmp, descs, q, true, false, s);
    run_gemms(tmp, g2, 1, s);
  }
  descs.push_back(rhs_desc(H.rootfd, BH[0], ldh[0], q));
  tri_solve_batch(tmp, descs, q, true, true, s);
  // backward: root to leaves
  for (int lv = 1; lv < H.nlev; ++lv) {
    std::vector<GemmProb<T>> g2;
    for (int i : H.lev[lv]) {
      const HNode<T>& x = nd[i];
      const int par = x.parent, r = x.r, nR = x.m - r;
      T* slot = BH[par] + x.off_in_parent;
      T* dst = x.left < 0 ? B + x.lo : BH[i];
      const int ldd = x.left < 0 ? ldb : ldh[i];
      if (nR > 0) {
        gemms.push_back(GemmProb<T>{x.fd.UR, slot, YR[i], nR, q, r, x.fd.ldu, ldh[par], ldy[i]});  // y -= (L^-1 P X_RS) x'_S
        descs.push_back(rhs_desc(x.fd, YR[i], ldy[i], q));                                          // x'_R = U^-1 y
        g2.push_back(GemmProb<T>{x.Tt, YR[i], slot, r, q, nR, x.ldtt, ldy[i], ldh[par]});           // x_S = x'_S - T^T x'_R
        rows.push_back(RowJob<T>{YR[i], ldy[i], dst, ldd, x.p + r, nR, q, ROW_SCATTER});
      }
      rows.push_back(RowJob<T>{slot, ldh[par], dst, ldd, x.p, r, q, ROW_SCATTER});
    }
    run_gemms(tmp, gemms, 1, s);
    tri_solve_batch(tmp, descs, q, false, true, s);
    run_gemms(tmp, g2, 1, s);
    run_rows(tmp, rows, s);
  }
  HSS_HIP(hipStreamSynchronize(s));
}

struct LruArgs {  // host-side description of the optional update  - C*M*Z  (pointers in the memory space `where` names)
  const void* C = nullptr;
  const void* M = nullptr;
  const void* Z = nullptr;
  int64_t ldc = 0, ldm = 0, ldz = 0, r1 = 0, r2 = 0;
};
template <class T>
HssT<T>* compress_impl(int64_t n, const T* A, int64_t lda, int where, const hs_hss_options* o, const int64_t* perm = nullptr, void* stream = nullptr,
                       const LruArgs* la = nullptr, BlockOp<T>* bop = nullptr) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    hs_set_error(HS_ERR_DEVICE, 0, "no HIP device available (the HSS module has no CPU fallback)");
    throw (int)HS_ERR_DEVICE;
  }
  if (n <= 0 || n > (1 << 30) || (!bop && (lda < n || !A))) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_compress needs n > 0, lda >= n and a matrix");
    throw (int)HS_ERR_ARGUMENT;
  }
  hs_hss_options opt;
  hs_hss_options_default(&opt);
  if (o) opt = *o;
  if (opt.leafsize < 1 || opt.atol < 0 || opt.rtol < 0 || opt.first_split < 0 || opt.first_split > n) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: leafsize >= 1, atol, rtol >= 0, 0 <= first_split <= n required");
    throw (int)HS_ERR_ARGUMENT;
  }
  if (opt.pad <= 0) opt.pad = 8;
  if (!(opt.level_scale > 0.0) || opt.level_scale > 1.0) opt.level_scale = 0.5;
  std::unique_ptr<HssT<T>> H(new HssT<T>());
  H->n = (int)n;
  H->opt = opt;
  if (stream) {
    H->s = (hipStream_t)stream;
  } else {
    HSS_HIP(hipStreamCreate(&H->s));
    H->own_stream = true;
  }
  if (perm) {  // must be a permutation of 0..n-1
    std::vector<int> hp((size_t)n);
    std::vector<char> seen((size_t)n, 0);
    for (int64_t i = 0; i < n; ++i) {
      if (perm[i] < 0 || perm[i] >= n || seen[(size_t)perm[i]]) {
        hs_set_error(HS_ERR_ARGUMENT, i, "ArgumentError: perm is not a permutation of 0..n-1 (entry %lld)", (long long)i);
        throw (int)HS_ERR_ARGUMENT;
      }
      seen[(size_t)perm[i]] = 1;
      hp[(size_t)i] = (int)perm[i];
    }
    H->hinvperm.assign((size_t)n, 0);
    for (int64_t i = 0; i < n; ++i) H->hinvperm[(size_t)hp[(size_t)i]] = (int)i;
    H->hperm = hp;
    H->perm = H->permpool.template get<int>((size_t)n);
    HSS_HIP(hipMemcpyAsync(H->perm, hp.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, H->s));
    HSS_HIP(hipStreamSynchronize(H->s));
  }
  Pool in(global_cache());
  const T* dA = A;
  int ld = (int)lda;
  if (where == 0 && !bop) {
    ld = ev((int)n);
    T* d = in.get<T>((size_t)ld * n);
    HSS_HIP(hipMemcpy2D(d, sizeof(T) * ld, A, sizeof(T) * lda, sizeof(T) * n, n, hipMemcpyHostToDevice));
    dA = d;
  }
  Lru<T> lru;
  if (la && la->r1 > 0 && la->r2 > 0) {
    const bool m_id = !la->M && where != 0 && la->r1 == la->r2;  // device operands: M == NULL stands for the identity
    if (!la->C || (!la->M && !m_id) || !la->Z || la->ldc < n || (la->M && la->ldm < la->r1) || la->ldz < la->r2 || la->r1 > n || la->r2 > n) {
      hs_set_error(HS_ERR_DIMENSION, 0, "DimensionMismatch: the low-rank update needs C (n x r1), M (r1 x r2), Z (r2 x n)");
      throw (int)HS_ERR_DIMENSION;
    }
    lru.r1 = (int)la->r1;
    lru.r2 = (int)la->r2;
    if (where == 0) {
      T* dC = in.get<T>((size_t)ev((int)n) * la->r1);
      T* dM = in.get<T>((size_t)ev(lru.r1) * la->r2);
      T* dZ = in.get<T>((size_t)ev(lru.r2) * n);
      HSS_HIP(hipMemcpy2D(dC, sizeof(T) * ev((int)n), la->C, sizeof(T) * la->ldc, sizeof(T) * n, la->r1, hipMemcpyHostToDevice));
      HSS_HIP(hipMemcpy2D(dM, sizeof(T) * ev(lru.r1), la->M, sizeof(T) * la->ldm, sizeof(T) * la->r1, la->r2, hipMemcpyHostToDevice));
      HSS_HIP(hipMemcpy2D(dZ, sizeof(T) * ev(lru.r2), la->Z, sizeof(T) * la->ldz, sizeof(T) * la->r2, n, hipMemcpyHostToDevice));
      lru.C = dC; lru.ldc = ev((int)n);
      lru.M = dM; lru.ldm = ev(lru.r1);
      lru.Z = dZ; lru.ldz = ev(lru.r2);
    } else {
      lru.C = (const T*)la->C; lru.ldc = (int)la->ldc;
      lru.M = (const T*)la->M; lru.ldm = (int)la->ldm;
      lru.Z = (const T*)la->Z; lru.ldz = (int)la->ldz;
    }
  }
  auto t0 = std::chrono::steady_clock::now();
  int k = (int)std::min<int64_t>(std::max<int64_t>(opt.kest > 0 ? opt.kest : 64, 8), n);
  if (bop) bop->begin(H->s);
  try {
    for (;;) {
      std::vector<CBlock<T>> cb(1);
      cb[0].A = dA; cb[0].lda = ld; cb[0].n = (int)n; cb[0].off = 0;
      cb[0].perm = H->perm; cb[0].hperm = H->hperm.empty() ? nullptr : H->hperm.data();
      cb[0].lru = lru;
      cb[0].leafsize = (int)H->opt.leafsize; cb[0].first_split = (int)H->opt.first_split;
      cb[0].seed = (uint64_t)H->opt.seed;
      if (compress_fixed<T>(*H, cb, k, bop)) break;
      if (k >= n) break;
      k = (int)std::min<int64_t>(2 * (int64_t)k, n);
    }
  } catch (...) {
    if (bop) bop->end(H->s);
    throw;
  }
  if (bop) bop->end(H->s);
  H->t_compress = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return H.release();
}

// C (rows x r, original row order) = P' [I; T]: row p[i] = e_i for i < r, row p[r + i] = T[i, :]
template <class T>
__global__ __launch_bounds__(256) void id_expand_kernel(const int* __restrict__ p, const T* __restrict__ Tm, int ldt, int rows, int r, T* __restrict__ C, int ldc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const int orow = p[i];
  for (int j = 0; j < r; ++j) C[(size_t)orow + (size_t)j * ldc] = i < r ? (i == j ? Scal<T>::one() : Scal<T>::zero()) : Tm[(size_t)(i - r) + (size_t)j * ldt];
}

}  // namespace

template <class T>
int lowrank_id_batch(LowRankJob<T>* jobs, int njobs, double atol, double rtol, hipStream_t s) {
  static const bool qtime = getenv("HS_QR_TIMING") != nullptr;
  std::vector<LowRankJob<T>> todo(jobs, jobs + njobs);
  std::vector<int> which(njobs);
  for (int a = 0; a < njobs; ++a) which[a] = a;
  try {
    for (int pass = 0; !todo.empty(); ++pass) {
      auto t0 = std::chrono::steady_clock::now();
      // pivot order: tournament-pivoted LU of the sketches (its own rank estimate is not used: |u_jj| overestimates the residual norms)
      // HS_QR_ORDER=norm: no pivoted LU -- the sketch alone, the windows of qr_refine chosen by downdated residual norms (a blocked column-pivoted QR)
      const bool norm_order = qr_norm_order();
      int st = lowrank_compress_batch<T>(todo.data(), (int)todo.size(), atol, rtol, s, false, true, norm_order);
      if (st != 0) return st;
      if (qtime) {
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[hs qr] sketch + tournament-pivoted LU of %zu blocks: %.2f ms\n", todo.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      }
      Pool tmp(global_cache()), keep(global_cache());
      std::vector<QrJob<T>> qj;
      std::vector<int> idx;
      for (size_t a = 0; a < todo.size(); ++a) {
        LowRank<T>& o = *todo[a].out;
        if (o.rows <= 0 || o.cols <= 0 || !o.Y0) continue;
        QrJob<T> q;
        q.M = o.Y0;
        q.ldm = o.ldp;
        q.m = o.rows;
        q.q = o.k;
        q.p = o.rperm;
        q.rmax = o.k;
        q.norm_select = norm_order;
        q.atol_scale = std::sqrt((double)o.k);  // the sketch is not normalised: |row of X*Omega| ~ sqrt(k) |row of X|
        qj.push_back(q);
        idx.push_back((int)a);
      }
      qr_refine<T>(tmp, keep, qj, atol, rtol, 0.0, s);
      std::vector<RowJob<T>> rows;
      std::vector<LowRankJob<T>> again;
      for (size_t b = 0; b < qj.size(); ++b) {
        LowRankJob<T>& J = todo[idx[b]];
        LowRank<T>& o = *J.out;
        const int r = qj[b].r, kmax = std::min(J.rows, J.cols);
        if (r + 8 > o.k && o.k < kmax) {  // the sketch was too narrow to see the end of the spectrum: once more with twice the width
          LowRankJob<T> n = J;
          n.kinit = std::min(2 * o.k, kmax);
          lowrank_free(o);
          again.push_back(n);
          continue;
        }
        o.r = r;
        o.top = qj[b].top;
        o.ldz = std::max(2, (r + 1) / 2 * 2);
        o.ldc = std::max(2, (o.rows + 1) / 2 * 2);
        if (hipMalloc((void**)&o.Z, sizeof(T) * ((size_t)o.ldz * J.cols + 32)) != hipSuccess ||
            hipMalloc((void**)&o.Cd, sizeof(T) * ((size_t)o.ldc * std::max(r, 1) + 32)) != hipSuccess) {
          hs_set_error(HS_ERR_NOMEM, 0, "hipMalloc of a low-rank factor (%d x %d) failed", r, J.cols);
          return HS_ERR_NOMEM;
        }
        if (r > 0) {
          rows.push_back(RowJob<T>{J.X, J.ldx, o.Z, o.ldz, o.rperm, r, J.cols, ROW_GATHER});  // Z = the skeleton rows of X
          hipLaunchKernelGGL(id_expand_kernel<T>, dim3((o.rows + 255) / 256), dim3(256), 0, s, (const int*)o.rperm, (const T*)qj[b].Tm, qj[b].ldt, o.rows, r, o.Cd, o.ldc);
        }
        // the packed LU and the sketch copy are not needed again
        hs_lr_free(o.Lp);
        hs_lr_free(o.Y0);
        o.Lp = nullptr;
        o.Y0 = nullptr;
      }
      run_rows(tmp, rows, s);
      HSS_HIP(hipStreamSynchronize(s));
      todo.swap(again);
    }
  } catch (int code) {
    return code;
  }
  return 0;
}
template int lowrank_id_batch<double>(LowRankJob<double>*, int, double, double, hipStream_t);
template int lowrank_id_batch<cplx>(LowRankJob<cplx>*, int, double, double, hipStream_t);

struct hs_hss {
  int is_complex;
  void* impl;
};

#define HSS_GUARD(...)                \
  try {                               \
    __VA_ARGS__;                      \
    return HS_OK;                     \
  } catch (int code) {                \
    return code;                      \
  } catch (const std::bad_alloc&) {   \
    hs_set_error(HS_ERR_NOMEM, 0, "host allocation failed"); \
    return HS_ERR_NOMEM;              \
  }

extern "C" void hs_hss_options_default(hs_hss_options* o) {
  o->leafsize = 64;
  o->first_split = 0;
  o->atol = 1e-6;
  o->rtol = 1e-6;
  o->kest = 64;
  o->pad = 8;
  o->seed = 123;
  o->level_scale = 0.5;
}

extern "C" int hs_hss_compress_d(int64_t n, const double* A, int64_t lda, int where, const hs_hss_options* o, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HSS_GUARD(*out = new hs_hss{0, compress_impl<double>(n, A, lda, where, o)});
}
extern "C" int hs_hss_compress_z(int64_t n, const double* A, int64_t lda, int where, const hs_hss_options* o, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HSS_GUARD(*out = new hs_hss{1, compress_impl<cplx>(n, (const cplx*)A, lda, where, o)});
}

#define HSS_DISPATCH(H, expr_d, expr_z) ((H)->is_complex ? (expr_z) : (expr_d))
#define HD(H) ((HssT<double>*)(H)->impl)
#define HZ(H) ((HssT<cplx>*)(H)->impl)

extern "C" int hs_hss_compress_ex_d(int64_t n, const double* A, int64_t lda, int where, const int64_t* perm, const hs_hss_options* o, void* stream,
                                    hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HSS_GUARD(*out = new hs_hss{0, compress_impl<double>(n, A, lda, where, o, perm, stream)});
}
extern "C" int hs_hss_compress_ex_z(int64_t n, const double* A, int64_t lda, int where, const int64_t* perm, const hs_hss_options* o, void* stream,
                                    hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HSS_GUARD(*out = new hs_hss{1, compress_impl<cplx>(n, (const cplx*)A, lda, where, o, perm, stream)});
}

// H ~= (B - C*M*Z)[perm, perm] without forming the matrix: the Schur complement of a compressed front as the reference samples it
// (`_schur_complement`, `_sample_schur!`, `_getindex_schur`, src/factorization.jl:228-249) under `randcompress_adaptive` (:110)
extern "C" int hs_hss_compress_lru_d(int64_t n, const double* B, int64_t ldb, const double* C_, int64_t ldc, const double* M, int64_t ldm, const double* Z,
                                     int64_t ldz, int64_t r1, int64_t r2, int where, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  LruArgs la{C_, M, Z, ldc, ldm, ldz, r1, r2};
  HSS_GUARD(*out = new hs_hss{0, compress_impl<double>(n, B, ldb, where, o, perm, stream, &la)});
}
extern "C" int hs_hss_compress_lru_z(int64_t n, const double* B, int64_t ldb, const double* C_, int64_t ldc, const double* M, int64_t ldm, const double* Z,
                                     int64_t ldz, int64_t r1, int64_t r2, int where, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  LruArgs la{C_, M, Z, ldc, ldm, ldz, r1, r2};
  HSS_GUARD(*out = new hs_hss{1, compress_impl<cplx>(n, (const cplx*)B, ldb, where, o, perm, stream, &la)});
}

extern "C" int hs_hss_set_stream(hs_hss* H, void* stream) {
  if (!H) return HS_ERR_ARGUMENT;  // stream == NULL: the default stream
  if (H->is_complex) {
    if (HZ(H)->own_stream && HZ(H)->s) (void)hipStreamDestroy(HZ(H)->s);
    HZ(H)->s = (hipStream_t)stream;
    HZ(H)->own_stream = false;
  } else {
    if (HD(H)->own_stream && HD(H)->s) (void)hipStreamDestroy(HD(H)->s);
    HD(H)->s = (hipStream_t)stream;
    HD(H)->own_stream = false;
  }
  return HS_OK;
}

extern "C" int64_t hs_hss_size(const hs_hss* H) { return H ? HSS_DISPATCH(H, HD(H)->n, HZ(H)->n) : 0; }
extern "C" int64_t hs_hss_samples(const hs_hss* H) { return H ? HSS_DISPATCH(H, HD(H)->k, HZ(H)->k) : 0; }
template <class T>
static int64_t bytes_of(const HssT<T>* H) {
  int64_t b = 0;
  for (auto& pr : H->keep.v) b += (int64_t)pr.second;
  for (auto& pr : H->permpool.v) b += (int64_t)pr.second;
  return b;
}
extern "C" int64_t hs_hss_bytes(const hs_hss* H) { return H ? HSS_DISPATCH(H, bytes_of(HD(H)), bytes_of(HZ(H))) : 0; }
extern "C" int64_t hs_hss_num_nodes(const hs_hss* H) { return H ? (int64_t)HSS_DISPATCH(H, HD(H)->nd.size(), HZ(H)->nd.size()) : 0; }
extern "C" double hs_hss_time(const hs_hss* H, int what) {
  if (!H) return 0.0;
  return what == 0 ? HSS_DISPATCH(H, HD(H)->t_compress, HZ(H)->t_compress) : HSS_DISPATCH(H, HD(H)->t_factor, HZ(H)->t_factor);
}
template <class T>
static int64_t rank_of(const HssT<T>* H) {
  int64_t r = 0;
  for (size_t i = 1; i < H->nd.size(); ++i) r = std::max<int64_t>(r, H->nd[i].r);
  return r;
}
extern "C" int64_t hs_hss_rank(const hs_hss* H) { return H ? HSS_DISPATCH(H, rank_of(HD(H)), rank_of(HZ(H))) : 0; }

template <class T>
static int node_info(const HssT<T>* H, int64_t i, int64_t* out) {
  if (i < 0 || i >= (int64_t)H->nd.size() || !out) {
    hs_set_error(HS_ERR_ARGUMENT, i, "ArgumentError: HSS node %lld out of range", (long long)i);
    return HS_ERR_ARGUMENT;
  }
  const HNode<T>& x = H->nd[i];
  out[0] = x.lo; out[1] = x.hi; out[2] = x.left; out[3] = x.right; out[4] = x.level; out[5] = x.m; out[6] = x.r; out[7] = x.left < 0;
  return HS_OK;
}
extern "C" int hs_hss_node_info(const hs_hss* H, int64_t node, int64_t out[8]) {
  if (!H) return HS_ERR_ARGUMENT;
  return HSS_DISPATCH(H, node_info(HD(H), node, out), node_info(HZ(H), node, out));
}

template <class T>
static void node_data(const HssT<T>* H, int64_t i, int64_t* p, T* Tm, T* D, T* B12, T* B21) {
  if (i < 0 || i >= (int64_t)H->nd.size()) {
    hs_set_error(HS_ERR_ARGUMENT, i, "ArgumentError: HSS node %lld out of range", (long long)i);
    throw (int)HS_ERR_ARGUMENT;
  }
  const HNode<T>& x = H->nd[i];
  const int m = x.m, r = x.r, nR = m - r;
  if (p && x.p) {
    std::vector<int> hp(m);
    HSS_HIP(hipMemcpy(hp.data(), x.p, sizeof(int) * m, hipMemcpyDeviceToHost));
    for (int a = 0; a < m; ++a) p[a] = hp[a];
  }
  if (Tm && x.Tm && nR > 0 && r > 0) HSS_HIP(hipMemcpy2D(Tm, sizeof(T) * nR, x.Tm, sizeof(T) * x.ldt, sizeof(T) * nR, r, hipMemcpyDeviceToHost));
  if (D && x.D) HSS_HIP(hipMemcpy2D(D, sizeof(T) * m, x.D, sizeof(T) * x.ldd, sizeof(T) * m, m, hipMemcpyDeviceToHost));
  if (x.left >= 0) {
    const int rl = H->nd[x.left].r, rr = H->nd[x.right].r;
    if (B12 && rl > 0 && rr > 0) HSS_HIP(hipMemcpy2D(B12, sizeof(T) * rl, x.B12, sizeof(T) * x.ld12, sizeof(T) * rl, rr, hipMemcpyDeviceToHost));
    if (B21 && rl > 0 && rr > 0) HSS_HIP(hipMemcpy2D(B21, sizeof(T) * rr, x.B21, sizeof(T) * x.ld21, sizeof(T) * rr, rl, hipMemcpyDeviceToHost));
  }
}
extern "C" int hs_hss_node_data(const hs_hss* H, int64_t node, int64_t* p, double* T_, double* D, double* B12, double* B21) {
  if (!H) return HS_ERR_ARGUMENT;
  HSS_GUARD(if (H->is_complex) node_data<cplx>(HZ(H), node, p, (cplx*)T_, (cplx*)D, (cplx*)B12, (cplx*)B21);
            else node_data<double>(HD(H), node, p, T_, D, B12, B21));
}

// host <-> device staging of n x q blocks
template <class T, class F>
static void with_device_block(int n, const T* Bin, int64_t ldin, T* Bout, int64_t ldout, int q, int where, F&& f) {
  if (where != 0) {
    f(Bin, (int)ldin, Bout, (int)ldout);
    return;
  }
  Pool st(global_cache());
  const int ld = ev(n);
  T* dI = st.get<T>((size_t)ld * q);
  T* dO = (Bout == Bin) ? dI : st.get<T>((size_t)ld * q);
  HSS_HIP(hipMemcpy2D(dI, sizeof(T) * ld, Bin, sizeof(T) * ldin, sizeof(T) * n, q, hipMemcpyHostToDevice));
  f(dI, ld, dO, ld);
  HSS_HIP(hipMemcpy2D(Bout, sizeof(T) * ldout, dO, sizeof(T) * ld, sizeof(T) * n, q, hipMemcpyDeviceToHost));
}

static int hss_mul_abi(bool trans, hs_hss* H, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t nrhs, int where) {
  if (!H || !X || !Y || nrhs < 0 || X == Y) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_mul needs distinct X and Y");
    return HS_ERR_ARGUMENT;
  }
  if (nrhs == 0) return HS_OK;
  HSS_GUARD(
      if (H->is_complex) with_device_block<cplx>(HZ(H)->n, (const cplx*)X, ldx, (cplx*)Y, ldy, (int)nrhs, where,
                                                 [&](const cplx* a, int la, cplx* b, int lb) { hss_mul<cplx>(*HZ(H), a, la, b, lb, (int)nrhs, trans); });
      else with_device_block<double>(HD(H)->n, X, ldx, Y, ldy, (int)nrhs, where,
                                     [&](const double* a, int la, double* b, int lb) { hss_mul<double>(*HD(H), a, la, b, lb, (int)nrhs, trans); }));
}

extern "C" int hs_hss_mul(hs_hss* H, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t nrhs, int where) {
  return hss_mul_abi(false, H, X, ldx, Y, ldy, nrhs, where);
}
extern "C" int hs_hss_mul_t(hs_hss* H, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t nrhs, int where) {
  return hss_mul_abi(true, H, X, ldx, Y, ldy, nrhs, where);
}

// `H.A11` (which = 0) / `H.A22` (which = 1): the diagonal block of the top-level split as an HSS matrix of its own that SHARES the
// generators of H (H must outlive it).  Its index space is the block's own: 0 .. size-1 in the order of the cluster tree.
template <class T>
static HssT<T>* child_impl(HssT<T>* P, int which, int top_node = -1) {
  if (P->nd[0].left < 0 && which != 2) {
    hs_set_error(HS_ERR_HSS_LEAF, 0, "One of the Schur complements turned into a leaf. Aborting.");  // factorization.jl:164
    throw (int)HS_ERR_HSS_LEAF;
  }
  const int top = top_node >= 0 ? top_node : (which == 2 ? 0 : (which == 0 ? P->nd[0].left : P->nd[0].right));  // top_node: a root of a forest
  const int off = P->nd[top].lo;
  std::vector<int> ids, cur{top};
  while (!cur.empty()) {  // breadth-first renumbering of the subtree
    std::vector<int> nxt;
    for (int i : cur) {
      ids.push_back(i);
      if (P->nd[i].left >= 0) {
        nxt.push_back(P->nd[i].left);
        nxt.push_back(P->nd[i].right);
      }
    }
    cur.swap(nxt);
  }
  std::vector<int> nw(P->nd.size(), -1);
  for (size_t k = 0; k < ids.size(); ++k) nw[ids[k]] = (int)k;
  std::unique_ptr<HssT<T>> C(new HssT<T>());
  C->n = P->nd[top].hi - off;
  C->k = P->k;
  C->opt = P->opt;
  C->s = P->s;
  C->own_stream = false;
  for (int old : ids) {
    HNode<T> y = P->nd[old];  // shares every device pointer
    y.lo -= off;
    y.hi -= off;
    if (which != 2) y.level -= 1;
    y.parent = old == top ? -1 : nw[y.parent];
    if (y.left >= 0) {
      y.left = nw[y.left];
      y.right = nw[y.right];
    }
    y.has_front = false;
    memset(&y.fd, 0, sizeof y.fd);
    if (old == top && which != 2) {  // a root keeps no basis
      y.p = nullptr; y.sk = nullptr; y.Tm = nullptr; y.Tt = nullptr; y.NTm = nullptr;
      y.m = y.left >= 0 ? P->nd[P->nd[old].left].r + P->nd[P->nd[old].right].r : y.hi - y.lo;
      y.r = 0;
      y.hinvp.clear();
    }
    C->nd.push_back(y);
  }
  C->nlev = 0;
  for (auto& x : C->nd) C->nlev = std::max(C->nlev, x.level + 1);
  C->lev.assign(C->nlev, {});
  for (int i = 0; i < (int)C->nd.size(); ++i) C->lev[C->nd[i].level].push_back(i);
  return C.release();
}
// `prune_leaves!(H)` of HssMatrices.jl as `_equilibrate_clusters` uses it (src/factorization.jl:143-168): every node whose two children are
// leaves becomes a leaf itself -- D = [D_l  U_l B12 U_r^T; U_r B21 U_l^T  D_r], basis U = blkdiag(U_l, U_r) U_node in the interpolative form
// P^T [I; T] (the skeleton of the node is a subset of its children's skeletons, so the identity rows are there already).  The result is a view:
// it shares every untouched generator with H (H must outlive it) and owns the merged blocks.
template <class T>
static HssT<T>* prune_impl(HssT<T>* P) {
  if (P->nd[0].left < 0) {
    hs_set_error(HS_ERR_HSS_LEAF, 0, "One of the Schur complements turned into a leaf. Aborting.");  // factorization.jl:163-165
    throw (int)HS_ERR_HSS_LEAF;
  }
  hipStream_t s = P->s;
  auto& nd = P->nd;
  const int N = (int)nd.size();
  std::vector<char> merge(N, 0), gone(N, 0);
  for (int i = 0; i < N; ++i)
    if (nd[i].left >= 0 && nd[nd[i].left].left < 0 && nd[nd[i].right].left < 0) {
      merge[i] = 1;
      gone[nd[i].left] = gone[nd[i].right] = 1;
    }
  std::unique_ptr<HssT<T>> C(new HssT<T>());
  C->n = P->n;
  C->k = P->k;
  C->opt = P->opt;
  C->s = s;
  C->own_stream = false;
  C->perm = P->perm;  // shared
  C->hinvperm = P->hinvperm;
  C->hperm = P->hperm;
  std::vector<int> nw(N, -1);
  int cnt = 0;
  for (int i = 0; i < N; ++i)
    if (!gone[i]) nw[i] = cnt++;  // breadth-first order is preserved
  Pool tmp(global_cache());
  std::vector<GemmProb<T>> g1, g2;
  std::vector<SubJob<T>> subs;
  for (int i = 0; i < N; ++i) {
    if (gone[i]) continue;
    HNode<T> y = nd[i];  // shares the device pointers
    y.parent = y.parent >= 0 ? nw[y.parent] : -1;
    y.has_front = false;
    memset(&y.fd, 0, sizeof y.fd);
    y.hinvp.clear();
    y.NTm = y.DTt = y.B12t = y.B21t = nullptr;
    if (!merge[i]) {
      if (y.left >= 0) {
        y.left = nw[y.left];
        y.right = nw[y.right];
      }
      C->nd.push_back(y);
      continue;
    }
    HNode<T>&l = nd[nd[i].left], &r = nd[nd[i].right];
    const int ml = l.m, mr = r.m, m = ml + mr, rl = l.r, rr = r.r;
    y.left = y.right = -1;
    y.B12 = y.B21 = nullptr;
    y.ldd = ev(m);
    y.D = C->keep.template get<T>((size_t)y.ldd * m);
    // D_l, D_r on the diagonal; U_l B12 U_r^T and U_r B21 U_l^T off it
    subs.push_back(SubJob<T>{l.D, l.ldd, nullptr, nullptr, 0, 0, ml, ml, y.D, y.ldd, 0});
    subs.push_back(SubJob<T>{r.D, r.ldd, nullptr, nullptr, 0, 0, mr, mr, y.D + ml + (size_t)ml * y.ldd, y.ldd, 0});
    T* Ul = tmp.get<T>((size_t)ev(ml) * std::max(rl, 1));
    T* Ur = tmp.get<T>((size_t)ev(mr) * std::max(rr, 1));
    hss_basis<T>(*P, nd[i].left, Ul, ev(ml));
    hss_basis<T>(*P, nd[i].right, Ur, ev(mr));
    T* UrT = tmp.get<T>((size_t)ev(rr) * mr);
    T* UlT = tmp.get<T>((size_t)ev(rl) * ml);
    subs.push_back(SubJob<T>{Ur, ev(mr), nullptr, nullptr, 0, 0, mr, rr, UrT, ev(rr), 1});
    subs.push_back(SubJob<T>{Ul, ev(ml), nullptr, nullptr, 0, 0, ml, rl, UlT, ev(rl), 1});
    T* t12 = tmp.get<T>((size_t)ev(ml) * std::max(rr, 1));
    T* t21 = tmp.get<T>((size_t)ev(mr) * std::max(rl, 1));
    HSS_HIP(hipMemsetAsync(t12, 0, sizeof(T) * (size_t)ev(ml) * std::max(rr, 1), s));
    HSS_HIP(hipMemsetAsync(t21, 0, sizeof(T) * (size_t)ev(mr) * std::max(rl, 1), s));
    HSS_HIP(hipMemset2DAsync(y.D + (size_t)ml * y.ldd, sizeof(T) * y.ldd, 0, sizeof(T) * ml, mr, s));
    HSS_HIP(hipMemset2DAsync(y.D + ml, sizeof(T) * y.ldd, 0, sizeof(T) * mr, ml, s));
    g1.push_back(GemmProb<T>{Ul, nd[i].B12, t12, ml, rr, rl, ev(ml), nd[i].ld12, ev(ml)});
    g1.push_back(GemmProb<T>{Ur, nd[i].B21, t21, mr, rl, rr, ev(mr), nd[i].ld21, ev(mr)});
    g2.push_back(GemmProb<T>{t12, UrT, y.D + (size_t)ml * y.ldd, ml, mr, rr, ev(ml), ev(rr), y.ldd});
    g2.push_back(GemmProb<T>{t21, UlT, y.D + ml, mr, ml, rl, ev(mr), ev(rl), y.ldd});
    if (i != 0) {
      // the node's basis over the merged leaf: E = blkdiag(U_l, U_r) U_node; its skeleton rows are identity rows of E
      const int rk = nd[i].r;
      std::vector<int> pp(nd[i].m), pl(ml), pr(mr);
      HSS_HIP(hipMemcpy(pp.data(), nd[i].p, sizeof(int) * nd[i].m, hipMemcpyDeviceToHost));
      HSS_HIP(hipMemcpy(pl.data(), l.p, sizeof(int) * ml, hipMemcpyDeviceToHost));
      HSS_HIP(hipMemcpy(pr.data(), r.p, sizeof(int) * mr, hipMemcpyDeviceToHost));
      std::vector<int> np(m);
      std::vector<char> isk(m, 0);
      for (int a = 0; a < rk; ++a) {
        const int q = pp[a];  // position in [sk_l; sk_r]
        const int loc = q < rl ? pl[q] : ml + pr[q - rl];
        np[a] = loc;
        isk[loc] = 1;
      }
      int at = rk;
      for (int a = 0; a < m; ++a)
        if (!isk[a]) np[at++] = a;
      T* E = tmp.get<T>((size_t)ev(m) * std::max(rk, 1));
      hss_basis<T>(*P, i, E, ev(m));
      y.m = m;
      y.p = C->keep.template get<int>(m);
      HSS_HIP(hipMemcpy(y.p, np.data(), sizeof(int) * m, hipMemcpyHostToDevice));
      y.ldt = ev(m - rk);
      y.ldtt = ev(rk);
      y.Tm = C->keep.template get<T>((size_t)y.ldt * std::max(rk, 1));
      y.Tt = C->keep.template get<T>((size_t)y.ldtt * std::max(m - rk, 1));
      subs.push_back(SubJob<T>{E, ev(m), y.p + rk, nullptr, 0, 0, m - rk, rk, y.Tm, y.ldt, 0});
      subs.push_back(SubJob<T>{E, ev(m), y.p + rk, nullptr, 0, 0, m - rk, rk, y.Tt, y.ldtt, 1});
      // sk (global indices of the skeleton) is unchanged: shared with H
    } else {
      y.m = m;
    }
    C->nd.push_back(y);
  }
  run_subs(tmp, subs, s);  // diagonal blocks, transposed bases, interpolation rows
  run_gemms(tmp, g1, 0, s);
  run_gemms(tmp, g2, 0, s);
  HSS_HIP(hipStreamSynchronize(s));
  C->nlev = 0;
  for (auto& x : C->nd) C->nlev = std::max(C->nlev, x.level + 1);
  C->lev.assign(C->nlev, {});
  for (int i = 0; i < (int)C->nd.size(); ++i) C->lev[C->nd[i].level].push_back(i);
  for (auto& x : C->nd)  // offsets inside the parents' local vectors (left child first)
    if (x.parent >= 0) x.off_in_parent = (&x == &C->nd[C->nd[x.parent].left]) ? 0 : C->nd[C->nd[x.parent].left].r;
  return C.release();
}
extern "C" int hs_hss_prune_leaves(hs_hss* H, hs_hss** out) {
  if (!H || !out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HSS_GUARD(*out = H->is_complex ? new hs_hss{1, prune_impl<cplx>(HZ(H))} : new hs_hss{0, prune_impl<double>(HD(H))});
}
// `compatible(cl1, cl2)`: the two cluster trees have the same shape (what HSS-by-HSS arithmetic needs of its operands)
template <class T>
static int same_shape(const HssT<T>* A, int a, const HssT<T>* B, int b) {
  const bool la = A->nd[a].left < 0, lb = B->nd[b].left < 0;
  if (la != lb) return 0;
  if (la) return 1;
  return same_shape(A, A->nd[a].left, B, B->nd[b].left) && same_shape(A, A->nd[a].right, B, B->nd[b].right);
}
extern "C" int hs_hss_compatible(const hs_hss* A, const hs_hss* B) {
  if (!A || !B || A->is_complex != B->is_complex) return 0;
  return A->is_complex ? same_shape(HZ(A), 0, HZ(B), 0) : same_shape(HD(A), 0, HD(B), 0);
}
extern "C" int64_t hs_hss_depth(const hs_hss* H) { return H ? HSS_DISPATCH(H, HD(H)->nlev, HZ(H)->nlev) : 0; }

// ------------------------------------------------------------------------------------------------
// One contiguous device buffer for a whole HSS matrix -- what crosses ranks at a join of the elimination tree when the child's Schur
// complement travels as an HssMatrix (src/factorization.jl:78-112: the parent reads S1, S2 as HSS; SURVEY.md 8(e): "ship HSS generators
// instead of dense S"): 100 MB of generators instead of a 2.1 GB dense block at the top of Poisson 128^3.  Layout: PackHdr, nnodes x
// PackNode, the permutation (n ints, if any), then every generator block at a 256-byte aligned offset with the leading dimension it
// has here.  Factors of an elimination (hs_hss_factor) and the lazily made transposes are not packed: the receiver rebuilds them on
// first use.  Unpacking copies the buffer ONCE into a block the new matrix owns and points its nodes into it.
// ------------------------------------------------------------------------------------------------
struct PackHdr {
  int64_t magic, bytes, is_complex, n, k, nlev, nnodes, has_perm, off_nodes, off_perm;
  hs_hss_options opt;
  int64_t reserved[4];
};
struct PackNode {
  int32_t lo, hi, left, right, parent, level, m, r, ldt, ldtt, ldd, ld12, ld21, off_in_parent, rl, rr;
  int64_t o_p, o_sk, o_Tm, o_Tt, o_D, o_B12, o_B21;  // byte offsets from the start of the buffer, -1: absent
};
static constexpr int64_t HS_PACK_MAGIC = 0x4853534850414b31ll;  // "HSSHPAK1"
static inline int64_t pack_up(int64_t b) { return (b + 255) / 256 * 256; }

template <class T>
static void pack_layout(const HssT<T>* H, PackHdr& hd, std::vector<PackNode>& pn) {
  const int N = (int)H->nd.size();
  memset(&hd, 0, sizeof hd);
  hd.magic = HS_PACK_MAGIC;
  hd.is_complex = sizeof(T) == 16;
  hd.n = H->n; hd.k = H->k; hd.nlev = H->nlev; hd.nnodes = N;
  hd.has_perm = H->perm != nullptr;
  hd.opt = H->opt;
  pn.assign((size_t)N, PackNode());
  int64_t at = pack_up((int64_t)sizeof(PackHdr));
  hd.off_nodes = at;
  at = pack_up(at + (int64_t)sizeof(PackNode) * N);
  hd.off_perm = -1;
  if (H->perm) {
    hd.off_perm = at;
    at = pack_up(at + (int64_t)sizeof(int) * H->n);
  }
  auto blob = [&](const void* ptr, int64_t bytes) -> int64_t {
    if (!ptr || bytes <= 0) return -1;
    const int64_t o = at;
    at = pack_up(at + bytes);
    return o;
  };
  for (int i = 0; i < N; ++i) {
    const HNode<T>& x = H->nd[i];
    PackNode& q = pn[(size_t)i];
    q.lo = x.lo; q.hi = x.hi; q.left = x.left; q.right = x.right; q.parent = x.parent; q.level = x.level; q.m = x.m; q.r = x.r;
    q.ldt = x.ldt; q.ldtt = x.ldtt; q.ldd = x.ldd; q.ld12 = x.ld12; q.ld21 = x.ld21; q.off_in_parent = x.off_in_parent;
    q.rl = x.left >= 0 ? H->nd[x.left].r : 0;
    q.rr = x.right >= 0 ? H->nd[x.right].r : 0;
    const int nR = x.m - x.r;
    q.o_p = blob(x.p, (int64_t)sizeof(int) * x.m);
    q.o_sk = blob(x.sk, (int64_t)sizeof(int) * x.r);
    q.o_Tm = blob(x.Tm, (int64_t)sizeof(T) * x.ldt * x.r);
    q.o_Tt = blob(x.Tt, (int64_t)sizeof(T) * x.ldtt * std::max(nR, 0));
    q.o_D = blob(x.left < 0 ? x.D : nullptr, (int64_t)sizeof(T) * x.ldd * x.m);
    q.o_B12 = blob(x.B12, (int64_t)sizeof(T) * x.ld12 * q.rr);
    q.o_B21 = blob(x.B21, (int64_t)sizeof(T) * x.ld21 * q.rl);
  }
  hd.bytes = at;
}

template <class T>
static int64_t pack_size_impl(const HssT<T>* H) {
  PackHdr hd;
  std::vector<PackNode> pn;
  pack_layout(H, hd, pn);
  return hd.bytes;
}

template <class T>
static void pack_impl(HssT<T>* H, void* buf, int64_t bytes, hipStream_t s) {
  PackHdr hd;
  std::vector<PackNode> pn;
  pack_layout(H, hd, pn);
  if (!buf || bytes < hd.bytes) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_pack needs a device buffer of %lld bytes, got %lld", (long long)hd.bytes, (long long)bytes);
    throw (int)HS_ERR_ARGUMENT;
  }
  if (!s) s = H->s;
  char* base = (char*)buf;
  Pool tmp(global_cache());
  void* hp = tmp.get_pinned(sizeof(PackHdr) + sizeof(PackNode) * pn.size());
  memcpy(hp, &hd, sizeof hd);
  memcpy((char*)hp + sizeof hd, pn.data(), sizeof(PackNode) * pn.size());
  HSS_HIP(hipMemcpyAsync(base, hp, sizeof hd, hipMemcpyHostToDevice, s));
  HSS_HIP(hipMemcpyAsync(base + hd.off_nodes, (char*)hp + sizeof hd, sizeof(PackNode) * pn.size(), hipMemcpyHostToDevice, s));
  if (H->perm) HSS_HIP(hipMemcpyAsync(base + hd.off_perm, H->perm, sizeof(int) * (size_t)H->n, hipMemcpyDeviceToDevice, s));
  auto put = [&](int64_t off, const void* src, int64_t nbytes) {
    if (off >= 0) HSS_HIP(hipMemcpyAsync(base + off, src, (size_t)nbytes, hipMemcpyDeviceToDevice, s));
  };
  for (size_t i = 0; i < pn.size(); ++i) {
    const HNode<T>& x = H->nd[i];
    const PackNode& q = pn[i];
    put(q.o_p, x.p, (int64_t)sizeof(int) * x.m);
    put(q.o_sk, x.sk, (int64_t)sizeof(int) * x.r);
    put(q.o_Tm, x.Tm, (int64_t)sizeof(T) * x.ldt * x.r);
    put(q.o_Tt, x.Tt, (int64_t)sizeof(T) * x.ldtt * std::max(x.m - x.r, 0));
    put(q.o_D, x.D, (int64_t)sizeof(T) * x.ldd * x.m);
    put(q.o_B12, x.B12, (int64_t)sizeof(T) * x.ld12 * q.rr);
    put(q.o_B21, x.B21, (int64_t)sizeof(T) * x.ld21 * q.rl);
  }
  HSS_HIP(hipStreamSynchronize(s));  // the pinned staging block goes back to its cache with `tmp`
}

template <class T>
static HssT<T>* unpack_impl(const void* buf, int64_t bytes, hipStream_t s) {
  PackHdr hd;
  if (!buf || bytes < (int64_t)sizeof hd) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_unpack: buffer of %lld bytes holds no header", (long long)bytes);
    throw (int)HS_ERR_ARGUMENT;
  }
  HSS_HIP(hipMemcpyAsync(&hd, buf, sizeof hd, hipMemcpyDeviceToHost, s));
  HSS_HIP(hipStreamSynchronize(s));
  if (hd.magic != HS_PACK_MAGIC || hd.bytes > bytes || hd.is_complex != (int64_t)(sizeof(T) == 16) || hd.nnodes <= 0 || hd.n <= 0 ||
      hd.off_nodes + (int64_t)sizeof(PackNode) * hd.nnodes > hd.bytes) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_unpack: not a packed %s HSS matrix of at most %lld bytes", sizeof(T) == 16 ? "ComplexF64" : "Float64", (long long)bytes);
    throw (int)HS_ERR_ARGUMENT;
  }
  std::vector<PackNode> pn((size_t)hd.nnodes);
  HSS_HIP(hipMemcpyAsync(pn.data(), (const char*)buf + hd.off_nodes, sizeof(PackNode) * pn.size(), hipMemcpyDeviceToHost, s));
  std::unique_ptr<HssT<T>> H(new HssT<T>());
  H->n = (int)hd.n; H->k = (int)hd.k; H->nlev = (int)hd.nlev;
  H->opt = hd.opt;
  H->s = s;
  H->own_stream = false;
  if (!s) {
    HSS_HIP(hipStreamCreate(&H->s));
    H->own_stream = true;
    s = H->s;
  }
  char* base = H->keep.template get<char>((size_t)hd.bytes);
  HSS_HIP(hipMemcpyAsync(base, buf, (size_t)hd.bytes, hipMemcpyDeviceToDevice, s));
  if (hd.has_perm) {
    H->perm = (int*)(base + hd.off_perm);
    H->hperm.assign((size_t)hd.n, 0);
    HSS_HIP(hipMemcpyAsync(H->hperm.data(), (const char*)buf + hd.off_perm, sizeof(int) * (size_t)hd.n, hipMemcpyDeviceToHost, s));
  }
  HSS_HIP(hipStreamSynchronize(s));
  if (hd.has_perm) {
    H->hinvperm.assign((size_t)hd.n, 0);
    for (int64_t i = 0; i < hd.n; ++i) {
      const int v = H->hperm[(size_t)i];
      if (v < 0 || v >= hd.n) {
        hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_unpack: corrupt permutation");
        throw (int)HS_ERR_ARGUMENT;
      }
      H->hinvperm[(size_t)v] = (int)i;
    }
  }
  auto at = [&](int64_t off, int64_t nbytes) -> char* {
    if (off < 0) return nullptr;
    if (off + nbytes > hd.bytes) {
      hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_unpack: a generator block lies outside the buffer");
      throw (int)HS_ERR_ARGUMENT;
    }
    return base + off;
  };
  H->nd.assign((size_t)hd.nnodes, HNode<T>());
  for (size_t i = 0; i < pn.size(); ++i) {
    const PackNode& q = pn[i];
    HNode<T>& x = H->nd[i];
    if (q.left >= hd.nnodes || q.right >= hd.nnodes || q.parent >= hd.nnodes || q.level < 0 || q.level >= hd.nlev || q.m < 0 || q.r < 0 || q.r > q.m) {
      hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_unpack: corrupt node table");
      throw (int)HS_ERR_ARGUMENT;
    }
    x.lo = q.lo; x.hi = q.hi; x.left = q.left; x.right = q.right; x.parent = q.parent; x.level = q.level; x.m = q.m; x.r = q.r;
    x.ldt = q.ldt; x.ldtt = q.ldtt; x.ldd = q.ldd; x.ld12 = q.ld12; x.ld21 = q.ld21; x.off_in_parent = q.off_in_parent;
    x.p = (int*)at(q.o_p, (int64_t)sizeof(int) * q.m);
    x.sk = (int*)at(q.o_sk, (int64_t)sizeof(int) * q.r);
    x.Tm = (T*)at(q.o_Tm, (int64_t)sizeof(T) * q.ldt * q.r);
    x.Tt = (T*)at(q.o_Tt, (int64_t)sizeof(T) * q.ldtt * std::max(q.m - q.r, 0));
    x.D = (T*)at(q.o_D, (int64_t)sizeof(T) * q.ldd * q.m);
    x.B12 = (T*)at(q.o_B12, (int64_t)sizeof(T) * q.ld12 * q.rr);
    x.B21 = (T*)at(q.o_B21, (int64_t)sizeof(T) * q.ld21 * q.rl);
  }
  H->lev.assign((size_t)H->nlev, {});
  for (int i = 0; i < (int)H->nd.size(); ++i) H->lev[(size_t)H->nd[(size_t)i].level].push_back(i);
  return H.release();
}

extern "C" int hs_hss_pack_size(const hs_hss* H, int64_t* bytes) {
  if (!H || !bytes) return HS_ERR_ARGUMENT;
  HSS_GUARD(*bytes = HSS_DISPATCH(H, pack_size_impl(HD(H)), pack_size_impl(HZ(H))));
}
extern "C" int hs_hss_pack(hs_hss* H, void* dev_buf, int64_t bytes, void* stream) {
  if (!H) return HS_ERR_ARGUMENT;
  HSS_GUARD(if (H->is_complex) pack_impl(HZ(H), dev_buf, bytes, (hipStream_t)stream); else pack_impl(HD(H), dev_buf, bytes, (hipStream_t)stream));
}
extern "C" int hs_hss_unpack(const void* dev_buf, int64_t bytes, int is_complex, void* stream, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HSS_GUARD(*out = is_complex ? new hs_hss{1, unpack_impl<cplx>(dev_buf, bytes, (hipStream_t)stream)} : new hs_hss{0, unpack_impl<double>(dev_buf, bytes, (hipStream_t)stream)});
}

extern "C" int hs_hss_child(hs_hss* H, int which, hs_hss** out) {
  if (!H || !out || which < 0 || which > 2) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HSS_GUARD(*out = H->is_complex ? new hs_hss{1, child_impl<cplx>(HZ(H), which)} : new hs_hss{0, child_impl<double>(HD(H), which)});
}

extern "C" int hs_hss_getindex(hs_hss* H, const int64_t* I, int64_t ni, const int64_t* J, int64_t nj, double* out, int64_t ldo, int where) {
  if (!H || !out || ni < 0 || nj < 0 || (ni > 0 && !I) || (nj > 0 && !J) || ldo < ni) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_getindex needs index lists and an ni x nj result");
    return HS_ERR_ARGUMENT;
  }
  if (ni == 0 || nj == 0) return HS_OK;
  HSS_GUARD(
      if (where != 0) {
        if (H->is_complex) hss_getindex<cplx>(*HZ(H), I, (int)ni, J, (int)nj, (cplx*)out, (int)ldo);
        else hss_getindex<double>(*HD(H), I, (int)ni, J, (int)nj, out, (int)ldo);
      } else {
        Pool st(global_cache());
        const int ld = ev((int)ni);
        if (H->is_complex) {
          cplx* d = st.get<cplx>((size_t)ld * nj);
          hss_getindex<cplx>(*HZ(H), I, (int)ni, J, (int)nj, d, ld);
          HSS_HIP(hipMemcpy2D(out, sizeof(cplx) * ldo, d, sizeof(cplx) * ld, sizeof(cplx) * ni, nj, hipMemcpyDeviceToHost));
        } else {
          double* d = st.get<double>((size_t)ld * nj);
          hss_getindex<double>(*HD(H), I, (int)ni, J, (int)nj, d, ld);
          HSS_HIP(hipMemcpy2D(out, sizeof(double) * ldo, d, sizeof(double) * ld, sizeof(double) * ni, nj, hipMemcpyDeviceToHost));
        }
      });
}

extern "C" int hs_hss_basis(hs_hss* H, int64_t node, double* out, int64_t ldo, int where) {
  if (!H || !out) return HS_ERR_ARGUMENT;
  HSS_GUARD(
      const int64_t nn = hs_hss_num_nodes(H);
      if (node <= 0 || node >= nn) {
        hs_set_error(HS_ERR_ARGUMENT, node, "ArgumentError: HSS node %lld has no basis", (long long)node);
        throw (int)HS_ERR_ARGUMENT;
      }
      int64_t info[8];
      (void)hs_hss_node_info(H, node, info);
      const int rows = (int)(info[1] - info[0]), r = (int)info[6];
      if (ldo < rows) {
        hs_set_error(HS_ERR_DIMENSION, 0, "DimensionMismatch: the basis has %d rows, ldo = %lld", rows, (long long)ldo);
        throw (int)HS_ERR_DIMENSION;
      }
      if (where != 0) {
        if (H->is_complex) hss_basis<cplx>(*HZ(H), (int)node, (cplx*)out, (int)ldo);
        else hss_basis<double>(*HD(H), (int)node, out, (int)ldo);
      } else {
        Pool st(global_cache());
        const int ld = ev(rows);
        if (H->is_complex) {
          cplx* d = st.get<cplx>((size_t)ld * r);
          hss_basis<cplx>(*HZ(H), (int)node, d, ld);
          HSS_HIP(hipMemcpy2D(out, sizeof(cplx) * ldo, d, sizeof(cplx) * ld, sizeof(cplx) * rows, r, hipMemcpyDeviceToHost));
        } else {
          double* d = st.get<double>((size_t)ld * r);
          hss_basis<double>(*HD(H), (int)node, d, ld);
          HSS_HIP(hipMemcpy2D(out, sizeof(double) * ldo, d, sizeof(double) * ld, sizeof(double) * rows, r, hipMemcpyDeviceToHost));
        }
      });
}

// The off-diagonal blocks of the top-level split in low-rank form (device or host output):
//   which = 0:  A12 = C * Z  with C = U_1 * B12 (n1 x r2), Z = U_2^T (r2 x n2)
//   which = 1:  A21 = C * Z  with C = U_2 * B21 (n2 x r1), Z = U_1^T (r1 x n1)
// -- `Uint = generators(S.A11)[1] * S.B12`, `Vbnd = generators(S.A22)[2]` of src/factorization.jl:129-132.
template <class T>
static void offdiag_impl(HssT<T>& H, int which, T* C_, int ldc, T* Z, int ldz) {
  if (H.nd[0].left < 0) {
    hs_set_error(HS_ERR_HSS_LEAF, 0, "One of the Schur complements turned into a leaf. Aborting.");
    throw (int)HS_ERR_HSS_LEAF;
  }
  hipStream_t s = H.s;
  Pool tmp(global_cache());
  const int a = which == 0 ? H.nd[0].left : H.nd[0].right, b = which == 0 ? H.nd[0].right : H.nd[0].left;
  const int na = H.nd[a].hi - H.nd[a].lo, nb = H.nd[b].hi - H.nd[b].lo, ra = H.nd[a].r, rb = H.nd[b].r;
  if (ldc < na || ldz < rb) {
    hs_set_error(HS_ERR_DIMENSION, 0, "DimensionMismatch: hs_hss_offdiag needs ldc >= %d, ldz >= %d", na, rb);
    throw (int)HS_ERR_DIMENSION;
  }
  T* Ua = tmp.get<T>((size_t)ev(na) * std::max(ra, 1));
  T* Ub = tmp.get<T>((size_t)ev(nb) * std::max(rb, 1));
  hss_basis<T>(H, a, Ua, ev(na));
  hss_basis<T>(H, b, Ub, ev(nb));
  const T* Bc = which == 0 ? H.nd[0].B12 : H.nd[0].B21;
  const int ldb = which == 0 ? H.nd[0].ld12 : H.nd[0].ld21;
  HSS_HIP(hipMemset2DAsync(C_, sizeof(T) * ldc, 0, sizeof(T) * na, rb, s));
  std::vector<GemmProb<T>> g{GemmProb<T>{Ua, Bc, C_, na, rb, ra, ev(na), ldb, ldc}};
  run_gemms(tmp, g, 0, s);
  std::vector<SubJob<T>> t{SubJob<T>{Ub, ev(nb), nullptr, nullptr, 0, 0, nb, rb, Z, ldz, 1}};
  run_subs(tmp, t, s);
  HSS_HIP(hipStreamSynchronize(s));
}
// Batched compression of `count` matrices B_b - C_b*M_b*Z_b (device operands), each with its own permutation, first split and leaf size:
// ONE forest compressed level by level (compress_fixed), then one view per matrix that shares the forest's generators and keeps it alive.
// atol, rtol, pad, level_scale are those of the first options block; the number of samples is common (the largest kest, doubled until
// every matrix passes the rank rule).
template <class T>
static void compress_multi_impl(int64_t count, const int64_t* n, const T* const* A, const int64_t* lda, const LruArgs* la, const int64_t* const* perm,
                                const hs_hss_options* const* opts, void* stream, hs_hss** out) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    hs_set_error(HS_ERR_DEVICE, 0, "no HIP device available (the HSS module has no CPU fallback)");
    throw (int)HS_ERR_DEVICE;
  }
  if (count <= 0 || !n || !A || !lda || !opts || !out) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: incomplete batched compression");
    throw (int)HS_ERR_ARGUMENT;
  }
  int64_t ntot = 0;
  for (int64_t b = 0; b < count; ++b) {
    if (n[b] <= 0 || !A[b] || lda[b] < n[b] || !opts[b] || opts[b]->leafsize < 1 || opts[b]->first_split < 0 || opts[b]->first_split > n[b]) {
      hs_set_error(HS_ERR_ARGUMENT, b, "ArgumentError: matrix %lld of a batched compression needs n > 0, lda >= n, options", (long long)b);
      throw (int)HS_ERR_ARGUMENT;
    }
    ntot += n[b];
  }
  if (ntot > (1 << 30)) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: batched compression too large");
    throw (int)HS_ERR_ARGUMENT;
  }
  hs_hss_options opt = *opts[0];
  if (opt.pad <= 0) opt.pad = 8;
  if (!(opt.level_scale > 0.0) || opt.level_scale > 1.0) opt.level_scale = 0.5;
  std::shared_ptr<HssT<T>> H(new HssT<T>());
  H->n = (int)ntot;
  H->opt = opt;
  if (stream) {
    H->s = (hipStream_t)stream;
  } else {
    HSS_HIP(hipStreamCreate(&H->s));
    H->own_stream = true;
  }
  std::vector<CBlock<T>> cb((size_t)count);
  std::vector<std::vector<int>> hps((size_t)count);
  int* dperm = H->permpool.template get<int>((size_t)ntot);
  int64_t off = 0, kest = 8, nmax = 0;
  for (int64_t b = 0; b < count; ++b) {
    CBlock<T>& B = cb[(size_t)b];
    B.A = A[b]; B.lda = (int)lda[b]; B.n = (int)n[b];
    B.leafsize = (int)opts[b]->leafsize; B.first_split = (int)opts[b]->first_split;
    B.seed = (uint64_t)opts[b]->seed;
    kest = std::max<int64_t>(kest, opts[b]->kest > 0 ? opts[b]->kest : 64);
    nmax = std::max(nmax, n[b]);
    if (perm && perm[b]) {
      std::vector<int>& hp = hps[(size_t)b];
      hp.resize((size_t)n[b]);
      std::vector<char> seen((size_t)n[b], 0);
      for (int64_t i = 0; i < n[b]; ++i) {
        if (perm[b][i] < 0 || perm[b][i] >= n[b] || seen[(size_t)perm[b][i]]) {
          hs_set_error(HS_ERR_ARGUMENT, i, "ArgumentError: perm is not a permutation of 0..n-1 (entry %lld)", (long long)i);
          throw (int)HS_ERR_ARGUMENT;
        }
        seen[(size_t)perm[b][i]] = 1;
        hp[(size_t)i] = (int)perm[b][i];
      }
      HSS_HIP(hipMemcpyAsync(dperm + off, hp.data(), sizeof(int) * (size_t)n[b], hipMemcpyHostToDevice, H->s));
      HSS_HIP(hipStreamSynchronize(H->s));
      B.perm = dperm + off;
      B.hperm = hp.data();
    }
    if (la && la[b].r1 > 0 && la[b].r2 > 0) {
      const LruArgs& a = la[b];
      const bool m_id = !a.M && a.r1 == a.r2;
      if (!a.C || (!a.M && !m_id) || !a.Z || a.ldc < n[b] || (a.M && a.ldm < a.r1) || a.ldz < a.r2 || a.r1 > n[b] || a.r2 > n[b]) {
        hs_set_error(HS_ERR_DIMENSION, b, "DimensionMismatch: the low-rank update needs C (n x r1), M (r1 x r2), Z (r2 x n)");
        throw (int)HS_ERR_DIMENSION;
      }
      B.lru.C = (const T*)a.C; B.lru.ldc = (int)a.ldc;
      B.lru.M = (const T*)a.M; B.lru.ldm = (int)a.ldm;
      B.lru.Z = (const T*)a.Z; B.lru.ldz = (int)a.ldz;
      B.lru.r1 = (int)a.r1; B.lru.r2 = (int)a.r2;
    }
    off += n[b];
  }
  auto t0 = std::chrono::steady_clock::now();
  int k = (int)std::min<int64_t>(kest, nmax);
  for (;;) {
    if (compress_fixed<T>(*H, cb, k, nullptr)) break;
    if (k >= nmax) break;
    k = (int)std::min<int64_t>(2 * (int64_t)k, nmax);
  }
  const double tc = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int64_t b = 0; b < count; ++b) out[b] = nullptr;
  try {
    for (int64_t b = 0; b < count; ++b) {
      HssT<T>* V = child_impl<T>(H.get(), 2, cb[(size_t)b].root);
      V->hold = H;
      V->opt = *opts[b];
      V->opt.pad = opt.pad;
      V->opt.level_scale = opt.level_scale;
      V->k = H->k;
      V->t_compress = tc / (double)count;
      if (cb[(size_t)b].perm) {
        V->perm = const_cast<int*>(cb[(size_t)b].perm);  // lives in the forest's pool
        V->hperm = hps[(size_t)b];
        V->hinvperm.assign((size_t)n[b], 0);
        for (int64_t i = 0; i < n[b]; ++i) V->hinvperm[(size_t)V->hperm[(size_t)i]] = (int)i;
      }
      out[b] = new hs_hss{sizeof(T) == 16, V};
    }
  } catch (...) {
    for (int64_t b = 0; b < count; ++b)
      if (out[b]) {
        hs_hss_free(out[b]);
        out[b] = nullptr;
      }
    throw;
  }
}

extern "C" int hs_hss_compress_lru_multi_d(int64_t count, const int64_t* n, const double* const* B, const int64_t* ldb, const double* const* C_, const int64_t* ldc,
                                           const double* const* M, const int64_t* ldm, const double* const* Z, const int64_t* ldz, const int64_t* r1, const int64_t* r2,
                                           const int64_t* const* perm, const hs_hss_options* const* opts, void* stream, hs_hss** out) {
  if (!out || count <= 0) return HS_ERR_ARGUMENT;
  HSS_GUARD(
      std::vector<LruArgs> la((size_t)count);
      for (int64_t b = 0; b < count; ++b)
        if (r1 && r2 && r1[b] > 0 && r2[b] > 0) la[(size_t)b] = LruArgs{C_[b], M ? M[b] : nullptr, Z[b], ldc[b], ldm ? ldm[b] : 0, ldz[b], r1[b], r2[b]};
      compress_multi_impl<double>(count, n, B, ldb, la.data(), perm, opts, stream, out));
}
extern "C" int hs_hss_compress_lru_multi_z(int64_t count, const int64_t* n, const double* const* B, const int64_t* ldb, const double* const* C_, const int64_t* ldc,
                                           const double* const* M, const int64_t* ldm, const double* const* Z, const int64_t* ldz, const int64_t* r1, const int64_t* r2,
                                           const int64_t* const* perm, const hs_hss_options* const* opts, void* stream, hs_hss** out) {
  if (!out || count <= 0) return HS_ERR_ARGUMENT;
  HSS_GUARD(
      std::vector<LruArgs> la((size_t)count);
      for (int64_t b = 0; b < count; ++b)
        if (r1 && r2 && r1[b] > 0 && r2[b] > 0) la[(size_t)b] = LruArgs{C_[b], M ? M[b] : nullptr, Z[b], ldc[b], ldm ? ldm[b] : 0, ldz[b], r1[b], r2[b]};
      compress_multi_impl<cplx>(count, n, (const cplx* const*)B, ldb, la.data(), perm, opts, stream, out));
}

extern "C" int hs_hss_expand(hs_hss* H, double* out, int64_t ldo, int where) {
  if (!H || !out || ldo < hs_hss_size(H)) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_expand needs an n x n result (ldo >= n)");
    return HS_ERR_ARGUMENT;
  }
  HSS_GUARD(
      if (where != 0) {
        if (H->is_complex) hss_expand<cplx>(*HZ(H), (cplx*)out, (int)ldo);
        else hss_expand<double>(*HD(H), out, (int)ldo);
      } else {
        Pool st(global_cache());
        const int n = (int)hs_hss_size(H), ld = ev(n);
        if (H->is_complex) {
          cplx* d = st.get<cplx>((size_t)ld * n);
          hss_expand<cplx>(*HZ(H), d, ld);
          HSS_HIP(hipMemcpy2D(out, sizeof(cplx) * ldo, d, sizeof(cplx) * ld, sizeof(cplx) * n, n, hipMemcpyDeviceToHost));
        } else {
          double* d = st.get<double>((size_t)ld * n);
          hss_expand<double>(*HD(H), d, ld);
          HSS_HIP(hipMemcpy2D(out, sizeof(double) * ldo, d, sizeof(double) * ld, sizeof(double) * n, n, hipMemcpyDeviceToHost));
        }
      });
}

extern "C" int hs_hss_offdiag(hs_hss* H, int which, double* C_, int64_t ldc, double* Z, int64_t ldz, int where) {
  if (!H || !C_ || !Z || (which != 0 && which != 1)) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_offdiag needs which in {0, 1} and two output blocks");
    return HS_ERR_ARGUMENT;
  }
  HSS_GUARD(
      int64_t ia[8]; int64_t ib[8];
      const int64_t l = HSS_DISPATCH(H, HD(H)->nd[0].left, HZ(H)->nd[0].left), r = HSS_DISPATCH(H, HD(H)->nd[0].right, HZ(H)->nd[0].right);
      if (l < 0) {
        hs_set_error(HS_ERR_HSS_LEAF, 0, "One of the Schur complements turned into a leaf. Aborting.");
        throw (int)HS_ERR_HSS_LEAF;
      }
      (void)hs_hss_node_info(H, which == 0 ? l : r, ia); (void)hs_hss_node_info(H, which == 0 ? r : l, ib);
      const int na = (int)(ia[1] - ia[0]), nb = (int)(ib[1] - ib[0]), rb = (int)ib[6];
      if (where != 0) {
        if (H->is_complex) offdiag_impl<cplx>(*HZ(H), which, (cplx*)C_, (int)ldc, (cplx*)Z, (int)ldz);
        else offdiag_impl<double>(*HD(H), which, C_, (int)ldc, Z, (int)ldz);
      } else {
        Pool st(global_cache());
        const int lc = ev(na), lz = ev(rb);
        const size_t esz = H->is_complex ? 16 : 8;
        void* dC = st.get<char>((size_t)lc * std::max(rb, 1) * esz);
        void* dZ = st.get<char>((size_t)lz * std::max(nb, 1) * esz);
        if (H->is_complex) offdiag_impl<cplx>(*HZ(H), which, (cplx*)dC, lc, (cplx*)dZ, lz);
        else offdiag_impl<double>(*HD(H), which, (double*)dC, lc, (double*)dZ, lz);
        if (rb > 0) {
          HSS_HIP(hipMemcpy2D(C_, esz * ldc, dC, esz * lc, esz * na, rb, hipMemcpyDeviceToHost));
          HSS_HIP(hipMemcpy2D(Z, esz * ldz, dZ, esz * lz, esz * rb, nb, hipMemcpyDeviceToHost));
        }
      });
}

// ---- operators made of HSS blocks and sparse couplings (hs_hss_op.h) behind the C ABI --------------------------------------------
template <class T>
static void make_blockop(const hs_hss_blockop* d, BlockOp<T>& op) {
  if (!d || d->n1 < 0 || d->n2 < 0 || d->n1 + d->n2 <= 0 || !d->gid || !d->A || !d->lpos || (d->n1 > 0 && !d->H1) || (d->n2 > 0 && !d->H2)) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: incomplete hs_hss_blockop");
    throw (int)HS_ERR_ARGUMENT;
  }
  op.n1 = (int)d->n1;
  op.n = (int)(d->n1 + d->n2);
  auto blk = [&](hs_hss* H, int64_t sz) -> HssT<T>* {
    if (sz == 0) return nullptr;
    if ((H->is_complex != 0) != (sizeof(T) == 16) || hs_hss_size(H) != sz) {
      hs_set_error(HS_ERR_DIMENSION, 0, "DimensionMismatch: diagonal block of the operator has size %lld, expected %lld", (long long)hs_hss_size(H), (long long)sz);
      throw (int)HS_ERR_DIMENSION;
    }
    return (HssT<T>*)H->impl;
  };
  op.H1 = blk(d->H1, d->n1);
  op.H2 = blk(d->H2, d->n2);
  op.hgid.resize((size_t)op.n);
  for (int i = 0; i < op.n; ++i) {
    if (d->gid[i] < 0 || d->gid[i] >= d->A->n) {
      hs_set_error(HS_ERR_DIMENSION, i, "BoundsError: global id %lld outside 0:%lld", (long long)d->gid[i], (long long)d->A->n - 1);
      throw (int)HS_ERR_DIMENSION;
    }
    op.hgid[(size_t)i] = (int)d->gid[i];
  }
  op.lpos = d->lpos;
  op.A.n = d->A->n;
  op.A.colptr = d->A->colptr; op.A.rowval = d->A->rowval; op.A.nz = (const T*)d->A->nzval;
  op.A.rowptr = d->A->rowptr; op.A.colind = d->A->colind; op.A.nzr = (const T*)d->A->nzval_r;
}
template <class T>
static hs_hss* compress_blockop(const hs_hss_blockop* d, const LruArgs& la, const int64_t* perm, const hs_hss_options* o, void* stream) {
  BlockOp<T> op;
  make_blockop<T>(d, op);
  return new hs_hss{sizeof(T) == 16, compress_impl<T>(op.n, nullptr, 0, 1, o, perm, stream, &la, &op)};
}
extern "C" int hs_hss_compress_blockop_d(const hs_hss_blockop* op, const double* C_, int64_t ldc, const double* M, int64_t ldm, const double* Z, int64_t ldz,
                                         int64_t r1, int64_t r2, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  LruArgs la{C_, M, Z, ldc, ldm, ldz, r1, r2};
  HSS_GUARD(*out = compress_blockop<double>(op, la, perm, o, stream));
}
extern "C" int hs_hss_compress_blockop_z(const hs_hss_blockop* op, const double* C_, int64_t ldc, const double* M, int64_t ldm, const double* Z, int64_t ldz,
                                         int64_t r1, int64_t r2, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  LruArgs la{C_, M, Z, ldc, ldm, ldz, r1, r2};
  HSS_GUARD(*out = compress_blockop<cplx>(op, la, perm, o, stream));
}
// Y = Op * X (trans != 0: Op^T * X) for device blocks of nrhs columns in the operator's index order
template <class T>
static void blockop_apply(const hs_hss_blockop* d, const T* X, int ldx, T* Y, int ldy, int q, int trans, hipStream_t s) {
  BlockOp<T> op;
  make_blockop<T>(d, op);
  op.begin(s);
  try {
    HSS_HIP(hipMemset2DAsync(Y, sizeof(T) * ldy, 0, sizeof(T) * op.n, q, s));
    op.mul(X, ldx, Y, ldy, q, trans != 0, s);
  } catch (...) {
    op.end(s);
    throw;
  }
  op.end(s);
}
extern "C" int hs_hss_blockop_apply(const hs_hss_blockop* op, int is_complex, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t nrhs, int trans, void* stream) {
  if (!op || !X || !Y || nrhs < 0 || X == Y) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_blockop_apply needs distinct device blocks X and Y");
    return HS_ERR_ARGUMENT;
  }
  if (nrhs == 0) return HS_OK;
  HSS_GUARD(if (is_complex) blockop_apply<cplx>(op, (const cplx*)X, (int)ldx, (cplx*)Y, (int)ldy, (int)nrhs, trans, (hipStream_t)stream);
            else blockop_apply<double>(op, X, (int)ldx, Y, (int)ldy, (int)nrhs, trans, (hipStream_t)stream));
}

// Give the recycled device and pinned blocks of this module and of the low-rank compressions back to the driver (they are kept across
// factorizations: up to 24 + 16 GiB of HBM); returns the device bytes released.  Nothing may be running on the library's streams.
extern "C" int64_t hs_hss_trim(void) {
  int64_t freed = 0;
  {
    BlockCache* c = global_cache();
    std::lock_guard<std::mutex> lk(c->mu);
    for (auto& kv : c->free_) {
      (void)hipFree(kv.second);
      freed += (int64_t)kv.first;
    }
    c->free_.clear();
    c->held = 0;
  }
  {
    PinnedCache* pc = pinned_cache();
    std::lock_guard<std::mutex> lk(pc->mu);
    for (auto& kv : pc->free_) (void)hipHostFree(kv.second);
    pc->free_.clear();
  }
  return freed + hs_lr_trim();
}

extern "C" int hs_hss_factor(hs_hss* H) {
  if (!H) return HS_ERR_ARGUMENT;
  HSS_GUARD(if (H->is_complex) hss_factor<cplx>(*HZ(H)); else hss_factor<double>(*HD(H)));
}

extern "C" int hs_hss_ldiv(hs_hss* H, double* B, int64_t ldb, int64_t nrhs, int where) {
  if (!H || !B || nrhs < 0) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_hss_ldiv needs a right-hand side");
    return HS_ERR_ARGUMENT;
  }
  if (nrhs == 0) return HS_OK;
  HSS_GUARD(
      if (H->is_complex) {
        hss_factor<cplx>(*HZ(H));
        with_device_block<cplx>(HZ(H)->n, (const cplx*)B, ldb, (cplx*)B, ldb, (int)nrhs, where,
                                [&](const cplx*, int la, cplx* b, int) { hss_ldiv<cplx>(*HZ(H), b, la, (int)nrhs); });
      } else {
        hss_factor<double>(*HD(H));
        with_device_block<double>(HD(H)->n, B, ldb, B, ldb, (int)nrhs, where,
                                  [&](const double*, int la, double* b, int) { hss_ldiv<double>(*HD(H), b, la, (int)nrhs); });
      });
}

extern "C" void hs_hss_free(hs_hss* H) {
  if (!H) return;
  if (H->is_complex)
    delete HZ(H);
  else
    delete HD(H);
  delete H;
}
