// hs_lowrank.hip -- low-rank compression of a dense block on the device: X (rows x cols) ~= C * Z.
//
// Role in the reference: the Gauss transforms of a compressed front are LowRankMatrix objects built
// with LowRankApprox.pqrfact (tolerance-stopped pivoted QR, optionally on a Gaussian sketch):
// `_lgauss_transform` / `_rgauss_transform` (src/factorization.jl:171-209), tolerances 0.5*atol,
// 0.5*rtol (:99-100).  LowRankApprox is not part of the reference tree (parity unpinned).
//
// Here the factorization is a randomized ROW interpolative decomposition that reuses the front
// kernels (MFMA GEMM, tournament-pivoted LU, TRSM):
//   1. sketch   Y = X * Omega,  Omega cols x k Gaussian                      (gemm kernel)
//   2. P*Y = L*U with pivot candidates over ALL rows                         (lu_rec, pivrows = rows)
//   3. rank r = 1 + last j with |u_jj| > max(atol, rtol*|u_11|); if r is within 8 of k, k doubles
//   4. the first r pivot rows form the skeleton:  X ~= P' * L[:, :r] * ( L[:r,:r]^-1 * (P*X)[:r, :] )
//      => C = P' * L[:, :r] (rows x r, unit-lower-trapezoidal, kept inside the packed L\U of Y)
//         Z = L11^-1 * (P*X)[:r, :]  (laswp + TRSM on X itself, then the top r rows are copied out)
// X is overwritten (it is replaced by its compressed form).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "hs_sched.h"
#include "hs_lowrank.h"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
__device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline double gauss_from(uint64_t a, uint64_t b) {
  double u1 = ((a >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0, 1]
  double u2 = (b >> 11) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}
__global__ __launch_bounds__(256) void randn_kernel(double* out, size_t n, uint64_t seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t h = splitmix64(seed ^ (i * 0xD1342543DE82EF95ull));
  out[i] = gauss_from(h, splitmix64(h));
}

template <class T>
__global__ __launch_bounds__(64) void absdiag_kernel(const T* __restrict__ LU, int ld, int k, double* __restrict__ out) {
  int j = blockIdx.x * 64 + threadIdx.x;
  if (j < k) out[j] = Scal<T>::abs1(LU[(size_t)j + (size_t)j * ld]);
}

template <class T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int r, int cols) {
  int c = blockIdx.x;
  for (int i = threadIdx.x; i < r; i += 256) dst[(size_t)i + (size_t)c * ldd] = src[(size_t)i + (size_t)c * lds];
  (void)cols;
}

// ------------------------------------------------------------------------------------------------
// driver
// ------------------------------------------------------------------------------------------------
extern "C" int64_t hs_arena_trim(void);
namespace {
struct LrCache {
  std::mutex mu;
  std::multimap<size_t, void*> free_;
  std::unordered_map<void*, size_t> live;
  size_t held = 0;
  static constexpr size_t limit = (size_t)16 << 30, max_block = (size_t)512 << 20;
};
LrCache* lr_cache() {
  static LrCache* c = new LrCache();  // never destroyed: a static destructor would run after the HIP runtime has shut down
  return c;
}
size_t lr_class(size_t bytes) {
  size_t cls = 4096;
  while (cls < bytes) cls <<= 1;
  if (cls > 16384 && cls - cls / 4 >= bytes) cls -= cls / 4;
  return cls;
}
}  // namespace
int hs_lr_alloc(void** out, size_t bytes) {
  *out = nullptr;
  LrCache* c = lr_cache();
  const bool cached = bytes <= LrCache::max_block;
  const size_t cls = cached ? lr_class(bytes ? bytes : 256) : bytes;
  if (cached) {
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->free_.find(cls);
    if (it != c->free_.end()) {
      *out = it->second;
      c->free_.erase(it);
      c->held -= cls;
      c->live[*out] = cls;
      return 0;
    }
  }
  hipError_t e = hipMalloc(out, cls);
  if (e != hipSuccess) {  // give the cached blocks back to the driver and try once more
    (void)hipGetLastError();
    {
      std::lock_guard<std::mutex> lk(c->mu);
      for (auto& kv : c->free_) (void)hipFree(kv.second);
      c->free_.clear();
      c->held = 0;
    }
    e = hipMalloc(out, cls);
    if (e != hipSuccess) {  // last: the arenas hs_free parked (hs_api.hip)
      (void)hipGetLastError();
      (void)hs_arena_trim();
      e = hipMalloc(out, cls);
    }
    if (e != hipSuccess) {
      (void)hipGetLastError();
      *out = nullptr;
      return (int)e;
    }
  }
  if (cached) {
    std::lock_guard<std::mutex> lk(c->mu);
    c->live[*out] = cls;
  }
  return 0;
}
void hs_lr_free(void* p) {
  if (!p) return;
  LrCache* c = lr_cache();
  {
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->live.find(p);
    if (it != c->live.end()) {
      const size_t cls = it->second;
      c->live.erase(it);
      if (c->held + cls <= LrCache::limit) {
        c->free_.insert({cls, p});
        c->held += cls;
        return;
      }
    }
  }
  (void)hipFree(p);
}

int64_t hs_lr_trim() {
  LrCache* c = lr_cache();
  std::lock_guard<std::mutex> lk(c->mu);
  int64_t freed = 0;
  for (auto& kv : c->free_) {
    (void)hipFree(kv.second);
    freed += (int64_t)kv.first;
  }
  c->free_.clear();
  c->held = 0;
  return freed;
}

template <class T>
void lowrank_free(LowRank<T>& lr) {
  hs_lr_free(lr.Lp);
  hs_lr_free(lr.Z);
  hs_lr_free(lr.rperm);
  hs_lr_free(lr.Cd);
  hs_lr_free(lr.Y0);
  lr = LowRank<T>();
}

#define LR_HIP(call)                          \
  do {                                        \
    hipError_t e__ = (call);                  \
    if (e__ != hipSuccess) {                  \
      hs_set_error(-6, 0, "%s failed: %s", #call, hipGetErrorString(e__)); \
      return -6;                              \
    }                                         \
  } while (0)

template <class T>
int lowrank_compress(T* X, int ldx, int rows, int cols, double atol, double rtol, int kinit, uint64_t seed, hipStream_t s, LowRank<T>* out) {
  *out = LowRank<T>();
  out->rows = rows;
  out->cols = cols;
  if (rows <= 0 || cols <= 0) return 0;
  const int kmax = std::min(rows, cols);
  int k = std::max(32, std::min(kinit > 0 ? kinit : 128, kmax));
  k = std::min(k, kmax);
  Profiler prof;  // off
  for (int attempt = 0;; ++attempt) {
    const int ldp = (rows + 1) / 2 * 2;
    const int nblk = (k + HS_PB - 1) / HS_PB;
    const int ncand = ((rows + HS_CHUNK - 1) / HS_CHUNK + 1) * HS_PB;
    T *Y = nullptr, *Om = nullptr, *inv = nullptr;
    int* ints = nullptr;
    double* dd = nullptr;
    NodeDesc<T>* dn = nullptr;
    GemmProb<T>* dp = nullptr;
    LR_HIP(hipMalloc((void**)&Y, sizeof(T) * ((size_t)ldp * k + 32)));
    LR_HIP(hipMalloc((void**)&Om, sizeof(T) * ((size_t)cols * k + 32)));
    LR_HIP(hipMalloc((void**)&inv, sizeof(T) * (size_t)2 * nblk * HS_PB * HS_PB));
    LR_HIP(hipMalloc((void**)&ints, sizeof(int) * ((size_t)k + rows + 2 * ncand + HS_PB + 1)));
    LR_HIP(hipMalloc((void**)&dd, sizeof(double) * k));
    LR_HIP(hipMalloc((void**)&dn, sizeof(NodeDesc<T>)));
    LR_HIP(hipMalloc((void**)&dp, sizeof(GemmProb<T>)));
    LR_HIP(hipMemsetAsync(ints, 0, sizeof(int) * ((size_t)k + rows + 2 * ncand + HS_PB + 1), s));
    const size_t nrand = (size_t)cols * k * (sizeof(T) / 8);
    hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((nrand + 255) / 256)), dim3(256), 0, s, (double*)Om, nrand, seed + 0x1000ull * attempt);
    GemmProb<T> gp{X, Om, Y, rows, k, cols, ldx, cols, ldp};
    LR_HIP(hipMemcpyAsync(dp, &gp, sizeof gp, hipMemcpyHostToDevice, s));
    launch_gemm_probs<T>(dp, 1, rows, k, 0, s);
    NodeDesc<T> d;
    memset(&d, 0, sizeof d);
    d.LF = Y;
    d.UR = X;
    d.SB = nullptr;
    d.invL = inv;
    d.invU = inv + (size_t)nblk * HS_PB * HS_PB;
    d.ipiv = ints;
    d.rperm = d.ipiv + k;
    d.cand0 = d.rperm + rows;
    d.cand1 = d.cand0 + ncand;
    d.pivlist = d.cand1 + ncand;
    d.info = d.pivlist + HS_PB;
    d.ni = k; d.nb = rows - k; d.m = rows;
    d.ldl = ldp; d.ldu = ldx; d.lds = 2;
    d.ni1 = k; d.nb1 = rows - k; d.isleaf = 1; d.node = 0;
    d.pivrows = rows;
    d.finalize();
    d.mrows[1] = k;   // "UR" = X: the TRSM touches its first k rows, the swaps reach every row
    d.mcols[1] = cols;
    d.mrows[2] = 0;
    d.mcols[2] = 0;
    LR_HIP(hipMemcpyAsync(dn, &d, sizeof d, hipMemcpyHostToDevice, s));
    LR_HIP(hipStreamSynchronize(s));  // gp / d are stack objects
    launch_init_fronts<T>(dn, 1, rows, s);
    Sched<T> sch{dn, 1, k, cols, rows, s, &prof, nullptr, nullptr, nullptr, rows};
    int P2 = HS_PB;
    while (P2 < k) P2 *= 2;
    sch.lu_rec(0, P2);
    hipLaunchKernelGGL(absdiag_kernel<T>, dim3((k + 63) / 64), dim3(64), 0, s, Y, ldp, k, dd);
    std::vector<double> hd(k);
    LR_HIP(hipMemcpyAsync(hd.data(), dd, sizeof(double) * k, hipMemcpyDeviceToHost, s));
    LR_HIP(hipStreamSynchronize(s));
    const double tau = std::max(atol, rtol * hd[0]);
    int r = 0;
    for (int j = 0; j < k; ++j)
      if (hd[j] > tau) r = j + 1;
    const bool grow = (r + 8 > k) && (k < kmax);
    if (grow) {
      (void)hipFree(Y); (void)hipFree(Om); (void)hipFree(inv); (void)hipFree(ints); (void)hipFree(dd); (void)hipFree(dn); (void)hipFree(dp);
      k = std::min(2 * k, kmax);
      continue;
    }
    // Z = L11^-1 * (P*X)[:k, :]  on X itself, then keep the first r rows
    sch.laswp(HS_MAT_UR, 0, HS_BIG, 0, P2);
    sch.trsm_rec(HS_MAT_UR, 0, P2, 0, HS_BIG);
    T* Z = nullptr;
    const int ldz = std::max(2, (r + 1) / 2 * 2);
    LR_HIP(hipMalloc((void**)&Z, sizeof(T) * ((size_t)ldz * cols + 32)));
    if (r > 0) hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(cols), dim3(256), 0, s, (const T*)X, ldx, Z, ldz, r, cols);
    int* rp = nullptr;
    LR_HIP(hipMalloc((void**)&rp, sizeof(int) * rows));
    LR_HIP(hipMemcpyAsync(rp, d.rperm, sizeof(int) * rows, hipMemcpyDeviceToDevice, s));
    LR_HIP(hipStreamSynchronize(s));
    (void)hipFree(Om); (void)hipFree(inv); (void)hipFree(ints); (void)hipFree(dd); (void)hipFree(dn); (void)hipFree(dp);
    out->Lp = Y;
    out->ldp = ldp;
    out->k = k;
    out->r = r;
    out->rperm = rp;
    out->Z = Z;
    out->ldz = ldz;
    return 0;
  }
}

template int lowrank_compress<double>(double*, int, int, int, double, double, int, uint64_t, hipStream_t, LowRank<double>*);
template int lowrank_compress<cplx>(cplx*, int, int, int, double, double, int, uint64_t, hipStream_t, LowRank<cplx>*);
template void lowrank_free<double>(LowRank<double>&);
template void lowrank_free<cplx>(LowRank<cplx>&);
