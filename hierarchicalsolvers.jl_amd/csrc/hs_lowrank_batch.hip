// hs_lowrank_batch.hip -- the randomized row ID of hs_lowrank.hip for a BATCH of independent blocks.
//
// Same algorithm per block (sketch Y = X*Omega, pivoted LU of Y over all rows, rank from |u_jj|, Z = L11^-1 (P X)[:r,:]),
// but every stage is one grouped launch over the blocks: the pivoted LU of a sketch is a chain of ~10 dependent
// tiny kernels per 32 columns, and 16 such chains run back to back (the Gauss transforms of 8 fronts) were the
// largest part of a compressed level.  Blocks whose rank reaches their sketch width are redone with twice the
// width while the others wait (they are finished); `kinit` of a block normally comes from the rank it had in the
// previous factorization, so the usual number of passes is one.
//
// Role in the reference: `pqrfact` inside `_lgauss_transform` / `_rgauss_transform` (src/factorization.jl:171-182).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <cmath>
#include <vector>

#include "hs_sched.h"
#include "hs_lowrank.h"

namespace {

__device__ inline uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
template <class T>
__device__ inline T from_scale(double a);
template <>
__device__ inline double from_scale<double>(double a) { return a; }
template <>
__device__ inline cplx from_scale<cplx>(double a) { return {a, 0.0}; }
__global__ __launch_bounds__(256) void randn_fill_kernel(double* out, size_t n, uint64_t seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint64_t a = mix64(seed ^ (i * 0xD1342543DE82EF95ull)), b = mix64(a);
  double u1 = ((a >> 11) + 1.0) * (1.0 / 9007199254740993.0);  // (0, 1]
  double u2 = (b >> 11) * (1.0 / 9007199254740992.0);
  out[i] = sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

// ------------------------------------------------------------------------------------------------
// Sparse sign sketch Y = X * Omega for LARGE blocks (the Gauss transforms of a compressed front: X is nb x ni, 12,544 x 12,320 complex at
// level 2 of Helmholtz 112^3): Omega is ni x k with HS_SKETCH_ZETA (8) nonzeros +-sqrt(k / zeta) per ROW -- every column of X is added, with a
// random sign, into zeta of the k sketch columns (Clarkson-Woodruff / sparse Johnson-Lindenstrauss embeddings; Martinsson & Tropp 2020,
// section 9.2: as good as a Gaussian test matrix for range finding from zeta ~ 8 on).  A Gaussian Omega costs 2 m n k flops in the MFMA GEMM --
// 1.6e12 for that block, 110 of the 204 ms the level spends compressing Aib / Abi -- the sparse one reads X zeta times: m n zeta adds,
// HBM / L2-bound (20 GB, ~5 ms).  Bucket of column j in repetition t: pi_t(j) mod k with pi_t(j) = (a_t j + b_t) mod n, gcd(a_t, n) = 1 (a
// permutation of the columns, 2-universal); the kernel is gather-form -- one workgroup per (256 rows, sketch column c) walks
// pi_t^-1(c), pi_t^-1(c + k), ... -- so it needs no atomics and is deterministic.  E |x Omega|^2 = k |x|^2 like the Gaussian sketch's.
// ------------------------------------------------------------------------------------------------
#define HS_SKETCH_ZETA 8
template <class T>
struct SparseSketchJob {
  const T* X;
  T* Y;
  int ldx, ldy, rows, cols, k;
  unsigned long long seed;
  unsigned int ainv[HS_SKETCH_ZETA], b[HS_SKETCH_ZETA], step[HS_SKETCH_ZETA];  // step = ainv * k mod cols: pi^-1(p + k) = pi^-1(p) + step (mod cols)
  double scale;
};
template <class T>
__global__ __launch_bounds__(256) void sparse_sketch_kernel(const SparseSketchJob<T>* __restrict__ jobs) {
  const SparseSketchJob<T> J = jobs[blockIdx.z];
  const int c = blockIdx.y;
  const int row = blockIdx.x * 256 + threadIdx.x;
  if (c >= J.k || (int)blockIdx.x * 256 >= J.rows) return;  // workgroup-uniform
  const bool live = row < J.rows;
  const T* xr = J.X + (size_t)(live ? row : 0);
  T acc = Scal<T>::zero();
  const unsigned long long n = (unsigned long long)J.cols;
#pragma unroll 1
  for (int t = 0; t < HS_SKETCH_ZETA; ++t) {
    const unsigned long long ai = J.ainv[t], bt = J.b[t], st = J.step[t];
    unsigned long long j = (ai * (((unsigned long long)c + n - bt) % n)) % n;  // pi_t^-1(c); the walk below needs no division (uniform over the workgroup)
#pragma unroll 8
    for (int p = c; p < J.cols; p += J.k) {  // positions c, c + k, ... of the permuted order: the columns of bucket c
      const unsigned long long h = mix64(J.seed ^ ((unsigned long long)(t + 1) << 48) ^ j);
      const T v = gld(xr + (size_t)j * J.ldx);
      acc = (h & 1ull) ? acc + v : acc - v;
      j += st;
      j = j >= n ? j - n : j;
    }
  }
  if (live) J.Y[(size_t)row + (size_t)c * J.ldy] = acc * from_scale<T>(J.scale);
}

template <class T>
struct DiagJob {
  const T* LU;
  int ld, k;
  double* out;
};
template <class T>
__global__ __launch_bounds__(64) void absdiag_batch_kernel(const DiagJob<T>* __restrict__ jobs) {
  const DiagJob<T> j = jobs[blockIdx.y];
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c < j.k) j.out[c] = Scal<T>::abs1(j.LU[(size_t)c + (size_t)c * j.ld]);
}

template <class T>
__global__ __launch_bounds__(256) void copy_top_rows_kernel(const T* __restrict__ src, int lds, T* __restrict__ dst, int ldd, int r) {
  const int c = blockIdx.x;
  for (int i = threadIdx.x; i < r; i += 256) dst[(size_t)i + (size_t)c * ldd] = src[(size_t)i + (size_t)c * lds];
}

struct Guard {  // frees what it was given, whatever the exit path
  std::vector<void*> p;
  ~Guard() {
    for (void* q : p) hs_lr_free(q);
  }
  template <class U>
  hipError_t alloc(U** out, size_t bytes) {
    *out = nullptr;
    hipError_t e = (hipError_t)hs_lr_alloc((void**)out, bytes ? bytes : 256);
    if (e == hipSuccess) p.push_back(*out);
    return e;
  }
};

}  // namespace

#define LRB_HIP(call)                                                         \
  do {                                                                        \
    hipError_t e__ = (call);                                                  \
    if (e__ != hipSuccess) {                                                  \
      hs_set_error(-6, 0, "%s failed: %s", #call, hipGetErrorString(e__));    \
      return -6;                                                              \
    }                                                                         \
  } while (0)

template <class T>
int lowrank_compress_batch(LowRankJob<T>* jobs, int njobs, double atol, double rtol, hipStream_t s, bool need_z, bool keep_sketch, bool sketch_only) {
  std::vector<int> todo;
  for (int i = 0; i < njobs; ++i) {
    LowRankJob<T>& J = jobs[i];
    *J.out = LowRank<T>();
    J.out->rows = J.rows;
    J.out->cols = J.cols;
    if (J.rows <= 0 || J.cols <= 0) continue;
    const int kmax = std::min(J.rows, J.cols);
    J.k = std::min(std::max(32, J.kinit > 0 ? J.kinit : 128), kmax);
    todo.push_back(i);
  }
  Profiler prof;  // off
  for (int pass = 0; !todo.empty(); ++pass) {
    const int nj = (int)todo.size();
    Guard g;  // temporaries of this pass
    // ---- layout of the temporaries: Omega, inverse blocks, ints, diagonals ---------------------------------------
    std::vector<size_t> offOm(nj), offInv(nj), offInt(nj), offDg(nj);
    size_t nOm = 0, nInv = 0, nInt = 0, nDg = 0;
    int maxk = 0, maxrows = 0, maxcols = 0;
    // large blocks take the sparse sign sketch (sparse_sketch_kernel above); HS_SKETCH=gauss keeps the Gaussian one everywhere (diagnostics)
    static const bool sparse_on = !(getenv("HS_SKETCH") && getenv("HS_SKETCH")[0] == 'g');
    std::vector<char> sparse(nj, 0);
    for (int a = 0; a < nj; ++a) {
      const LowRankJob<T>& J = jobs[todo[a]];
      sparse[a] = sparse_on && (size_t)J.rows * J.cols >= ((size_t)1 << 22) && J.cols >= 4 * J.k && J.k >= 2 * HS_SKETCH_ZETA;
      const int nblk = (J.k + HS_PB - 1) / HS_PB, ncand = ((J.rows + 127) / 128 + 1) * HS_PB;
      offOm[a] = nOm;
      if (!sparse[a]) nOm += (size_t)J.cols * J.k + 32;
      offInv[a] = nInv;
      nInv += (size_t)2 * nblk * HS_PB * HS_PB;
      offInt[a] = nInt;
      nInt += (size_t)J.k + J.rows + 2 * ncand + HS_PB + 8;
      offDg[a] = nDg;
      nDg += (size_t)J.k;
      maxk = std::max(maxk, J.k);
      maxrows = std::max(maxrows, J.rows);
      maxcols = std::max(maxcols, J.cols);
    }
    T *dOm, *dInv;
    int* dInt;
    double* dDg;
    NodeDesc<T>* dn;
    GemmProb<T>* dgp;
    DiagJob<T>* ddj;
    LRB_HIP(g.alloc(&dOm, sizeof(T) * std::max<size_t>(nOm, 32)));
    LRB_HIP(g.alloc(&dInv, sizeof(T) * nInv));
    LRB_HIP(g.alloc(&dInt, sizeof(int) * nInt));
    LRB_HIP(g.alloc(&dDg, sizeof(double) * nDg));
    LRB_HIP(g.alloc(&dn, sizeof(NodeDesc<T>) * nj));
    LRB_HIP(g.alloc(&dgp, sizeof(GemmProb<T>) * nj));
    LRB_HIP(g.alloc(&ddj, sizeof(DiagJob<T>) * nj));
    LRB_HIP(hipMemsetAsync(dInt, 0, sizeof(int) * nInt, s));
    const size_t nrand = std::max<size_t>(nOm, 32) * (sizeof(T) / 8);
    hipLaunchKernelGGL(randn_fill_kernel, dim3((unsigned)((nrand + 255) / 256)), dim3(256), 0, s, (double*)dOm, nrand,
                       jobs[todo[0]].seed + 0x1000ull * pass);
    // ---- per block: the sketch Y (kept: it becomes the packed L\U that holds C), descriptors ------------------------
    std::vector<NodeDesc<T>> hn(nj);
    std::vector<GemmProb<T>> hgp(nj);
    std::vector<DiagJob<T>> hdj(nj);
    std::vector<T*> Y(nj, nullptr);
    std::vector<int> ldp(nj);
    for (int a = 0; a < nj; ++a) {
      LowRankJob<T>& J = jobs[todo[a]];
      ldp[a] = (J.rows + 1) / 2 * 2;
      if (hs_lr_alloc((void**)&Y[a], sizeof(T) * ((size_t)ldp[a] * J.k + 32)) != 0) {
        for (T* y : Y) hs_lr_free(y);
        hs_set_error(-7, 0, "hipMalloc of a %d x %d sketch failed", J.rows, J.k);
        return -7;
      }
      const int nblk = (J.k + HS_PB - 1) / HS_PB, ncand = ((J.rows + 127) / 128 + 1) * HS_PB;
      hgp[a] = GemmProb<T>{J.X, dOm + offOm[a], Y[a], sparse[a] ? 0 : J.rows, J.k, J.cols, J.ldx, J.cols, ldp[a]};  // (M = 0: the GEMM skips a block that is sketched sparsely)
      NodeDesc<T>& d = hn[a];
      memset(&d, 0, sizeof d);
      d.LF = Y[a];
      d.UR = J.X;
      d.SB = nullptr;
      d.invL = dInv + offInv[a];
      d.invU = d.invL + (size_t)nblk * HS_PB * HS_PB;
      d.ipiv = dInt + offInt[a];
      d.rperm = d.ipiv + J.k;
      d.cand0 = d.rperm + J.rows;
      d.cand1 = d.cand0 + ncand;
      d.pivlist = d.cand1 + ncand;
      d.info = d.pivlist + HS_PB;
      d.ni = J.k; d.nb = J.rows - J.k; d.m = J.rows;
      d.ldl = ldp[a]; d.ldu = J.ldx; d.lds = 2;
      d.ni1 = J.k; d.nb1 = J.rows - J.k; d.isleaf = 1; d.node = a;
      d.pivrows = J.rows;
      d.finalize();
      d.mrows[1] = J.k;   // "UR" = X: the TRSM touches its first k rows, the swaps reach every row
      d.mcols[1] = J.cols;
      d.mrows[2] = 0;
      d.mcols[2] = 0;
      hdj[a] = DiagJob<T>{Y[a], ldp[a], J.k, dDg + offDg[a]};
    }
    std::vector<T*> Y0(nj, nullptr);
    auto free_Y = [&]() {
      for (T* y : Y)
        hs_lr_free(y);
      for (T* y : Y0)
        hs_lr_free(y);
    };
    hipError_t e = hipMemcpyAsync(dn, hn.data(), sizeof(NodeDesc<T>) * nj, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(dgp, hgp.data(), sizeof(GemmProb<T>) * nj, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ddj, hdj.data(), sizeof(DiagJob<T>) * nj, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);  // the sources are host vectors
    if (e != hipSuccess) {
      free_Y();
      hs_set_error(-6, 0, "upload of the compression descriptors failed: %s", hipGetErrorString(e));
      return -6;
    }
    launch_gemm_probs<T>(dgp, nj, maxrows, maxk, 0, s);  // Y = X * Omega, all blocks with a Gaussian Omega
    {
      std::vector<SparseSketchJob<T>> sj;
      int smaxrows = 0, smaxk = 0;
      for (int a = 0; a < nj; ++a) {
        if (!sparse[a]) continue;
        const LowRankJob<T>& J = jobs[todo[a]];
        SparseSketchJob<T> q;
        q.X = J.X; q.Y = Y[a]; q.ldx = J.ldx; q.ldy = ldp[a]; q.rows = J.rows; q.cols = J.cols; q.k = J.k;
        q.seed = J.seed * 0x9E3779B97F4A7C15ull + 0x5bd1e995ull * (unsigned long long)(pass + 1);
        q.scale = std::sqrt((double)J.k / HS_SKETCH_ZETA);
        // pi_t(j) = (a_t j + b_t) mod cols with gcd(a_t, cols) = 1; the kernel walks the inverse permutation
        unsigned long long x = q.seed ^ 0xD1B54A32D192ED03ull;
        auto next = [&]() {
          x += 0x9E3779B97F4A7C15ull;
          unsigned long long z = x;
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
          z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
          return z ^ (z >> 31);
        };
        const long long n = J.cols;
        for (int t = 0; t < HS_SKETCH_ZETA; ++t) {
          long long at;
          do {
            at = 1 + (long long)(next() % (unsigned long long)std::max<long long>(n - 1, 1));
          } while (std::gcd(at, n) != 1);
          // modular inverse by the extended Euclidean algorithm
          long long r0 = n, r1 = at, t0 = 0, t1 = 1;
          while (r1 != 0) {
            const long long qd = r0 / r1, r2 = r0 - qd * r1, t2 = t0 - qd * t1;
            r0 = r1; r1 = r2; t0 = t1; t1 = t2;
          }
          q.ainv[t] = (unsigned int)((t0 % n + n) % n);
          q.b[t] = (unsigned int)(next() % (unsigned long long)n);
          q.step[t] = (unsigned int)(((unsigned long long)q.ainv[t] * (unsigned long long)(J.k % n)) % (unsigned long long)n);
        }
        sj.push_back(q);
        smaxrows = std::max(smaxrows, J.rows);
        smaxk = std::max(smaxk, J.k);
      }
      if (!sj.empty()) {
        SparseSketchJob<T>* dsj = nullptr;
        LRB_HIP(g.alloc(&dsj, sizeof(SparseSketchJob<T>) * sj.size()));
        hipError_t es = hipMemcpyAsync(dsj, sj.data(), sizeof(SparseSketchJob<T>) * sj.size(), hipMemcpyHostToDevice, s);
        if (es == hipSuccess) es = hipStreamSynchronize(s);  // the source is a host vector
        if (es != hipSuccess) {
          free_Y();
          hs_set_error(-6, 0, "upload of the sketch descriptors failed: %s", hipGetErrorString(es));
          return -6;
        }
        hipLaunchKernelGGL(sparse_sketch_kernel<T>, dim3((unsigned)((smaxrows + 255) / 256), (unsigned)smaxk, (unsigned)sj.size()), dim3(256), 0, s, (const SparseSketchJob<T>*)dsj);
      }
    }
    if (sketch_only) {  // no pivoted LU: the sketches themselves are the result (rows ordered by the caller)
      for (int a = 0; a < nj; ++a) {
        LowRankJob<T>& J = jobs[todo[a]];
        LowRank<T>& o = *J.out;
        if (hs_lr_alloc((void**)&o.rperm, sizeof(int) * (size_t)J.rows) != 0) {
          free_Y();
          hs_set_error(-7, 0, "hipMalloc of a row list (%d) failed", J.rows);
          return -7;
        }
        o.Y0 = Y[a];
        Y[a] = nullptr;
        o.Lp = nullptr;
        o.ldp = ldp[a];
        o.k = J.k;
        o.r = 0;
        o.top = 0.0;
      }
      LRB_HIP(hipStreamSynchronize(s));
      return 0;
    }
    if (keep_sketch) {  // the orthogonalisation that refines rank and interpolation reads the sketch itself, not its L\U
      for (int a = 0; a < nj; ++a) {
        const LowRankJob<T>& J = jobs[todo[a]];
        const size_t el = (size_t)ldp[a] * J.k + 32;
        if (hs_lr_alloc((void**)&Y0[a], sizeof(T) * el) != 0) {
          free_Y();
          hs_set_error(-7, 0, "hipMalloc of a %d x %d sketch copy failed", J.rows, J.k);
          return -7;
        }
        (void)hipMemcpyAsync(Y0[a], Y[a], sizeof(T) * (size_t)ldp[a] * J.k, hipMemcpyDeviceToDevice, s);
      }
    }
    launch_init_fronts<T>(dn, nj, maxrows, s);
    Sched<T> sch{dn, nj, maxk, maxcols, maxrows, s, &prof, nullptr, nullptr, nullptr, maxrows};
    int P2 = HS_PB;
    while (P2 < maxk) P2 *= 2;
    sch.lu_rec(0, P2);
    hipLaunchKernelGGL(absdiag_batch_kernel<T>, dim3((maxk + 63) / 64, nj), dim3(64), 0, s, (const DiagJob<T>*)ddj);
    std::vector<double> hd(nDg);
    e = hipMemcpyAsync(hd.data(), dDg, sizeof(double) * nDg, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      free_Y();
      hs_set_error(-6, 0, "pivoted LU of the sketches failed: %s", hipGetErrorString(e));
      return -6;
    }
    // ---- ranks; blocks whose rank reaches the sketch width go to the next pass with twice the width ---------------
    std::vector<int> next;
    std::vector<int> rank(nj, 0);
    bool any_done = false;
    for (int a = 0; a < nj; ++a) {
      LowRankJob<T>& J = jobs[todo[a]];
      const double* dg = hd.data() + offDg[a];
      const double tau = std::max(atol, rtol * dg[0]);
      int r = 0;
      for (int j = 0; j < J.k; ++j)
        if (dg[j] > tau) r = j + 1;
      const int kmax = std::min(J.rows, J.cols);
      if (!keep_sketch && r + 8 > J.k && J.k < kmax) {  // (with keep_sketch the caller judges the width from the refined rank)
        J.k = std::min(2 * J.k, kmax);
        next.push_back(todo[a]);
        hs_lr_free(Y[a]);
        Y[a] = nullptr;
        hs_lr_free(Y0[a]);
        Y0[a] = nullptr;
        hn[a].mcols[1] = 0;  // takes no part in the second phase of this pass
        hn[a].mrows[1] = 0;
        rank[a] = -1;
      } else {
        rank[a] = r;
        any_done = true;
      }
    }
    if (any_done) {
      // ---- Z = L11^-1 * (P*X)[:k, :] on X itself for the finished blocks, then keep the first r rows ---------------
      e = hipMemcpyAsync(dn, hn.data(), sizeof(NodeDesc<T>) * nj, hipMemcpyHostToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        free_Y();
        hs_set_error(-6, 0, "upload of the compression descriptors failed: %s", hipGetErrorString(e));
        return -6;
      }
      if (need_z) {
        sch.laswp(HS_MAT_UR, 0, HS_BIG, 0, P2);
        sch.trsm_rec(HS_MAT_UR, 0, P2, 0, HS_BIG);
      }
      for (int a = 0; a < nj; ++a) {
        if (rank[a] < 0) continue;
        LowRankJob<T>& J = jobs[todo[a]];
        LowRank<T>& o = *J.out;
        const int r = rank[a];
        o.ldz = std::max(2, (r + 1) / 2 * 2);
        if ((need_z && hs_lr_alloc((void**)&o.Z, sizeof(T) * ((size_t)o.ldz * J.cols + 32)) != 0) ||
            hs_lr_alloc((void**)&o.rperm, sizeof(int) * J.rows) != 0) {
          free_Y();
          hs_set_error(-7, 0, "hipMalloc of a low-rank factor (%d x %d) failed", r, J.cols);
          return -7;
        }
        if (r > 0 && need_z) hipLaunchKernelGGL(copy_top_rows_kernel<T>, dim3(J.cols), dim3(256), 0, s, (const T*)J.X, J.ldx, o.Z, o.ldz, r);
        (void)hipMemcpyAsync(o.rperm, hn[a].rperm, sizeof(int) * J.rows, hipMemcpyDeviceToDevice, s);
        o.Lp = Y[a];
        Y[a] = nullptr;
        o.Y0 = Y0[a];
        Y0[a] = nullptr;
        o.ldp = ldp[a];
        o.k = J.k;
        o.r = r;
        o.top = (hd.data() + offDg[a])[0];
      }
    }
    LRB_HIP(hipStreamSynchronize(s));  // the temporaries of this pass are released by `g`
    todo.swap(next);
  }
  return 0;
}

template int lowrank_compress_batch<double>(LowRankJob<double>*, int, double, double, hipStream_t, bool, bool, bool);
template int lowrank_compress_batch<cplx>(LowRankJob<cplx>*, int, double, double, hipStream_t, bool, bool, bool);
