// hs_lrdense.hip -- dense-factor helpers of the compressed fronts (hs_compress.h).
//
//   lr_expand      C (rows x r, dense) = P' * unit-lower-trapezoid(Lp[:, :r])   -- the row-ID factor of hs_lowrank.hip
//   lr_dense       dst[didx ? didx[i] : i] -= sum_j C[i, j] * t[j]              -- apply a dense low-rank factor in ldiv!
//   rtrsm_upper    Xout = X * U^-1 for the upper factor of a front's interior block (right-looking, MFMA GEMMs on
//                  plain problems: the update is X[:, mid:c1) -= Xout[:, c0:mid) * U[c0:mid, mid:c1), the base case
//                  multiplies by the stored inverse of the 32x32 diagonal block)
//
// Reference roles: `L.V = (L.V' * Aii^-1)'` of `_lgauss_transform` (src/factorization.jl:174) is rtrsm_upper
// (the L^-1*P half is applied to the other factor at solve time, see hs_compress.h); `_lsolve!` / `_rsolve!`
// with LowRankMatrix operands (src/factornode.jl:77-88) are lr_zmul + lr_dense.
#include <vector>

#include "hs_common.h"
#include "hs_lowrank.h"

template <class T>
__global__ __launch_bounds__(256) void lr_expand_kernel(const T* __restrict__ Lp, int ldp, int rows, int r, const int* __restrict__ rp, T* __restrict__ C, int ldc) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const int orow = rp[i];
  for (int j = 0; j < r; ++j) {
    T v = Scal<T>::zero();
    if (i == j)
      v = Scal<T>::one();
    else if (i > j)
      v = Lp[(size_t)i + (size_t)j * ldp];
    C[(size_t)orow + (size_t)j * ldc] = v;
  }
}

template <class T>
__global__ __launch_bounds__(256) void lr_dense_kernel(const T* __restrict__ C, int ldc, int rows, int r, const T* __restrict__ t, T* __restrict__ dst, const int* __restrict__ didx) {
  __shared__ T s_t[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  T acc = Scal<T>::zero();
  for (int j0 = 0; j0 < r; j0 += 256) {
    __syncthreads();
    if (j0 + (int)threadIdx.x < r) s_t[threadIdx.x] = t[j0 + threadIdx.x];
    __syncthreads();
    const int jn = min(256, r - j0);
    if (i < rows)
      for (int j = 0; j < jn; ++j) acc = Scal<T>::fma(C[(size_t)i + (size_t)(j0 + j) * ldc], s_t[j], acc);
  }
  if (i < rows) {
    const int d = didx ? didx[i] : i;
    dst[d] = dst[d] - acc;
  }
}

template <class T>
void lowrank_expand(LowRank<T>& lr, hipStream_t s) {
  if (lr.Cd || lr.r <= 0 || lr.rows <= 0) return;
  lr.ldc = (lr.rows + 1) / 2 * 2;
  if (hipMalloc((void**)&lr.Cd, sizeof(T) * ((size_t)lr.ldc * lr.r + 32)) != hipSuccess) {
    lr.Cd = nullptr;
    return;
  }
  hipLaunchKernelGGL(lr_expand_kernel<T>, dim3((lr.rows + 255) / 256), dim3(256), 0, s, (const T*)lr.Lp, lr.ldp, lr.rows, lr.r, (const int*)lr.rperm, lr.Cd, lr.ldc);
}

template <class T>
void launch_lr_dense(const T* C, int ldc, int rows, int r, const T* t, T* dst, const int* didx, hipStream_t s) {
  if (rows <= 0 || r <= 0) return;
  hipLaunchKernelGGL(lr_dense_kernel<T>, dim3((rows + 255) / 256), dim3(256), 0, s, C, ldc, rows, r, t, dst, didx);
}

// Xout (r x n) = X (r x n) * U^-1, U = upper triangle of LF[0:n, 0:n] (ld ldl), invU = its inverted 32x32 diagonal
// blocks.  X is overwritten (workspace).  Asynchronous on `s`; *dprobs_out is a device array the caller frees after
// synchronising.  Returns 0, or a HIP error code.
template <class T>
int rtrsm_upper(T* X, int ldx, T* Xout, int ldo, const T* LF, int ldl, const T* invU, int n, int r, hipStream_t s, void** dprobs_out) {
  *dprobs_out = nullptr;
  if (n <= 0 || r <= 0) return 0;
  std::vector<GemmProb<T>> probs;
  std::vector<char> minus;
  int P2 = HS_PB;
  while (P2 < n) P2 *= 2;
  struct Rec {
    static void run(int c0, int c1, int n, int r, T* X, int ldx, T* Xout, int ldo, const T* LF, int ldl, const T* invU, std::vector<GemmProb<T>>& probs,
                    std::vector<char>& minus) {
      if (c0 >= n) return;
      if (c1 - c0 == HS_PB) {
        const int w = std::min(HS_PB, n - c0);
        probs.push_back({X + (size_t)c0 * ldx, invU + (size_t)(c0 / HS_PB) * HS_PB * HS_PB, Xout + (size_t)c0 * ldo, r, w, w, ldx, HS_PB, ldo});
        minus.push_back(0);
        return;
      }
      const int mid = (c0 + c1) / 2;
      run(c0, mid, n, r, X, ldx, Xout, ldo, LF, ldl, invU, probs, minus);
      if (mid < n) {
        const int ce = std::min(c1, n);
        probs.push_back({Xout + (size_t)c0 * ldo, LF + (size_t)c0 + (size_t)mid * ldl, X + (size_t)mid * ldx, r, ce - mid, mid - c0, ldo, ldl, ldx});
        minus.push_back(1);
        run(mid, c1, n, r, X, ldx, Xout, ldo, LF, ldl, invU, probs, minus);
      }
    }
  };
  Rec::run(0, P2, n, r, X, ldx, Xout, ldo, LF, ldl, invU, probs, minus);
  GemmProb<T>* dp = nullptr;
  hipError_t e = hipMalloc((void**)&dp, sizeof(GemmProb<T>) * probs.size());
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync(dp, probs.data(), sizeof(GemmProb<T>) * probs.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // probs is a host vector
  if (e != hipSuccess) {
    (void)hipFree(dp);
    return (int)e;
  }
  for (size_t i = 0; i < probs.size(); ++i) launch_gemm_probs<T>(dp + i, 1, probs[i].M, probs[i].N, minus[i], s);
  *dprobs_out = dp;  // the launches are in flight: the caller frees it after synchronising the stream
  return 0;
}

// The same solve for several fronts at once: every step of the recursion is ONE grouped launch over the fronts
// (a front that has no columns in a step contributes an empty problem).
template <class T>
int rtrsm_upper_batch(const RtrsmJob<T>* jobs, int nj, hipStream_t s, void** dprobs_out) {
  *dprobs_out = nullptr;
  int maxn = 0, maxr = 0;
  for (int a = 0; a < nj; ++a) {
    maxn = std::max(maxn, jobs[a].r > 0 ? jobs[a].n : 0);
    maxr = std::max(maxr, jobs[a].r);
  }
  if (maxn <= 0 || maxr <= 0) return 0;
  int P2 = HS_PB;
  while (P2 < maxn) P2 *= 2;
  std::vector<GemmProb<T>> probs;
  std::vector<char> minus;
  std::vector<int> stepN;
  const GemmProb<T> empty{nullptr, nullptr, nullptr, 0, 0, 0, 2, 2, 2};
  struct Rec {
    static void run(int c0, int c1, const RtrsmJob<T>* jobs, int nj, int maxn, std::vector<GemmProb<T>>& probs, std::vector<char>& minus, std::vector<int>& stepN,
                    const GemmProb<T>& empty) {
      if (c0 >= maxn) return;
      bool wide = (c1 - c0 == 256);
      for (int a = 0; a < nj && wide; ++a) wide = jobs[a].inv256U != nullptr;
      if (wide) {  // 256-column base case: one product with the stored inverse of the 256 x 256 diagonal block
        int mx = 0;
        for (int a = 0; a < nj; ++a) {
          const RtrsmJob<T>& J = jobs[a];
          const int w = std::min(256, J.n - c0);
          if (J.r <= 0 || w <= 0) {
            probs.push_back(empty);
            continue;
          }
          probs.push_back({J.X + (size_t)c0 * J.ldx, J.inv256U + (size_t)(c0 / 256) * 65536, J.Xout + (size_t)c0 * J.ldo, J.r, w, w, J.ldx, 256, J.ldo});
          mx = std::max(mx, w);
        }
        minus.push_back(0);
        stepN.push_back(mx);
        return;
      }
      if (c1 - c0 == HS_PB) {
        int mx = 0;
        for (int a = 0; a < nj; ++a) {
          const RtrsmJob<T>& J = jobs[a];
          const int w = std::min(HS_PB, J.n - c0);
          if (J.r <= 0 || w <= 0) {
            probs.push_back(empty);
            continue;
          }
          probs.push_back({J.X + (size_t)c0 * J.ldx, J.invU + (size_t)(c0 / HS_PB) * HS_PB * HS_PB, J.Xout + (size_t)c0 * J.ldo, J.r, w, w, J.ldx, HS_PB, J.ldo});
          mx = std::max(mx, w);
        }
        minus.push_back(0);
        stepN.push_back(mx);
        return;
      }
      const int mid = (c0 + c1) / 2;
      run(c0, mid, jobs, nj, maxn, probs, minus, stepN, empty);
      if (mid < maxn) {
        int mx = 0;
        for (int a = 0; a < nj; ++a) {
          const RtrsmJob<T>& J = jobs[a];
          const int ce = std::min(c1, J.n);
          if (J.r <= 0 || ce <= mid) {
            probs.push_back(empty);
            continue;
          }
          probs.push_back({J.Xout + (size_t)c0 * J.ldo, J.LF + (size_t)c0 + (size_t)mid * J.ldl, J.X + (size_t)mid * J.ldx, J.r, ce - mid, mid - c0, J.ldo, J.ldl, J.ldx});
          mx = std::max(mx, ce - mid);
        }
        minus.push_back(1);
        stepN.push_back(mx);
        run(mid, c1, jobs, nj, maxn, probs, minus, stepN, empty);
      }
    }
  };
  Rec::run(0, P2, jobs, nj, maxn, probs, minus, stepN, empty);
  GemmProb<T>* dp = nullptr;
  hipError_t e = hipMalloc((void**)&dp, sizeof(GemmProb<T>) * probs.size());
  if (e != hipSuccess) return (int)e;
  e = hipMemcpyAsync(dp, probs.data(), sizeof(GemmProb<T>) * probs.size(), hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);  // probs is a host vector
  if (e != hipSuccess) {
    (void)hipFree(dp);
    return (int)e;
  }
  for (size_t t = 0; t < minus.size(); ++t)
    if (stepN[t] > 0) launch_gemm_probs<T>(dp + t * nj, nj, maxr, stepN[t], minus[t], s);
  *dprobs_out = dp;  // launches in flight: the caller frees it after synchronising the stream
  return 0;
}
template int rtrsm_upper_batch<double>(const RtrsmJob<double>*, int, hipStream_t, void**);
template int rtrsm_upper_batch<cplx>(const RtrsmJob<cplx>*, int, hipStream_t, void**);

template void lowrank_expand<double>(LowRank<double>&, hipStream_t);
template void lowrank_expand<cplx>(LowRank<cplx>&, hipStream_t);
template void launch_lr_dense<double>(const double*, int, int, int, const double*, double*, const int*, hipStream_t);
template void launch_lr_dense<cplx>(const cplx*, int, int, int, const cplx*, cplx*, const int*, hipStream_t);
template int rtrsm_upper<double>(double*, int, double*, int, const double*, int, const double*, int, int, hipStream_t, void**);
template int rtrsm_upper<cplx>(cplx*, int, cplx*, int, const cplx*, int, const cplx*, int, int, hipStream_t, void**);
