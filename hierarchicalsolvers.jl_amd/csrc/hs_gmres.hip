// hs_gmres.hip -- right-preconditioned restarted GMRES with every vector on the device, behind the C ABI (include/hs_solver.h).
//
// Reference scenario (test/rungmres.jl:47-48):  x, ch = gmres(A, b; Pr=F, reltol=1e-9, restart=30, log=true, maxiter=30)
// IterativeSolvers.jl (0.9.0) is not part of the reference tree: the algorithm is the textbook one (Saad & Schultz 1986; Arnoldi with
// classical Gram-Schmidt + one re-orthogonalisation pass, Givens rotations on the Hessenberg matrix, right preconditioning
// x = x0 + Pr^-1 V y) and matches the Python mirror hierarchicalsolvers.jl_amd/gmres.py step for step (same iteration counts).
// PARITY UNPINNED against the Julia package.
//
// Kernels (all HBM-bound, n-vectors): CSR SpMV (one thread per row; a stencil row has <= 27 entries), the (k+1) simultaneous dot
// products V^H w of an Arnoldi step (per-workgroup partial sums, then one small reduction), the update w -= V h, norm and scale;
// the Hessenberg / Givens recurrences and the final triangular solve run in one single-thread kernel each on device-resident H, c, s,
// g -- the host reads ONE double (the residual estimate) per iteration to decide whether to go on: a preconditioner application costs
// three orders of magnitude more than that read, and an iteration that is not needed costs a whole `ldiv!`.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/hs_solver.h"
#include "hs_common.h"

#define GM_MAXK 64  // restart length limit (Krylov vectors held: restart + 1)

namespace {

template <class T>
__device__ inline T conj_(T a);
template <>
__device__ inline double conj_<double>(double a) { return a; }
template <>
__device__ inline cplx conj_<cplx>(cplx a) { return {a.re, -a.im}; }
template <class T>
__device__ inline double abs2_(T a);
template <>
__device__ inline double abs2_<double>(double a) { return a * a; }
template <>
__device__ inline double abs2_<cplx>(cplx a) { return a.re * a.re + a.im * a.im; }
template <class T>
__device__ inline T scale_(T a, double s);
template <>
__device__ inline double scale_<double>(double a, double s) { return a * s; }
template <>
__device__ inline cplx scale_<cplx>(cplx a, double s) { return {a.re * s, a.im * s}; }

// y = A x (CSR, 0-based)   mode 1: y = b - A x
template <class T>
__global__ __launch_bounds__(256) void spmv_csr_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ colind, const T* __restrict__ val,
                                                       const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ b, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T acc = Scal<T>::zero();
  for (int64_t e = rowptr[i]; e < rowptr[i + 1]; ++e) acc = Scal<T>::fma(val[e], x[colind[e]], acc);
  y[i] = b ? b[i] - acc : acc;
}

// part[blockIdx.x * (k+1) + j] = sum over the block's rows of conj(V[j][i]) * w[i],  j = 0..k
template <class T>
__global__ __launch_bounds__(256) void multi_dot_kernel(const T* __restrict__ V, int64_t ldv, int k1, const T* __restrict__ w, T* __restrict__ part, int64_t n) {
  __shared__ T sh[256];
  const int64_t i0 = (int64_t)blockIdx.x * 1024;
  T wv[4];
  for (int t = 0; t < 4; ++t) {
    const int64_t i = i0 + t * 256 + threadIdx.x;
    wv[t] = i < n ? w[i] : Scal<T>::zero();
  }
  for (int j = 0; j < k1; ++j) {
    T acc = Scal<T>::zero();
    for (int t = 0; t < 4; ++t) {
      const int64_t i = i0 + t * 256 + threadIdx.x;
      if (i < n) acc = Scal<T>::fma(conj_(V[(size_t)j * ldv + i]), wv[t], acc);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * k1 + j] = sh[0];
    __syncthreads();
  }
}
// h[j] (+)= sum over blocks of part[b * k1 + j]   (one workgroup; deterministic order)
template <class T>
__global__ __launch_bounds__(256) void reduce_parts_kernel(const T* __restrict__ part, int nblk, int k1, T* __restrict__ h, T* __restrict__ hacc) {
  __shared__ T sh[256];
  for (int j = 0; j < k1; ++j) {
    T acc = Scal<T>::zero();
    for (int b = threadIdx.x; b < nblk; b += 256) acc = acc + part[(size_t)b * k1 + j];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
      if ((int)threadIdx.x < st) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + st];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      h[j] = sh[0];
      if (hacc) hacc[j] = hacc[j] + sh[0];
    }
    __syncthreads();
  }
}
// w[i] -= sum_j h[j] * V[j][i]
template <class T>
__global__ __launch_bounds__(256) void multi_axpy_kernel(const T* __restrict__ V, int64_t ldv, int k1, const T* __restrict__ h, T* __restrict__ w, int64_t n) {
  __shared__ T sh[GM_MAXK + 1];
  if ((int)threadIdx.x < k1) sh[threadIdx.x] = h[threadIdx.x];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T acc = w[i];
  for (int j = 0; j < k1; ++j) acc = Scal<T>::fnma(sh[j], V[(size_t)j * ldv + i], acc);
  w[i] = acc;
}
// out[i] = x[i] + sum_j y[j] * V[j][i]  (x may be null: out = V^T y)
template <class T>
__global__ __launch_bounds__(256) void combine_kernel(const T* __restrict__ V, int64_t ldv, int k, const T* __restrict__ y, const T* __restrict__ x, T* __restrict__ out,
                                                      int64_t n) {
  __shared__ T sh[GM_MAXK + 1];
  if ((int)threadIdx.x < k) sh[threadIdx.x] = y[threadIdx.x];
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  T acc = x ? x[i] : Scal<T>::zero();
  for (int j = 0; j < k; ++j) acc = Scal<T>::fma(sh[j], V[(size_t)j * ldv + i], acc);
  out[i] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void norm2_part_kernel(const T* __restrict__ w, double* __restrict__ part, int64_t n) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int t = 0; t < 4; ++t) {
    const int64_t i = (int64_t)blockIdx.x * 1024 + t * 256 + threadIdx.x;
    if (i < n) acc += abs2_(w[i]);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void norm2_final_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) acc += part[b];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = sqrt(sh[0]);
}
// v[i] = w[i] / *nrm  (nothing when the norm is zero: breakdown is handled by the caller through the residual estimate)
template <class T>
__global__ __launch_bounds__(256) void scale_into_kernel(const T* __restrict__ w, const double* __restrict__ nrm, T* __restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double a = *nrm;
  v[i] = a > 0.0 ? scale_(w[i], 1.0 / a) : w[i];
}

// Device-resident small state of one restart cycle: H ((m+1) x m, column-major, ld m+1), c, s, g, the new column h (+ its norm)
template <class T>
struct GmSmall {
  T* H;
  T* cs;  // real cosines stored as T
  T* sn;
  T* g;
  T* h;         // column k of the Hessenberg matrix before the rotations: h[0..k] = (h1 + h2), h[k+1] = ||w||
  double* hn;   // ||w||
  double* res;  // |g[k+1]| after the rotation
  T* y;
  int ld;
};
__device__ inline double absT(double a) { return fabs(a); }
__device__ inline double absT(cplx a) { return sqrt(a.re * a.re + a.im * a.im); }

// applies the previous rotations to column k, creates the rotation that annihilates H[k+1, k], updates g (gmres.py lines "apply the
// previous rotations ...")
template <class T>
__global__ void givens_kernel(GmSmall<T> S, int k) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  T* Hc = S.H + (size_t)k * S.ld;
  for (int i = 0; i <= k; ++i) Hc[i] = S.h[i];
  T hk1 = Scal<T>::zero();
  *((double*)&hk1) = *S.hn;  // real part = ||w||
  Hc[k + 1] = hk1;
  for (int i = 0; i < k; ++i) {
    const T t = S.cs[i] * Hc[i] + S.sn[i] * Hc[i + 1];
    Hc[i + 1] = Scal<T>::zero() - conj_(S.sn[i]) * Hc[i] + S.cs[i] * Hc[i + 1];
    Hc[i] = t;
  }
  const T a = Hc[k], b = Hc[k + 1];
  const double aa = absT(a), den = sqrt(aa * aa + absT(b) * absT(b));
  T c = Scal<T>::one(), s = Scal<T>::zero();
  if (den != 0.0) {
    c = Scal<T>::zero();
    *((double*)&c) = aa / den;
    const T ph = aa > 0.0 ? scale_(a, 1.0 / aa) : Scal<T>::one();
    s = scale_(ph * conj_(b), 1.0 / den);
  }
  S.cs[k] = c;
  S.sn[k] = s;
  Hc[k] = c * a + s * b;
  Hc[k + 1] = Scal<T>::zero();
  S.g[k + 1] = Scal<T>::zero() - conj_(s) * S.g[k];
  S.g[k] = c * S.g[k];
  *S.res = absT(S.g[k + 1]);
}
// y = triu(H[:k, :k]) \ g[:k]
template <class T>
__global__ void hess_solve_kernel(GmSmall<T> S, int k) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = k - 1; i >= 0; --i) {
    T acc = S.g[i];
    for (int j = i + 1; j < k; ++j) acc = Scal<T>::fnma(S.H[(size_t)i + (size_t)j * S.ld], S.y[j], acc);
    S.y[i] = acc / S.H[(size_t)i + (size_t)i * S.ld];
  }
}
template <class T>
__global__ void gm_reset_kernel(GmSmall<T> S, int m, const double* beta) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int i = 0; i <= m; ++i) S.g[i] = Scal<T>::zero();
  T b = Scal<T>::zero();
  *((double*)&b) = *beta;
  S.g[0] = b;
}

struct DevBuf {
  std::vector<void*> p;
  ~DevBuf() {
    for (void* q : p)
      if (q) (void)hipFree(q);
  }
  template <class U>
  U* get(size_t count) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<size_t>(count * sizeof(U), 256)) != hipSuccess) {
      hs_set_error(HS_ERR_NOMEM, 0, "hipMalloc of %zu bytes failed (GMRES workspace)", count * sizeof(U));
      throw (int)HS_ERR_NOMEM;
    }
    p.push_back(q);
    return (U*)q;
  }
};

#define GM_HIP(call)                                                                              \
  do {                                                                                            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      hs_set_error(HS_ERR_DEVICE, 0, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      throw (int)HS_ERR_DEVICE;                                                                   \
    }                                                                                             \
  } while (0)

template <class T>
int prec_apply(hs_handle* F, T* out, const T* in, int64_t n, hipStream_t s);
template <>
int prec_apply<double>(hs_handle* F, double* out, const double* in, int64_t n, hipStream_t s) {
  return hs_ldiv_dev_d(F, out, n, in, n, n, 1, (void*)s);
}
template <>
int prec_apply<cplx>(hs_handle* F, cplx* out, const cplx* in, int64_t n, hipStream_t s) {
  return hs_ldiv_dev_z(F, (double*)out, n, (const double*)in, n, n, 1, (void*)s);
}

template <class T>
void gmres_device(hs_handle* F, int64_t n, const int64_t* rowptr, const int32_t* colind, const T* val, const T* b, T* x, int use_x0, double reltol, double abstol,
                  int restart, int64_t maxiter, double* hist, int64_t* iters_out, int* conv_out, hipStream_t s) {
  DevBuf buf;
  const int m = restart;
  const int64_t ldv = (n + 1) / 2 * 2;
  T* V = buf.get<T>((size_t)(m + 1) * ldv);
  T* w = buf.get<T>((size_t)ldv);
  T* z = buf.get<T>((size_t)ldv);
  T* r = buf.get<T>((size_t)ldv);
  const int nblk = (int)((n + 1023) / 1024);
  T* part = buf.get<T>((size_t)nblk * (m + 2));
  double* dpart = buf.get<double>((size_t)nblk);
  double* dscal = buf.get<double>(8);  // [0] beta, [1] hn, [2] res
  GmSmall<T> S;
  S.ld = m + 1;
  S.H = buf.get<T>((size_t)(m + 1) * m);
  S.cs = buf.get<T>(m + 1);
  S.sn = buf.get<T>(m + 1);
  S.g = buf.get<T>(m + 2);
  S.h = buf.get<T>(m + 2);
  S.y = buf.get<T>(m + 1);
  S.hn = dscal + 1;
  S.res = dscal + 2;
  T* h2 = buf.get<T>(m + 2);
  const unsigned gn = (unsigned)((n + 255) / 256);
  auto norm = [&](const T* v, double* dout) {
    hipLaunchKernelGGL(norm2_part_kernel<T>, dim3(nblk), dim3(256), 0, s, v, dpart, n);
    hipLaunchKernelGGL(norm2_final_kernel, dim3(1), dim3(256), 0, s, (const double*)dpart, nblk, dout);
  };
  auto read = [&](const double* d) {
    double v = 0.0;
    GM_HIP(hipMemcpyAsync(&v, d, sizeof(double), hipMemcpyDeviceToHost, s));
    GM_HIP(hipStreamSynchronize(s));
    return v;
  };
  if (!use_x0) GM_HIP(hipMemsetAsync(x, 0, sizeof(T) * (size_t)n, s));
  // r = b - A x
  if (use_x0)
    hipLaunchKernelGGL(spmv_csr_kernel<T>, dim3(gn), dim3(256), 0, s, rowptr, colind, val, (const T*)x, r, b, n);
  else
    GM_HIP(hipMemcpyAsync(r, b, sizeof(T) * (size_t)n, hipMemcpyDeviceToDevice, s));
  norm(r, dscal);
  double beta = read(dscal);
  const double tol = std::max(reltol * beta, abstol);
  int64_t it = 0;
  size_t nh = 0;
  hist[nh++] = beta;
  bool converged = beta <= tol;
  while (!converged && it < maxiter) {
    hipLaunchKernelGGL(scale_into_kernel<T>, dim3(gn), dim3(256), 0, s, (const T*)r, (const double*)dscal, V, n);
    hipLaunchKernelGGL(gm_reset_kernel<T>, dim3(1), dim3(1), 0, s, S, m, (const double*)dscal);
    int k_used = 0;
    for (int k = 0; k < m && it < maxiter; ++k) {
      const T* vk = V + (size_t)k * ldv;
      const T* zz = vk;
      if (F) {
        const int st = prec_apply<T>(F, z, vk, n, s);
        if (st != 0) throw st;
        zz = z;
      }
      hipLaunchKernelGGL(spmv_csr_kernel<T>, dim3(gn), dim3(256), 0, s, rowptr, colind, val, zz, w, (const T*)nullptr, n);
      // classical Gram-Schmidt with one re-orthogonalisation pass: h = V^H w; w -= V h; h2 = V^H w; w -= V h2; H[:, k] = h + h2
      const int k1 = k + 1;
      hipLaunchKernelGGL(multi_dot_kernel<T>, dim3(nblk), dim3(256), 0, s, (const T*)V, ldv, k1, (const T*)w, part, n);
      hipLaunchKernelGGL(reduce_parts_kernel<T>, dim3(1), dim3(256), 0, s, (const T*)part, nblk, k1, S.h, (T*)nullptr);
      hipLaunchKernelGGL(multi_axpy_kernel<T>, dim3(gn), dim3(256), 0, s, (const T*)V, ldv, k1, (const T*)S.h, w, n);
      hipLaunchKernelGGL(multi_dot_kernel<T>, dim3(nblk), dim3(256), 0, s, (const T*)V, ldv, k1, (const T*)w, part, n);
      hipLaunchKernelGGL(reduce_parts_kernel<T>, dim3(1), dim3(256), 0, s, (const T*)part, nblk, k1, h2, S.h);
      hipLaunchKernelGGL(multi_axpy_kernel<T>, dim3(gn), dim3(256), 0, s, (const T*)V, ldv, k1, (const T*)h2, w, n);
      norm(w, S.hn);
      hipLaunchKernelGGL(scale_into_kernel<T>, dim3(gn), dim3(256), 0, s, (const T*)w, (const double*)S.hn, V + (size_t)(k + 1) * ldv, n);
      hipLaunchKernelGGL(givens_kernel<T>, dim3(1), dim3(1), 0, s, S, k);
      double two[2];  // hn, res (adjacent)
      GM_HIP(hipMemcpyAsync(two, S.hn, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
      GM_HIP(hipStreamSynchronize(s));
      ++it;
      k_used = k + 1;
      hist[nh++] = two[1];
      if (two[1] <= tol || two[0] == 0.0) {
        converged = two[1] <= tol;
        break;
      }
    }
    if (k_used > 0) {  // x += Pr^-1 (V_k y),  H y = g
      hipLaunchKernelGGL(hess_solve_kernel<T>, dim3(1), dim3(1), 0, s, S, k_used);
      if (F) {
        hipLaunchKernelGGL(combine_kernel<T>, dim3(gn), dim3(256), 0, s, (const T*)V, ldv, k_used, (const T*)S.y, (const T*)nullptr, w, n);
        const int st = prec_apply<T>(F, z, w, n, s);
        if (st != 0) throw st;
        T one = Scal<T>::one();
        GM_HIP(hipMemcpyAsync(S.y, &one, sizeof(T), hipMemcpyHostToDevice, s));
        GM_HIP(hipStreamSynchronize(s));
        hipLaunchKernelGGL(combine_kernel<T>, dim3(gn), dim3(256), 0, s, (const T*)z, ldv, 1, (const T*)S.y, (const T*)x, x, n);
      } else {
        hipLaunchKernelGGL(combine_kernel<T>, dim3(gn), dim3(256), 0, s, (const T*)V, ldv, k_used, (const T*)S.y, (const T*)x, x, n);
      }
    }
    hipLaunchKernelGGL(spmv_csr_kernel<T>, dim3(gn), dim3(256), 0, s, rowptr, colind, val, (const T*)x, r, b, n);
    norm(r, dscal);
    beta = read(dscal);
    converged = converged || beta <= tol;
    if (k_used == 0) break;
  }
  GM_HIP(hipStreamSynchronize(s));
  *iters_out = it;
  *conv_out = converged ? 1 : 0;
}

// CSR (0-based, device) of a host CSC matrix given with 1-based Julia fields
template <class T>
void upload_csr(DevBuf& buf, int64_t n, const int64_t* colptr, const int64_t* rowval, const T* nz, int64_t** d_rp, int32_t** d_ci, T** d_v) {
  if (colptr[0] != 1) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: colptr must be 1-based (SparseMatrixCSC)");
    throw (int)HS_ERR_ARGUMENT;
  }
  const int64_t nnz = colptr[n] - 1;
  std::vector<int64_t> rp((size_t)n + 1, 0);
  std::vector<int32_t> ci((size_t)nnz);
  std::vector<T> v((size_t)nnz);
  for (int64_t e = 0; e < nnz; ++e) {
    if (rowval[e] < 1 || rowval[e] > n) {
      hs_set_error(HS_ERR_DIMENSION, e, "BoundsError: rowval[%lld] = %lld outside 1:%lld", (long long)e + 1, (long long)rowval[e], (long long)n);
      throw (int)HS_ERR_DIMENSION;
    }
    rp[(size_t)rowval[e]]++;
  }
  for (int64_t r = 0; r < n; ++r) rp[(size_t)r + 1] += rp[(size_t)r];
  std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
  for (int64_t c = 0; c < n; ++c)
    for (int64_t e = colptr[c] - 1; e < colptr[c + 1] - 1; ++e) {
      const int64_t at = fill[(size_t)(rowval[e] - 1)]++;
      ci[(size_t)at] = (int32_t)c;
      v[(size_t)at] = nz[e];
    }
  *d_rp = buf.get<int64_t>((size_t)n + 1);
  *d_ci = buf.get<int32_t>((size_t)nnz);
  *d_v = buf.get<T>((size_t)nnz);
  GM_HIP(hipMemcpy(*d_rp, rp.data(), sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(*d_ci, ci.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
  GM_HIP(hipMemcpy(*d_v, v.data(), sizeof(T) * (size_t)nnz, hipMemcpyHostToDevice));
}

template <class T>
int gmres_entry(hs_handle* F, int64_t n, const int64_t* colptr, const int64_t* rowval, const T* nz, const T* b, T* x, int where, int use_x0, double reltol, double abstol,
                int64_t restart, int64_t maxiter, double* resnorm, int64_t* iters, int* converged, void* stream) {
  if (n <= 0 || !colptr || !rowval || !nz || !b || !x || !iters || !converged) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_gmres needs A (CSC), b, x and the two result slots");
    return HS_ERR_ARGUMENT;
  }
  if (F && (hs_size(F) != n || (hs_is_complex(F) != 0) != (sizeof(T) == 16))) {
    hs_set_error(HS_ERR_DIMENSION, 0, "DimensionMismatch: the preconditioner is %lld x %lld %s, A is %lld x %lld", (long long)hs_size(F), (long long)hs_size(F),
                 hs_is_complex(F) ? "ComplexF64" : "Float64", (long long)n, (long long)n);
    return HS_ERR_DIMENSION;
  }
  // defaults of IterativeSolvers 0.9: restart = min(20, n), maxiter = n, reltol = sqrt(eps)
  if (restart <= 0) restart = std::min<int64_t>(20, n);
  if (restart > GM_MAXK) {
    hs_set_error(HS_ERR_ARGUMENT, restart, "ArgumentError: restart = %lld exceeds the limit of %d", (long long)restart, GM_MAXK);
    return HS_ERR_ARGUMENT;
  }
  if (maxiter < 0) maxiter = n;
  if (!(reltol >= 0.0)) reltol = 1.4901161193847656e-08;
  try {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
      hs_set_error(HS_ERR_DEVICE, 0, "no HIP device available: this library has no CPU fallback");
      return HS_ERR_DEVICE;
    }
    DevBuf buf;
    hipStream_t s = (hipStream_t)stream;
    int64_t* d_rp;
    int32_t* d_ci;
    T* d_v;
    upload_csr<T>(buf, n, colptr, rowval, nz, &d_rp, &d_ci, &d_v);
    std::vector<double> hist((size_t)maxiter + 2, 0.0);
    const T* db = b;
    T* dx = x;
    if (where == 0) {
      T* tb = buf.get<T>((size_t)n);
      T* tx = buf.get<T>((size_t)n);
      GM_HIP(hipMemcpy(tb, b, sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
      if (use_x0) GM_HIP(hipMemcpy(tx, x, sizeof(T) * (size_t)n, hipMemcpyHostToDevice));
      db = tb;
      dx = tx;
    }
    gmres_device<T>(F, n, d_rp, d_ci, d_v, db, dx, use_x0, reltol, abstol, (int)restart, maxiter, hist.data(), iters, converged, s);
    if (where == 0) GM_HIP(hipMemcpy(x, dx, sizeof(T) * (size_t)n, hipMemcpyDeviceToHost));
    if (resnorm)
      for (int64_t i = 0; i <= *iters; ++i) resnorm[i] = hist[(size_t)i];
    return HS_OK;
  } catch (int code) {
    return code;
  } catch (const std::bad_alloc&) {
    hs_set_error(HS_ERR_NOMEM, 0, "host allocation failed");
    return HS_ERR_NOMEM;
  }
}

}  // namespace

extern "C" int hs_gmres_d(hs_handle* Pr, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, double* x, int where, int use_x0,
                          double reltol, double abstol, int64_t restart, int64_t maxiter, double* resnorm, int64_t* iters, int* converged, void* stream) {
  return gmres_entry<double>(Pr, n, colptr, rowval, nzval, b, x, where, use_x0, reltol, abstol, restart, maxiter, resnorm, iters, converged, stream);
}
extern "C" int hs_gmres_z(hs_handle* Pr, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, double* x, int where, int use_x0,
                          double reltol, double abstol, int64_t restart, int64_t maxiter, double* resnorm, int64_t* iters, int* converged, void* stream) {
  return gmres_entry<cplx>(Pr, n, colptr, rowval, (const cplx*)nzval, (const cplx*)b, (cplx*)x, where, use_x0, reltol, abstol, restart, maxiter, resnorm, iters, converged,
                           stream);
}
