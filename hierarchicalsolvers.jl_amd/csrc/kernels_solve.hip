// kernels_solve.hip -- ldiv!(F, B): level-batched triangular sweeps (HBM-bound at nrhs = 1).
//
// Reference (src/factornode.jl:62-99):
//   _lsolve!  post-order  rhs[bnd] -= L * rhs[int]          L = Abi * D^-1
//   _dsolve!              rhs[int]  = D \ rhs[int]          (fresh LU per call in the reference)
//   _rsolve!  pre-order   rhs[int] -= R * rhs[bnd]          R = D^-1 * Aib
// With the stored factors P*D = L11*U11, Lbi = Abi*U11^-1, Uib = L11^-1*P*Aib the same three sweeps
// are
//   forward  (leaves -> root):  y = L11^-1 * P * rhs[int];   rhs[bnd] -= Lbi * y
//   backward (root -> leaves):  rhs[int] = U11^-1 * (y - Uib * rhs[bnd])
// i.e. every factor entry is read exactly once per sweep and nothing is re-factorised.
//
// The triangular solves advance 32 columns per launch: every workgroup recomputes the solved
// 32-vector from the stored inverse diagonal block (32x32 matvec in LDS), then updates its own 256
// rows with the 32-column panel below (forward; rows >= ni are the Lbi rows and update rhs[bnd]
// through the front's index list) or above (backward).  One right-hand side per launch.
#include "hs_common.h"

template <class T>
__global__ __launch_bounds__(256) void fwd_gather_kernel(const SolveNode<T>* __restrict__ nodes, const T* __restrict__ b, T* __restrict__ w) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nd.ni) return;
  w[nd.woff + i] = b[gld(nd.fidx + gld(nd.rperm + i))];
}

// forward step on column block `blk`: y_blk = invL * w_blk ; rows below -= L[:, blk] * y_blk
template <class T>
__global__ __launch_bounds__(256) void fwd_step_kernel(const SolveNode<T>* __restrict__ nodes, int blk, T* __restrict__ w, T* __restrict__ y,
                                                       T* __restrict__ b) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int c0 = blk * HS_PB;
  if (c0 >= nd.ni) return;
  const int wl = min(HS_PB, nd.ni - c0);
  const int r0 = c0 + wl;
  if (blockIdx.x > 0 && (int)blockIdx.x * 256 >= nd.mrows - r0) return;
  __shared__ T s_raw[HS_PB];
  __shared__ T s_y[HS_PB];
  const int t = threadIdx.x;
  if (t < HS_PB) s_raw[t] = (t < wl) ? w[nd.woff + c0 + t] : Scal<T>::zero();
  __syncthreads();
  if (t < HS_PB) {
    const T* il = nd.invL + (size_t)blk * HS_PB * HS_PB;
    T s = Scal<T>::zero();
    for (int j = 0; j <= t; ++j) s = Scal<T>::fma(gld(il + t + j * HS_PB), s_raw[j], s);
    s_y[t] = s;
    if (blockIdx.x == 0 && t < wl) y[nd.woff + c0 + t] = s;
  }
  __syncthreads();
  const int r = r0 + blockIdx.x * 256 + t;
  if (r >= nd.mrows) return;
  const T* a = nd.LF + (size_t)r + (size_t)c0 * nd.ldl;
  T acc = Scal<T>::zero();
#pragma unroll 32
  for (int j = 0; j < wl; ++j) acc = Scal<T>::fma(gld(a + (size_t)j * nd.ldl), s_y[j], acc);
  if (r < nd.ni) {
    w[nd.woff + r] = w[nd.woff + r] - acc;
  } else {
    int g = gld(nd.fidx + r);
    b[g] = b[g] - acc;
  }
}

// w1[i] = y[i] - sum_j UR[i, j] * b[bnd_j]   -- column-split partial sums, deterministic two-pass reduce
#define UPD_CS 512
template <class T>
__global__ __launch_bounds__(256) void int_update_partial_kernel(const SolveNode<T>* __restrict__ nodes, const T* __restrict__ b,
                                                                 T* __restrict__ part) {
  const SolveNode<T> nd = nodes[blockIdx.z];
  const int j0 = blockIdx.y * UPD_CS;
  if (j0 >= nd.nb || nd.compressed) return;
  if ((int)blockIdx.x * 256 >= nd.ni) return;
  const int j1 = min(nd.nb, j0 + UPD_CS);
  __shared__ T s_b[UPD_CS];
  for (int j = threadIdx.x; j < j1 - j0; j += 256) s_b[j] = b[gld(nd.fidx + nd.ni + j0 + j)];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nd.ni) return;
  const T* u = nd.UR + (size_t)i + (size_t)j0 * nd.ldu;
  T acc = Scal<T>::zero();
#pragma unroll 32
  for (int j = 0; j < j1 - j0; ++j) acc = Scal<T>::fma(gld(u + (size_t)j * nd.ldu), s_b[j], acc);
  part[nd.poff + (long long)blockIdx.y * nd.ni + i] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void int_update_reduce_kernel(const SolveNode<T>* __restrict__ nodes, const T* __restrict__ part,
                                                                const T* __restrict__ y, T* __restrict__ w) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nd.ni) return;
  const int ns = nd.compressed ? 0 : (nd.nb + UPD_CS - 1) / UPD_CS;  // compressed: R is applied by the lr kernels
  T acc = y[nd.woff + i];
  for (int s = 0; s < ns; ++s) acc = acc - part[nd.poff + (long long)s * nd.ni + i];
  w[nd.woff + i] = acc;
}

// backward step on column block `blk`: x_blk = invU * w_blk ; rows above -= U[:, blk] * x_blk
template <class T>
__global__ __launch_bounds__(256) void bwd_step_kernel(const SolveNode<T>* __restrict__ nodes, int blk, T* __restrict__ w, T* __restrict__ x) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int c0 = blk * HS_PB;
  if (c0 >= nd.ni) return;
  const int wl = min(HS_PB, nd.ni - c0);
  if (blockIdx.x > 0 && (int)blockIdx.x * 256 >= c0) return;
  __shared__ T s_raw[HS_PB];
  __shared__ T s_x[HS_PB];
  const int t = threadIdx.x;
  if (t < HS_PB) s_raw[t] = (t < wl) ? w[nd.woff + c0 + t] : Scal<T>::zero();
  __syncthreads();
  if (t < HS_PB) {
    const T* iu = nd.invU + (size_t)blk * HS_PB * HS_PB;
    T s = Scal<T>::zero();
    for (int j = t; j < HS_PB; ++j) s = Scal<T>::fma(gld(iu + t + j * HS_PB), s_raw[j], s);
    s_x[t] = s;
    if (blockIdx.x == 0 && t < wl) x[nd.woff + c0 + t] = s;
  }
  __syncthreads();
  const int r = blockIdx.x * 256 + t;
  if (r >= c0) return;
  const T* a = nd.LF + (size_t)r + (size_t)c0 * nd.ldl;
  T acc = Scal<T>::zero();
#pragma unroll 32
  for (int j = 0; j < wl; ++j) acc = Scal<T>::fma(gld(a + (size_t)j * nd.ldl), s_x[j], acc);
  w[nd.woff + r] = w[nd.woff + r] - acc;
}

template <class T>
__global__ __launch_bounds__(256) void bwd_scatter_kernel(const SolveNode<T>* __restrict__ nodes, T* __restrict__ b, const T* __restrict__ x) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= nd.ni) return;
  b[gld(nd.fidx + i)] = x[nd.woff + i];
}

// ---- low-rank Gauss transforms (compressed fronts) ---------------------------------------------------
#define LR_CS 512
template <class T>
__global__ __launch_bounds__(256) void lr_zmul_partial_kernel(const T* __restrict__ Z, int ldz, int r, int cols, const T* __restrict__ x,
                                                              const int* __restrict__ xidx, T* __restrict__ part) {
  const int j0 = blockIdx.y * LR_CS, j1 = min(cols, j0 + LR_CS);
  __shared__ T s_x[LR_CS];
  for (int j = threadIdx.x; j < j1 - j0; j += 256) s_x[j] = xidx ? x[xidx[j0 + j]] : x[j0 + j];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= r) return;
  const T* z = Z + (size_t)i + (size_t)j0 * ldz;
  T acc = Scal<T>::zero();
#pragma unroll 8
  for (int j = 0; j < j1 - j0; ++j) acc = Scal<T>::fma(z[(size_t)j * ldz], s_x[j], acc);
  part[(size_t)blockIdx.y * r + i] = acc;
}
template <class T>
__global__ __launch_bounds__(256) void lr_zmul_reduce_kernel(const T* __restrict__ part, int r, int ns, T* __restrict__ t) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= r) return;
  T acc = Scal<T>::zero();
  for (int s = 0; s < ns; ++s) acc = acc + part[(size_t)s * r + i];
  t[i] = acc;
}
// u = unit-lower-trapezoid(Lp[:, :r]) * t, one thread per row; dst[(didx ? didx[rp[i]] : rp[i])] -= u[i]
template <class T>
__global__ __launch_bounds__(256) void lr_trap_kernel(const T* __restrict__ Lp, int ldp, int rows, int r, const int* __restrict__ rp,
                                                      const T* __restrict__ t, T* __restrict__ dst, const int* __restrict__ didx) {
  __shared__ T s_t[256];
  const int i = blockIdx.x * 256 + threadIdx.x;
  T acc = Scal<T>::zero();
  for (int j0 = 0; j0 < r; j0 += 256) {
    __syncthreads();
    if (j0 + (int)threadIdx.x < r) s_t[threadIdx.x] = t[j0 + threadIdx.x];
    __syncthreads();
    if (i < rows) {
      const int jn = min(256, r - j0);
      const T* a = Lp + (size_t)i + (size_t)j0 * ldp;
      for (int j = 0; j < jn; ++j) {
        const int col = j0 + j;
        if (col < i)
          acc = Scal<T>::fma(a[(size_t)j * ldp], s_t[j], acc);
        else if (col == i)
          acc = acc + s_t[j];  // unit diagonal
      }
    }
  }
  if (i < rows) {
    const int o = rp[i];
    const int g = didx ? didx[o] : o;
    dst[g] = dst[g] - acc;
  }
}
template <class T>
void launch_lr_zmul(const T* Z, int ldz, int r, int cols, const T* x, const int* xidx, T* part, T* t, hipStream_t s) {
  if (r <= 0) return;
  const int ns = (cols + LR_CS - 1) / LR_CS;
  hipLaunchKernelGGL(lr_zmul_partial_kernel<T>, dim3((r + 255) / 256, ns), dim3(256), 0, s, Z, ldz, r, cols, x, xidx, part);
  hipLaunchKernelGGL(lr_zmul_reduce_kernel<T>, dim3((r + 255) / 256), dim3(256), 0, s, (const T*)part, r, ns, t);
}
template <class T>
void launch_lr_trap(const T* Lp, int ldp, int rows, int r, const int* rp, const T* t, T* dst, const int* didx, hipStream_t s) {
  if (r <= 0 || rows <= 0) return;
  hipLaunchKernelGGL(lr_trap_kernel<T>, dim3((rows + 255) / 256), dim3(256), 0, s, Lp, ldp, rows, r, rp, t, dst, didx);
}
template void launch_lr_zmul<double>(const double*, int, int, int, const double*, const int*, double*, double*, hipStream_t);
template void launch_lr_zmul<cplx>(const cplx*, int, int, int, const cplx*, const int*, cplx*, cplx*, hipStream_t);
template void launch_lr_trap<double>(const double*, int, int, int, const int*, const double*, double*, const int*, hipStream_t);
template void launch_lr_trap<cplx>(const cplx*, int, int, int, const int*, const cplx*, cplx*, const int*, hipStream_t);

// ---- multi-rank helpers: boundary segments cross ranks as contiguous vectors ------------------------
__global__ __launch_bounds__(256) void pack_idx_kernel(const int* __restrict__ idx, int cnt, const char* __restrict__ b, char* __restrict__ buf, int esz) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const double* s = reinterpret_cast<const double*>(b + (size_t)idx[i] * esz);
  double* d = reinterpret_cast<double*>(buf + (size_t)i * esz);
  for (int k = 0; k < esz / 8; ++k) d[k] = s[k];
}
__global__ __launch_bounds__(256) void unpack_idx_kernel(const int* __restrict__ idx, int cnt, char* __restrict__ b, const char* __restrict__ buf, int esz) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  double* d = reinterpret_cast<double*>(b + (size_t)idx[i] * esz);
  const double* s = reinterpret_cast<const double*>(buf + (size_t)i * esz);
  for (int k = 0; k < esz / 8; ++k) d[k] = s[k];
}
// out[idx[i]] = b[idx[i]]: the entries of the solution one rank owns, in one launch over a precomputed index list
__global__ __launch_bounds__(256) void copy_idx_kernel(const int* __restrict__ idx, int cnt, const char* __restrict__ b, char* __restrict__ out, int esz) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const size_t o = (size_t)idx[i] * esz;
  const double* s = reinterpret_cast<const double*>(b + o);
  double* d = reinterpret_cast<double*>(out + o);
  for (int k = 0; k < esz / 8; ++k) d[k] = s[k];
}
void launch_copy_idx(const int* idx, int cnt, const void* b, void* out, int esz, hipStream_t s) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(copy_idx_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, idx, cnt, (const char*)b, (char*)out, esz);
}
void launch_pack_idx(const int* idx, int cnt, const void* b, void* buf, int esz, hipStream_t s) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(pack_idx_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, idx, cnt, (const char*)b, (char*)buf, esz);
}
void launch_unpack_idx(const int* idx, int cnt, void* b, const void* buf, int esz, hipStream_t s) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(unpack_idx_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, idx, cnt, (char*)b, (const char*)buf, esz);
}

// ---- launchers ---------------------------------------------------------------------------------
template <class T>
void launch_fwd_gather(const SolveNode<T>* dn, int nbatch, int maxni, const T* b, T* w, hipStream_t s) {
  if (nbatch <= 0 || maxni <= 0) return;
  hipLaunchKernelGGL(fwd_gather_kernel<T>, dim3((maxni + 255) / 256, nbatch), dim3(256), 0, s, dn, b, w);
}
template <class T>
void launch_fwd_step(const SolveNode<T>* dn, int nbatch, int blk, int maxm, T* w, T* y, T* b, hipStream_t s) {
  if (nbatch <= 0) return;
  int rows = maxm - blk * HS_PB;
  int gx = rows > 0 ? (rows + 255) / 256 : 1;
  hipLaunchKernelGGL(fwd_step_kernel<T>, dim3(gx, nbatch), dim3(256), 0, s, dn, blk, w, y, b);
}
template <class T>
void launch_int_update(const SolveNode<T>* dn, int nbatch, int maxni, int maxnb, const T* b, T* part, const T* y, T* w, hipStream_t s) {
  if (nbatch <= 0 || maxni <= 0) return;
  if (maxnb > 0)
    hipLaunchKernelGGL(int_update_partial_kernel<T>, dim3((maxni + 255) / 256, (maxnb + UPD_CS - 1) / UPD_CS, nbatch), dim3(256), 0, s, dn,
                       b, part);
  hipLaunchKernelGGL(int_update_reduce_kernel<T>, dim3((maxni + 255) / 256, nbatch), dim3(256), 0, s, dn, part, y, w);
}
template <class T>
void launch_bwd_step(const SolveNode<T>* dn, int nbatch, int blk, T* w, T* x, hipStream_t s) {
  if (nbatch <= 0) return;
  int rows = blk * HS_PB;
  int gx = rows > 0 ? (rows + 255) / 256 : 1;
  hipLaunchKernelGGL(bwd_step_kernel<T>, dim3(gx, nbatch), dim3(256), 0, s, dn, blk, w, x);
}
template <class T>
void launch_bwd_scatter(const SolveNode<T>* dn, int nbatch, int maxni, T* b, const T* x, hipStream_t s) {
  if (nbatch <= 0 || maxni <= 0) return;
  hipLaunchKernelGGL(bwd_scatter_kernel<T>, dim3((maxni + 255) / 256, nbatch), dim3(256), 0, s, dn, b, x);
}

#define INST(T)                                                                                              \
  template void launch_fwd_gather<T>(const SolveNode<T>*, int, int, const T*, T*, hipStream_t);              \
  template void launch_fwd_step<T>(const SolveNode<T>*, int, int, int, T*, T*, T*, hipStream_t);             \
  template void launch_int_update<T>(const SolveNode<T>*, int, int, int, const T*, T*, const T*, T*, hipStream_t); \
  template void launch_bwd_step<T>(const SolveNode<T>*, int, int, T*, T*, hipStream_t);                      \
  template void launch_bwd_scatter<T>(const SolveNode<T>*, int, int, T*, const T*, hipStream_t);
INST(double)
INST(cplx)
