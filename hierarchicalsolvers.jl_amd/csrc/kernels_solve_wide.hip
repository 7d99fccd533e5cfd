// kernels_solve_wide.hip -- the triangular sweeps of ldiv! advancing 256 columns per launch.
//
// Reference: `_lsolve!` / `_dsolve!` / `_rsolve!` (src/factornode.jl:77-99); kernels_solve.hip has the 32-column
// version of the same right-looking sweeps.  A 32-column step moves at most 8 MB (a 32,768-row panel), far too little to
// be bandwidth-bound, and a 256-column step that solves its diagonal block with 8 dependent 32 x 32 matvecs spends
// ~40 us in that chain (measured: no gain).  So the factorization also leaves the INVERSES of the 256 x 256 triangular
// diagonal blocks of L and U (inv256_kernel, built from the stored 32 x 32 inverses by block forward / backward
// substitution, +6 % factor memory); a sweep step is then  y = inv256 * w  (one 256 x 256 matvec, every workgroup
// redundantly, the block comes from L2) followed by the 256-column panel update of the workgroup's own 256 rows:
// up to 64 MB per launch and 8x fewer launches.
#include "hs_common.h"

#define HS_SW 256  // columns per launch

// ------------------------------------------------------------------------------------------------
// inverses of the 256 x 256 diagonal blocks.  grid.x = (256-block, block column j of it), grid.y = front, grid.z = L / U
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void inv256_kernel(const SolveNode<T>* __restrict__ nodes, int first_block) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* X = reinterpret_cast<T*>(smem_raw);   // 8 blocks of 32 x 32 (column-major): X_kj of the current block column
  T* S = X + 8 * HS_PB * HS_PB;            // 32 x 32 accumulator
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int upper = blockIdx.z;
  const int b256 = first_block + (blockIdx.x >> 3), jc = blockIdx.x & 7;
  const int c0 = b256 * HS_SW;
  if (c0 >= nd.ni) return;
  const int wl = min(HS_SW, nd.ni - c0);
  const int nsb = (wl + HS_PB - 1) / HS_PB;
  if (jc >= nsb) return;
  const int t = threadIdx.x;
  const T* inv32 = upper ? nd.invU : nd.invL;
  T* out = (upper ? nd.inv256U : nd.inv256L) + (size_t)b256 * HS_SW * HS_SW;
  const int sb0 = c0 / HS_PB;  // first 32-block of this wide block
  // X_jj = stored inverse of the diagonal 32-block
  for (int e = t; e < HS_PB * HS_PB; e += 256) X[jc * 1024 + e] = inv32[(size_t)(sb0 + jc) * 1024 + e];
  __syncthreads();
  const int istep = upper ? -1 : 1;
  for (int i = jc + istep; i >= 0 && i < nsb; i += istep) {
    // S = sum_k A_ik * X_kj over the already known blocks k between j and i
    T acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = Scal<T>::zero();
    const int k0 = upper ? i + 1 : jc, k1 = upper ? jc : i - 1;
    for (int k = k0; k <= k1; ++k) {
      // stage A_ik (32 x 32, zero outside the block's extent) in the S area: every output needs a whole row of it
      const T* Aik = nd.LF + (size_t)(c0 + i * HS_PB) + (size_t)(c0 + k * HS_PB) * nd.ldl;
      const int wk = min(HS_PB, wl - k * HS_PB), wi = min(HS_PB, wl - i * HS_PB);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = t + 256 * u, a = e & 31, q = e >> 5;
        S[e] = (a < wi && q < wk) ? Aik[(size_t)a + (size_t)q * nd.ldl] : Scal<T>::zero();
      }
      __syncthreads();
      const T* Xk = X + k * 1024;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = t + 256 * u, a = e & 31, b = e >> 5;
        T s = acc[u];
#pragma unroll 8
        for (int q = 0; q < HS_PB; ++q) s = Scal<T>::fma(S[a + q * HS_PB], Xk[q + b * HS_PB], s);
        acc[u] = s;
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) S[t + 256 * u] = acc[u];
    __syncthreads();
    // X_ij = -inv32_i * S
    const T* Ii = inv32 + (size_t)(sb0 + i) * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = t + 256 * u, a = e & 31, b = e >> 5;
      T s = Scal<T>::zero();
      for (int q = 0; q < HS_PB; ++q) s = Scal<T>::fma(Ii[a + q * HS_PB], S[q + b * HS_PB], s);
      X[i * 1024 + e] = Scal<T>::zero() - s;
    }
    __syncthreads();
  }
  // write the WHOLE block column (256 rows x 32 columns): the blocks of the triangle, zeros elsewhere -- the TRSM base
  // case multiplies by the full 256 x 256 block (GemmOp::ainv 3..6)
  const int ilo = upper ? 0 : jc, ihi = upper ? jc : nsb - 1;
  for (int i = 0; i < HS_SW / HS_PB; ++i)
    for (int e = t; e < HS_PB * HS_PB; e += 256) {
      const int a = e & 31, b = e >> 5;
      out[(size_t)(i * HS_PB + a) + (size_t)(jc * HS_PB + b) * HS_SW] = (i >= ilo && i <= ihi) ? X[i * 1024 + e] : Scal<T>::zero();
    }
}

// forward: y_blk = L[blk,blk]^-1 * w_blk ; rows below -= L[:, blk] * y_blk   (rows >= ni are the Abi*U^-1 rows: they update rhs[bnd])
template <class T>
__global__ __launch_bounds__(256) void fwd_wide_kernel(const SolveNode<T>* __restrict__ nodes, int blk, T* __restrict__ w, T* __restrict__ y,
                                                       T* __restrict__ b) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int c0 = blk * HS_SW;
  if (c0 >= nd.ni) return;
  const int wl = min(HS_SW, nd.ni - c0);
  const int r0 = c0 + wl;
  if (blockIdx.x > 0 && (int)blockIdx.x * 256 >= nd.mrows - r0) return;
  __shared__ T s_w[HS_SW];
  __shared__ T s_y[HS_SW];
  const int t = threadIdx.x;
  s_w[t] = (t < wl) ? w[nd.woff + c0 + t] : Scal<T>::zero();
  __syncthreads();
  {
    const T* iv = nd.inv256L + (size_t)blk * HS_SW * HS_SW;
    T s = Scal<T>::zero();
    if (t < wl) {
      const int jn = min(wl, (t / HS_PB + 1) * HS_PB);  // lower triangle, whole 32-blocks
#pragma unroll 8
      for (int j = 0; j < jn; ++j) s = Scal<T>::fma(iv[(size_t)t + (size_t)j * HS_SW], s_w[j], s);
    }
    s_y[t] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < wl) y[nd.woff + c0 + t] = s_y[t];
  const int r = r0 + blockIdx.x * 256 + t;
  if (r >= nd.mrows) return;
  const T* a = nd.LF + (size_t)r + (size_t)c0 * nd.ldl;
  T acc = Scal<T>::zero();
#pragma unroll 8
  for (int j = 0; j < wl; ++j) acc = Scal<T>::fma(a[(size_t)j * nd.ldl], s_y[j], acc);
  if (r < nd.ni) {
    w[nd.woff + r] = w[nd.woff + r] - acc;
  } else {
    const int g = nd.fidx[r];
    b[g] = b[g] - acc;
  }
}

// backward: x_blk = U[blk,blk]^-1 * w_blk ; rows above -= U[:, blk] * x_blk
template <class T>
__global__ __launch_bounds__(256) void bwd_wide_kernel(const SolveNode<T>* __restrict__ nodes, int blk, T* __restrict__ w, T* __restrict__ x) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int c0 = blk * HS_SW;
  if (c0 >= nd.ni) return;
  const int wl = min(HS_SW, nd.ni - c0);
  if (blockIdx.x > 0 && (int)blockIdx.x * 256 >= c0) return;
  __shared__ T s_w[HS_SW];
  __shared__ T s_x[HS_SW];
  const int t = threadIdx.x;
  s_w[t] = (t < wl) ? w[nd.woff + c0 + t] : Scal<T>::zero();
  __syncthreads();
  {
    const T* iv = nd.inv256U + (size_t)blk * HS_SW * HS_SW;
    T s = Scal<T>::zero();
    if (t < wl) {
#pragma unroll 8
      for (int j = t / HS_PB * HS_PB; j < wl; ++j) s = Scal<T>::fma(iv[(size_t)t + (size_t)j * HS_SW], s_w[j], s);  // upper triangle, whole 32-blocks
    }
    s_x[t] = s;
  }
  __syncthreads();
  if (blockIdx.x == 0 && t < wl) x[nd.woff + c0 + t] = s_x[t];
  const int r = blockIdx.x * 256 + t;
  if (r >= c0) return;
  const T* a = nd.LF + (size_t)r + (size_t)c0 * nd.ldl;
  T acc = Scal<T>::zero();
#pragma unroll 8
  for (int j = 0; j < wl; ++j) acc = Scal<T>::fma(a[(size_t)j * nd.ldl], s_x[j], acc);
  w[nd.woff + r] = w[nd.woff + r] - acc;
}

template <class T>
void launch_inv256(const SolveNode<T>* dn, int nbatch, int maxni, hipStream_t s, int only_block) {
  if (nbatch <= 0 || maxni <= 0) return;
  const int first = only_block >= 0 ? only_block : 0;
  const int nb256 = only_block >= 0 ? 1 : (maxni + HS_SW - 1) / HS_SW;
  if (first * HS_SW >= maxni) return;
  constexpr int lds_bytes = (int)(sizeof(T) * 9 * HS_PB * HS_PB);
  static bool attr_set = false;
  if (!attr_set) {  // 72 KiB (double) / 144 KiB (complex) of LDS per workgroup needs the opt-in
    (void)hipFuncSetAttribute((const void*)inv256_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(inv256_kernel<T>, dim3(nb256 * 8, nbatch, 2), dim3(256), lds_bytes, s, dn, first);
}
template <class T>
void launch_fwd_wide(const SolveNode<T>* dn, int nbatch, int blk, int maxm, T* w, T* y, T* b, hipStream_t s) {
  if (nbatch <= 0) return;
  const int rows = maxm - blk * HS_SW;
  const int gx = rows > 0 ? (rows + 255) / 256 : 1;
  hipLaunchKernelGGL(fwd_wide_kernel<T>, dim3(gx, nbatch), dim3(256), 0, s, dn, blk, w, y, b);
}
template <class T>
void launch_bwd_wide(const SolveNode<T>* dn, int nbatch, int blk, T* w, T* x, hipStream_t s) {
  if (nbatch <= 0) return;
  const int rows = blk * HS_SW;
  const int gx = rows > 0 ? (rows + 255) / 256 : 1;
  hipLaunchKernelGGL(bwd_wide_kernel<T>, dim3(gx, nbatch), dim3(256), 0, s, dn, blk, w, x);
}
int hs_solve_wide_cols() { return HS_SW; }

template void launch_inv256<double>(const SolveNode<double>*, int, int, hipStream_t, int);
template void launch_inv256<cplx>(const SolveNode<cplx>*, int, int, hipStream_t, int);
template void launch_fwd_wide<double>(const SolveNode<double>*, int, int, int, double*, double*, double*, hipStream_t);
template void launch_fwd_wide<cplx>(const SolveNode<cplx>*, int, int, int, cplx*, cplx*, cplx*, hipStream_t);
template void launch_bwd_wide<double>(const SolveNode<double>*, int, int, double*, double*, hipStream_t);
template void launch_bwd_wide<cplx>(const SolveNode<cplx>*, int, int, cplx*, cplx*, hipStream_t);
