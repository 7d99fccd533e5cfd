// kernels_mf.hip -- small HBM-bound movers of the matrix-free compressed fronts (hs_mffront.h).
//
// Reference: the sparse blocks `A[int1, bnd2]`, `A[int2, bnd1]`, ... that `_assemble_blocks` keeps next to the children's
// generators (src/factorization.jl:136-137) and that the Gauss transforms factor separately (`X = [0 A12; A21 0]`, :186-192,
// :199-205).  A stencil coupling has at most a few entries per row: its exact factorization by its nonzero rows (or columns)
// is a list of entries written into zero-filled thin factors.
#include "hs_common.h"

template <class T>
__global__ __launch_bounds__(256) void fill_entries_kernel(const HsFillEntry* __restrict__ ent, int cnt, const T* __restrict__ nz, T* __restrict__ out, int ld) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= cnt) return;
  const HsFillEntry f = ent[i];
  out[(size_t)f.row + (size_t)f.col * ld] = f.e >= 0 ? nz[f.e] : Scal<T>::one();
}
template <class T>
void launch_fill_entries(const HsFillEntry* ent, int cnt, const T* nz, T* out, int ld, hipStream_t s) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(fill_entries_kernel<T>, dim3((cnt + 255) / 256), dim3(256), 0, s, ent, cnt, nz, out, ld);
}

// X (n x cols, ld) = columns c0 .. c0+cols-1 of the n x n identity
template <class T>
__global__ __launch_bounds__(256) void identity_cols_kernel(T* __restrict__ X, int ld, int n, int c0, int cols) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * cols) return;
  const int r = (int)(i % n), c = (int)(i / n);
  X[(size_t)r + (size_t)c * ld] = (r == c0 + c) ? Scal<T>::one() : Scal<T>::zero();
}
template <class T>
void launch_identity_cols(T* X, int ld, int n, int c0, int cols, hipStream_t s) {
  if (n <= 0 || cols <= 0) return;
  const size_t tot = (size_t)n * cols;
  hipLaunchKernelGGL(identity_cols_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, X, ld, n, c0, cols);
}

// dst[i] = src[perm[i]]: the values of A in CSR order from the CSC values (the transposition is a fixed permutation of the entries)
template <class T>
__global__ __launch_bounds__(256) void perm_gather_kernel(const T* __restrict__ src, const int64_t* __restrict__ perm, T* __restrict__ dst, int64_t cnt) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < cnt) dst[i] = src[perm[i]];
}
template <class T>
void launch_perm_gather(const T* src, const int64_t* perm, T* dst, int64_t cnt, hipStream_t s) {
  if (cnt <= 0) return;
  hipLaunchKernelGGL(perm_gather_kernel<T>, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, s, src, perm, dst, cnt);
}

// out (rows x cols, ldo) = in (cols x rows, ldi) transposed
template <class T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, int ldi, T* __restrict__ out, int ldo, int rows, int cols) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  const int r = (int)(i % rows), c = (int)(i / rows);
  out[(size_t)r + (size_t)c * ldo] = in[(size_t)c + (size_t)r * ldi];
}
template <class T>
void launch_transpose(const T* in, int ldi, T* out, int ldo, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  const size_t tot = (size_t)rows * cols;
  hipLaunchKernelGGL(transpose_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, in, ldi, out, ldo, rows, cols);
}

// A (rows x cols, ld) = -A
template <class T>
__global__ __launch_bounds__(256) void negate_kernel(T* __restrict__ A, int ld, int rows, int cols) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * cols) return;
  T* p = A + (size_t)(i % rows) + (size_t)(i / rows) * ld;
  *p = -*p;
}
template <class T>
void launch_negate(T* A, int ld, int rows, int cols, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return;
  const size_t tot = (size_t)rows * cols;
  hipLaunchKernelGGL(negate_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, A, ld, rows, cols);
}

#define INST(T)                                                                                              \
  template void launch_fill_entries<T>(const HsFillEntry*, int, const T*, T*, int, hipStream_t);           \
  template void launch_identity_cols<T>(T*, int, int, int, int, hipStream_t);                               \
  template void launch_perm_gather<T>(const T*, const int64_t*, T*, int64_t, hipStream_t);                  \
  template void launch_transpose<T>(const T*, int, T*, int, int, int, hipStream_t);                        \
  template void launch_negate<T>(T*, int, int, int, hipStream_t);
INST(double)
INST(cplx)
