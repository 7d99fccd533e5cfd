// kernels_assemble.hip -- sparse-to-dense front assembly (HBM-bound integer/copy work).
//
// Reference: `Matrix(view(A, I, J))` at src/factorization.jl:33,35,37,40 (leaf) and the eight
// coupling gathers + eight child-Schur slices of `_assemble_blocks` (src/factorization.jl:115-123),
// followed by the child's `S[perm, perm]` (src/factorization.jl:41,73-74).  The author patched
// SparseArrays' getindex for this step (src/mygetindex.jl, test/rungmres.jl:29).
//
// Here a front is built in place in three steps, all grouped over the fronts of one tree level:
//   mark    : own[g] = node, pos[g] = position of global DOF g in the node's front (+ child side)
//   gather  : one thread per front column walks that CSC column of A and drops every entry whose row
//             lives in the same front into Aii/Abi (LF), Aib (UR) or Abb (SB).  For a branch only
//             entries coupling the LEFT child's DOFs with the RIGHT child's are taken -- same-child
//             couplings already sit inside the child's Schur complement.
//   scatter : the child's Schur complement (in the child's own bnd order) is written straight into
//             the parent's blocks through cmap = [int_loc -> int part, bnd_loc -> bnd part]; this fuses
//             S[perm,perm] and the eight S1[a:b,c:d] / S2[...] slice copies into one pass.
#include "hs_common.h"

#define POS_MASK 0x0fffffff
#define SIDE_SHIFT 28

template <class T>
__device__ inline void front_store(const NodeDesc<T>& nd, int P, int Q, T v) {
  if (Q < nd.ni)
    nd.LF[(size_t)P + (size_t)Q * nd.ldl] = v;
  else if (P < nd.ni)
    nd.UR[(size_t)P + (size_t)(Q - nd.ni) * nd.ldu] = v;
  else
    nd.SB[(size_t)(P - nd.ni) + (size_t)(Q - nd.ni) * nd.lds] = v;
}

template <class T>
__device__ inline int side_of(const NodeDesc<T>& nd, int p) {
  if (nd.isleaf) return 0;
  if (nd.spos && p < nd.s_ni) p = nd.spos[p];
  if (p < nd.s_ni) return (p < nd.s_ni1) ? 1 : 2;
  return (p - nd.s_ni < nd.s_nb1) ? 1 : 2;
}

template <class T>
__global__ __launch_bounds__(256) void init_fronts_kernel(const NodeDesc<T>* __restrict__ nodes) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) {
    *nd.info = 0;
    if (nd.growth) *nd.growth = 0;
  }
  if (i < nd.pivrows) nd.rperm[i] = i;
}

template <class T>
__global__ __launch_bounds__(256) void mark_kernel(const NodeDesc<T>* __restrict__ nodes, int* __restrict__ own, int* __restrict__ pos) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= nd.m) return;
  int g = nd.fidx[p];
  own[g] = nd.node;
  pos[g] = p | (side_of(nd, p) << SIDE_SHIFT);
}

template <class T>
__global__ __launch_bounds__(256) void gather_kernel(const NodeDesc<T>* __restrict__ nodes, const int64_t* __restrict__ colptr,
                                                     const int32_t* __restrict__ rowval, const T* __restrict__ nzval,
                                                     const int* __restrict__ own, const int* __restrict__ pos) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= nd.m) return;
  int g = nd.fidx[c];
  int side_c = side_of(nd, c);
  int64_t e0 = colptr[g], e1 = colptr[g + 1];
  for (int64_t e = e0; e < e1; ++e) {
    int r = rowval[e];
    if (own[r] != nd.node) continue;
    int pr = pos[r];
    if (!nd.isleaf && (pr >> SIDE_SHIFT) == side_c) continue;
    front_store(nd, pr & POS_MASK, c, nzval[e]);
  }
}

template <class T>
__global__ __launch_bounds__(256) void scatter_kernel(const NodeDesc<T>* __restrict__ nodes, const ScatterDesc<T>* __restrict__ scs) {
  const ScatterDesc<T> sc = scs[blockIdx.y];
  const int nt = (sc.nbc + 63) / 64;
  if ((int)blockIdx.x >= nt * nt) return;
  const NodeDesc<T> nd = nodes[sc.parent];
  const int ta = blockIdx.x % nt, tb = blockIdx.x / nt;
  const int a = ta * 64 + (threadIdx.x & 63);
  if (a >= sc.nbc) return;
  const int P = sc.cmap[a];
  if (P < 0) return;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    int b = tb * 64 + (threadIdx.x >> 6) + 4 * i;
    if (b >= sc.nbc) break;
    int Q = sc.cmap[b];
    if (Q < 0) continue;
    front_store(nd, P, Q, sc.S[(size_t)a + (size_t)b * sc.lds]);
  }
}

template <class T>
void launch_init_fronts(const NodeDesc<T>* dnodes, int nbatch, int maxni, hipStream_t s) {
  if (nbatch <= 0) return;
  hipLaunchKernelGGL(init_fronts_kernel<T>, dim3(maxni > 0 ? (maxni + 255) / 256 : 1, nbatch), dim3(256), 0, s, dnodes);
}
template <class T>
void launch_mark(const NodeDesc<T>* dnodes, int nbatch, int maxm, int* own, int* pos, hipStream_t s) {
  if (nbatch <= 0 || maxm <= 0) return;
  hipLaunchKernelGGL(mark_kernel<T>, dim3((maxm + 255) / 256, nbatch), dim3(256), 0, s, dnodes, own, pos);
}
template <class T>
void launch_gather(const NodeDesc<T>* dnodes, int nbatch, int maxm, const int64_t* colptr, const int32_t* rowval, const T* nzval,
                   const int* own, const int* pos, hipStream_t s) {
  if (nbatch <= 0 || maxm <= 0) return;
  hipLaunchKernelGGL(gather_kernel<T>, dim3((maxm + 255) / 256, nbatch), dim3(256), 0, s, dnodes, colptr, rowval, nzval, own, pos);
}
template <class T>
void launch_scatter(const NodeDesc<T>* dnodes, const ScatterDesc<T>* dsc, int nsc, int maxnbc, hipStream_t s) {
  if (nsc <= 0 || maxnbc <= 0) return;
  int nt = (maxnbc + 63) / 64;
  hipLaunchKernelGGL(scatter_kernel<T>, dim3(nt * nt, nsc), dim3(256), 0, s, dnodes, dsc);
}

#define INST(T)                                                                                                          \
  template void launch_init_fronts<T>(const NodeDesc<T>*, int, int, hipStream_t);                                        \
  template void launch_mark<T>(const NodeDesc<T>*, int, int, int*, int*, hipStream_t);                                   \
  template void launch_gather<T>(const NodeDesc<T>*, int, int, const int64_t*, const int32_t*, const T*, const int*,    \
                                 const int*, hipStream_t);                                                               \
  template void launch_scatter<T>(const NodeDesc<T>*, const ScatterDesc<T>*, int, int, hipStream_t);
INST(double)
INST(cplx)
