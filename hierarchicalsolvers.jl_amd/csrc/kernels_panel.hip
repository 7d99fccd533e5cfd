// kernels_panel.hip -- in-front LU panels with tournament pivoting, row swaps, diagonal-block solves.
//
// Replaces the dense `\` and `/` of the reference's interior eliminations -- `L = Abi / D`,
// `R = D \ Aib` (src/factorization.jl:36-37) and every `A11\`, `/A11`, `S22\`, `/S22` inside
// blockfactor/blockldiv/blockrdiv (src/blockmatrix.jl:118,162-170,177-185) -- which in Julia are
// LAPACK getrf+getrs (partial pivoting) on the interior block.  Here the interior block Aii of a
// front is factored ONCE, P*Aii = L*U, 32 columns at a time:
//
//   1. tournament pivoting (CALU): every 256-row chunk of the panel runs Gaussian elimination with
//      partial pivoting entirely in registers (one row per thread, 32 columns = 64 VGPRs) and
//      nominates its 32 pivot rows; nominees play off in further rounds until 32 rows remain.
//      The last round IS partial pivoting on the surviving rows, so eliminating them in that order
//      without further pivoting reproduces it.  Pivot candidates are restricted to the rows of Aii
//      (rows < ni), exactly like `\` on D / A11 in the reference.
//   2. panel_pivot: turn the winners into LAPACK-style swaps, swap the panel's rows, LU the 32x32
//      top block in LDS and invert its L and U factors (kept for TRSM-by-GEMM and for ldiv!).
//   3. panel_l21: every row below (including the Abi rows, which gives L_bi = Abi*U^-1 for free)
//      is multiplied by inv(U11).
//   laswp / trsm_blk apply the swaps and inv(L11) to the other columns as the recursion demands.
//
// All kernels are "grouped": blockIdx.y selects the front of the current level batch.
#include "hs_common.h"

// ------------------------------------------------------------------------------------------------
// argmax of (key, idx) over a 256-thread block; ties -> smallest idx.  Returns idx of the max key
// (key < 0 means "not a candidate"; result -1 if no candidate has key > 0).
// ------------------------------------------------------------------------------------------------
__device__ inline void wave_argmax(double& key, int& idx) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    double ok = __shfl_xor(key, off, 64);
    int oi = __shfl_xor(idx, off, 64);
    if (ok > key || (ok == key && oi < idx)) {
      key = ok;
      idx = oi;
    }
  }
}

template <class T>
__global__ __launch_bounds__(HS_CHUNK) void tournament_kernel(const NodeDesc<T>* __restrict__ nodes, int pb, int round) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  // candidate counts per round: cnt_0 = ni - c0, cnt_{r+1} = ceil(cnt_r / 256) * 32
  int cnt = nd.ni - c0;
  int nch = (cnt + HS_CHUNK - 1) / HS_CHUNK;
  for (int r = 0; r < round; ++r) {
    if (nch == 1) return;  // this front finished in an earlier round
    cnt = nch * HS_PB;
    nch = (cnt + HS_CHUNK - 1) / HS_CHUNK;
  }
  const int chunk = blockIdx.x;
  if (chunk >= nch) return;
  const bool final_round = (nch == 1);
  const int* cin = (round & 1) ? nd.cand1 : nd.cand0;   // written by round-1
  int* cout = final_round ? nd.pivlist : ((round & 1) ? nd.cand0 : nd.cand1);
  const int obase = final_round ? 0 : chunk * HS_PB;

  const int t = threadIdx.x;
  const int q = chunk * HS_CHUNK + t;
  int row = -1;
  if (q < cnt) row = (round == 0) ? (c0 + q) : cin[q];
  bool live = row >= 0;

  T a[HS_PB];
#pragma unroll
  for (int j = 0; j < HS_PB; ++j) {
    a[j] = Scal<T>::zero();
    if (live && j < w) a[j] = nd.LF[(size_t)row + (size_t)(c0 + j) * nd.ldl];
  }

  __shared__ T prow[HS_PB];
  __shared__ double wkey[4];
  __shared__ int widx[4];
  __shared__ int s_win;

#pragma unroll
  for (int k = 0; k < HS_PB; ++k) {
    if (k < w) {
      double key = live ? Scal<T>::abs1(a[k]) : -1.0;
      int idx = t;
      wave_argmax(key, idx);
      if ((t & 63) == 0) {
        wkey[t >> 6] = key;
        widx[t >> 6] = idx;
      }
      __syncthreads();
      if (t == 0) {
        double bk = wkey[0];
        int bi = widx[0];
#pragma unroll
        for (int v = 1; v < 4; ++v)
          if (wkey[v] > bk || (wkey[v] == bk && widx[v] < bi)) {
            bk = wkey[v];
            bi = widx[v];
          }
        s_win = (bk > 0.0) ? bi : -1;
      }
      __syncthreads();
      const int win = s_win;
      if (win >= 0) {
        if (t == win) {
#pragma unroll
          for (int j = 0; j < HS_PB; ++j) prow[j] = a[j];
          cout[obase + k] = row;
          live = false;
        }
        __syncthreads();
        if (live) {
          T l = a[k] / prow[k];
#pragma unroll
          for (int j = 0; j < HS_PB; ++j)
            if (j > k) a[j] = Scal<T>::fnma(l, prow[j], a[j]);
        }
      } else {
        if (t == 0) cout[obase + k] = -1;  // column is exactly zero below the diagonal: singular
      }
      __syncthreads();
    } else {
      if (t == 0) cout[obase + k] = -1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// panel_pivot: one workgroup per front.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void panel_pivot_kernel(const NodeDesc<T>* __restrict__ nodes, int pb) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  const int t = threadIdx.x;

  __shared__ int s_piv[HS_PB];            // swap target of row c0+k
  __shared__ int s_pos[2 * HS_PB];        // positions touched so far
  __shared__ int s_who[2 * HS_PB];        // original row now living at s_pos[i]
  __shared__ T s_a[HS_PB][HS_PB + 1];     // [row][col]
  __shared__ T s_il[HS_PB][HS_PB + 1];
  __shared__ T s_iu[HS_PB][HS_PB + 1];

  if (t == 0) {
    // winners (original row ids, in elimination order) -> sequential swaps.  Track which original
    // row currently sits at every touched position.
    int ntouch = 0;
    for (int k = 0; k < w; ++k) {
      int r = nd.pivlist[k];
      int target = c0 + k;
      int p;
      if (r < 0) {
        p = target;  // no pivot: leave the row, flag singular
        int old = *nd.info;
        if (old == 0 || old > c0 + k + 1) *nd.info = c0 + k + 1;
      } else {
        // current position of original row r
        p = r;
        for (int i = 0; i < ntouch; ++i)
          if (s_who[i] == r) p = s_pos[i];
      }
      s_piv[k] = p;
      if (p != target) {
        // who is at target now?
        int qrow = target;
        int it = -1, ip = -1;
        for (int i = 0; i < ntouch; ++i) {
          if (s_pos[i] == target) { qrow = s_who[i]; it = i; }
        }
        for (int i = 0; i < ntouch; ++i)
          if (s_pos[i] == p) ip = i;
        // after the swap: position target holds r, position p holds qrow
        if (it < 0) { it = ntouch++; s_pos[it] = target; }
        s_who[it] = r;
        if (ip < 0) { ip = ntouch++; s_pos[ip] = p; }
        s_who[ip] = qrow;
      }
      nd.ipiv[c0 + k] = p;
    }
  }
  __syncthreads();
  // swap the panel's own columns (thread j owns column c0+j), sequentially over k
  if (t < w) {
    T* col = nd.LF + (size_t)(c0 + t) * nd.ldl;
    for (int k = 0; k < w; ++k) {
      int p = s_piv[k];
      if (p != c0 + k) {
        T tmp = col[c0 + k];
        col[c0 + k] = col[p];
        col[p] = tmp;
      }
    }
  } else if (t == 64) {
    // the accumulated row permutation is one more "column" that takes the same swaps
    for (int k = 0; k < w; ++k) {
      int p = s_piv[k];
      if (p != c0 + k) {
        int tmp = nd.rperm[c0 + k];
        nd.rperm[c0 + k] = nd.rperm[p];
        nd.rperm[p] = tmp;
      }
    }
  }
  __syncthreads();
  // load the top w x w block (identity-padded to 32)
  for (int e = t; e < HS_PB * HS_PB; e += 256) {
    int i = e & 31, j = e >> 5;
    T v = (i == j) ? Scal<T>::one() : Scal<T>::zero();
    if (i < w && j < w) v = nd.LF[(size_t)(c0 + i) + (size_t)(c0 + j) * nd.ldl];
    s_a[i][j] = v;
  }
  __syncthreads();
  // unpivoted LU (pivot order fixed by the tournament)
  for (int k = 0; k < HS_PB; ++k) {
    T piv = s_a[k][k];
    bool zero_piv = (Scal<T>::abs1(piv) == 0.0);
    if (t < HS_PB && t > k && !zero_piv) s_a[t][k] = s_a[t][k] / piv;
    __syncthreads();
    if (!zero_piv) {
      for (int e = t; e < HS_PB * HS_PB; e += 256) {
        int i = e & 31, j = e >> 5;
        if (i > k && j > k) s_a[i][j] = Scal<T>::fnma(s_a[i][k], s_a[k][j], s_a[i][j]);
      }
    } else if (t == 0 && k < w) {
      int old = *nd.info;
      if (old == 0 || old > c0 + k + 1) *nd.info = c0 + k + 1;
    }
    __syncthreads();
  }
  // inverses of L (unit lower) and U (upper): thread j < 32 solves column j
  if (t < HS_PB) {
    const int j = t;
    // L * x = e_j  (forward)
    for (int i = 0; i < HS_PB; ++i) {
      T s = (i == j) ? Scal<T>::one() : Scal<T>::zero();
      for (int p = j; p < i; ++p) s = Scal<T>::fnma(s_a[i][p], s_il[p][j], s);
      s_il[i][j] = (i < j) ? Scal<T>::zero() : s;
    }
  } else if (t >= 64 && t < 64 + HS_PB) {
    const int j = t - 64;
    // U * x = e_j  (backward); zero pivots are treated as 1 (front is already flagged singular)
    for (int i = HS_PB - 1; i >= 0; --i) {
      T s = (i == j) ? Scal<T>::one() : Scal<T>::zero();
      for (int p = i + 1; p <= j; ++p) s = Scal<T>::fnma(s_a[i][p], s_iu[p][j], s);
      T d = s_a[i][i];
      if (Scal<T>::abs1(d) == 0.0) d = Scal<T>::one();
      s_iu[i][j] = (i > j) ? Scal<T>::zero() : s / d;
    }
  }
  __syncthreads();
  for (int e = t; e < HS_PB * HS_PB; e += 256) {
    int i = e & 31, j = e >> 5;
    if (i < w && j < w) nd.LF[(size_t)(c0 + i) + (size_t)(c0 + j) * nd.ldl] = s_a[i][j];
    nd.invL[(size_t)pb * HS_PB * HS_PB + e] = s_il[i][j];
    nd.invU[(size_t)pb * HS_PB * HS_PB + e] = s_iu[i][j];
  }
}

// ------------------------------------------------------------------------------------------------
// panel_l21: rows below the diagonal block, one row per thread:  x <- x * inv(U11)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void panel_l21_kernel(const NodeDesc<T>* __restrict__ nodes, int pb) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  const int r0 = c0 + w;
  if ((int)blockIdx.x * 256 >= nd.m - r0) return;
  __shared__ T s_iu[HS_PB * HS_PB];  // column-major, ld 32
  for (int e = threadIdx.x; e < HS_PB * HS_PB; e += 256) s_iu[e] = nd.invU[(size_t)pb * HS_PB * HS_PB + e];
  __syncthreads();
  const int row = r0 + blockIdx.x * 256 + threadIdx.x;
  if (row >= nd.m) return;
  T* base = nd.LF + (size_t)row + (size_t)c0 * nd.ldl;
  T a[HS_PB];
#pragma unroll
  for (int j = 0; j < HS_PB; ++j) a[j] = (j < w) ? base[(size_t)j * nd.ldl] : Scal<T>::zero();
#pragma unroll
  for (int j = 0; j < HS_PB; ++j) {
    if (j < w) {
      T s = Scal<T>::zero();
#pragma unroll
      for (int i = 0; i < HS_PB; ++i)
        if (i <= j) s = Scal<T>::fma(a[i], s_iu[i + j * HS_PB], s);
      base[(size_t)j * nd.ldl] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// laswp: apply swaps ipiv[k0:k1) to columns [c0, c1) of LF or UR (one column per thread)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void laswp_kernel(const NodeDesc<T>* __restrict__ nodes, int mat, int c0, int c1, int k0, int k1) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  T* p;
  int ld, rows, cols;
  mat_of(nodes + blockIdx.y, mat, p, ld, rows, cols);
  c1 = min(c1, cols);
  k1 = min(k1, nd.ni);
  const int c = c0 + blockIdx.x * 256 + threadIdx.x;
  if (c >= c1 || k0 >= k1) return;
  T* col = p + (size_t)c * ld;
  for (int k = k0; k < k1; ++k) {
    int pv = nd.ipiv[k];
    if (pv != k) {
      T tmp = col[k];
      col[k] = col[pv];
      col[pv] = tmp;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// trsm_blk: X[r0:r0+32, c0:c1) <- inv(L11[r0/32]) * X[r0:r0+32, c0:c1)   (one column per thread)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void trsm_blk_kernel(const NodeDesc<T>* __restrict__ nodes, int mat, int r0, int c0, int c1) {
  const NodeDesc<T> nd = nodes[blockIdx.y];
  if (r0 >= nd.ni) return;
  T* p;
  int ld, rows, cols;
  mat_of(nodes + blockIdx.y, mat, p, ld, rows, cols);
  c1 = min(c1, cols);
  if (c0 + (int)blockIdx.x * 256 >= c1) return;
  const int w = min(HS_PB, nd.ni - r0);
  __shared__ T s_il[HS_PB * HS_PB];
  const int pb = r0 / HS_PB;
  for (int e = threadIdx.x; e < HS_PB * HS_PB; e += 256) s_il[e] = nd.invL[(size_t)pb * HS_PB * HS_PB + e];
  __syncthreads();
  const int c = c0 + blockIdx.x * 256 + threadIdx.x;
  if (c >= c1) return;
  T* x = p + (size_t)r0 + (size_t)c * ld;
  T a[HS_PB];
#pragma unroll
  for (int i = 0; i < HS_PB; ++i) a[i] = (i < w) ? x[i] : Scal<T>::zero();
#pragma unroll
  for (int i = 0; i < HS_PB; ++i) {
    if (i < w) {
      T s = Scal<T>::zero();
#pragma unroll
      for (int j = 0; j < HS_PB; ++j)
        if (j <= i) s = Scal<T>::fma(s_il[i + j * HS_PB], a[j], s);
      x[i] = s;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <class T>
void launch_tournament_round(const NodeDesc<T>* dnodes, int nbatch, int pb, int round, int maxchunks, hipStream_t s) {
  if (nbatch <= 0 || maxchunks <= 0) return;
  hipLaunchKernelGGL(tournament_kernel<T>, dim3(maxchunks, nbatch), dim3(HS_CHUNK), 0, s, dnodes, pb, round);
}
template <class T>
void launch_panel_pivot(const NodeDesc<T>* dnodes, int nbatch, int pb, hipStream_t s) {
  if (nbatch <= 0) return;
  hipLaunchKernelGGL(panel_pivot_kernel<T>, dim3(1, nbatch), dim3(256), 0, s, dnodes, pb);
}
template <class T>
void launch_panel_l21(const NodeDesc<T>* dnodes, int nbatch, int pb, int maxrows, hipStream_t s) {
  if (nbatch <= 0 || maxrows <= 0) return;
  hipLaunchKernelGGL(panel_l21_kernel<T>, dim3((maxrows + 255) / 256, nbatch), dim3(256), 0, s, dnodes, pb);
}
template <class T>
void launch_laswp(const NodeDesc<T>* dnodes, int nbatch, int mat, int c0, int c1, int k0, int k1, int maxcols, hipStream_t s) {
  if (nbatch <= 0 || maxcols <= 0 || k1 <= k0) return;
  hipLaunchKernelGGL(laswp_kernel<T>, dim3((maxcols + 255) / 256, nbatch), dim3(256), 0, s, dnodes, mat, c0, c1, k0, k1);
}
template <class T>
void launch_trsm_blk(const NodeDesc<T>* dnodes, int nbatch, int mat, int r0, int c0, int c1, int maxcols, hipStream_t s) {
  if (nbatch <= 0 || maxcols <= 0) return;
  hipLaunchKernelGGL(trsm_blk_kernel<T>, dim3((maxcols + 255) / 256, nbatch), dim3(256), 0, s, dnodes, mat, r0, c0, c1);
}

#define INST(T)                                                                                         \
  template void launch_tournament_round<T>(const NodeDesc<T>*, int, int, int, int, hipStream_t);        \
  template void launch_panel_pivot<T>(const NodeDesc<T>*, int, int, hipStream_t);                       \
  template void launch_panel_l21<T>(const NodeDesc<T>*, int, int, int, hipStream_t);                    \
  template void launch_laswp<T>(const NodeDesc<T>*, int, int, int, int, int, int, int, hipStream_t);    \
  template void launch_trsm_blk<T>(const NodeDesc<T>*, int, int, int, int, int, int, hipStream_t);
INST(double)
INST(cplx)
