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
//   laswp applies the swaps to the other columns as the recursion demands (inv(L11) goes through the MFMA tile code: trsm_inv_kernel).
//
// All kernels are "grouped": blockIdx.y selects the front of the current level batch.
#include <cstdlib>
#include "hs_common.h"

// ------------------------------------------------------------------------------------------------
// Pivot search: max over the 256 rows of a chunk of a 64-bit key = |a_k| bits with the low 8 bits
// replaced by 255 - thread id (so ties, and values equal to 44 mantissa bits, go to the lowest row).
// In-wave: 4 DPP steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror) + 4 readlanes, no LDS;
// across the 4 waves: one LDS slot each and ONE barrier per elimination step.
// ------------------------------------------------------------------------------------------------
__device__ inline unsigned long long dpp_max_u64(unsigned long long v, const int ctrl_sel) {
  int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
  int olo, ohi;
  switch (ctrl_sel) {
    case 0: olo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); break;   // quad_perm [1,0,3,2]
    case 1: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); break;   // quad_perm [2,3,0,1]
    case 2: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); break; // row_half_mirror
    default: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); break; // row_mirror
  }
  unsigned long long o = ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo;
  return o > v ? o : v;
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v) {
  v = dpp_max_u64(v, 0);
  v = dpp_max_u64(v, 1);
  v = dpp_max_u64(v, 2);
  v = dpp_max_u64(v, 3);  // every row of 16 lanes now holds its maximum
  int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
  unsigned long long m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned long long x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, 16 * r) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, 16 * r);
    m = x > m ? x : m;
  }
  return m;
}

__device__ inline double lane_bcast(double v, int lane) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_readlane(hi, lane), __builtin_amdgcn_readlane(lo, lane));
}
__device__ inline cplx lane_bcast(cplx v, int lane) { return {lane_bcast(v.re, lane), lane_bcast(v.im, lane)}; }

template <class T>
__global__ __launch_bounds__(HS_CHUNK) void tournament_kernel(const NodeDesc<T>* __restrict__ nodes, int pb, int round) {
  __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win instruction issue over co-resident GEMM waves
  const NodeDesc<T> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  // candidate counts per round: cnt_0 = pivrows - c0, cnt_{r+1} = ceil(cnt_r / 256) * 32
  int cnt = nd.pivrows - c0;
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

  __shared__ T prow[2][HS_PB + 1];              // winner's row (+ reciprocal of the pivot), double-buffered by k parity
  __shared__ unsigned long long wkey[2][4];     // per-wave maxima, double-buffered by k parity

#pragma unroll
  for (int k = 0; k < HS_PB; ++k) {
    if (k < w) {
      unsigned long long key = 0;
      if (live) key = ((unsigned long long)__double_as_longlong(Scal<T>::abs1(a[k])) & ~0xffull) | (unsigned long long)(255 - t);
      unsigned long long wm = wave_max_u64(key);
      if ((t & 63) == 0) wkey[k & 1][t >> 6] = wm;
      __syncthreads();
      unsigned long long best = wkey[k & 1][0];
#pragma unroll
      for (int v = 1; v < 4; ++v) best = wkey[k & 1][v] > best ? wkey[k & 1][v] : best;
      const int win = ((best >> 8) != 0) ? 255 - (int)(best & 0xff) : -1;  // wave-uniform, identical in all waves
      if (win >= 0) {
        if (t == win) {
#pragma unroll
          for (int j = 0; j < HS_PB; ++j) prow[k & 1][j] = a[j];
          prow[k & 1][HS_PB] = Scal<T>::one() / a[k];
          cout[obase + k] = row;
          live = false;
        }
        __syncthreads();
        if (live) {
          T l = a[k] * prow[k & 1][HS_PB];
#pragma unroll
          for (int j = 0; j < HS_PB; ++j)
            if (j > k) a[j] = Scal<T>::fnma(l, prow[k & 1][j], a[j]);
        }
      } else {
        if (t == 0) cout[obase + k] = -1;  // column is exactly zero below the diagonal: singular
      }
    } else {
      if (t == 0) cout[obase + k] = -1;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// panel_pivot: one workgroup per front.
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void panel_pivot_kernel(const NodeDesc<T>* __restrict__ nodes, int pb, int fuse) {
  __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win instruction issue over co-resident GEMM waves
  const NodeDesc<T> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  const int t = threadIdx.x;
  // fuse & 1: first panel of a 64-column pair -- also swap the NEXT 32 columns and leave U12 = inv(L11)*A12 in them
  //           (panel_l21 then applies the rank-32 update to those columns): no laswp / TRSM / GEMM launches in the pair
  // fuse & 2: second panel of the pair -- also swap the PREVIOUS 32 columns (the left-looking swap of the pair)
  const int w2 = (fuse & 1) ? max(0, min(HS_PB, nd.ni - (c0 + HS_PB))) : 0;

  __shared__ int s_piv[HS_PB];     // swap target of row c0+k
  __shared__ int s_where[HS_PB];   // current position of the row that started at top position c0+i
  __shared__ int s_what[HS_PB];    // row (by starting position) now sitting at top position c0+i
  __shared__ T s_a[HS_PB][HS_PB + 1];   // [row][col]
  __shared__ T s_il[HS_PB][HS_PB + 1];
  __shared__ T s_iu[HS_PB][HS_PB + 1];

  if (t < HS_PB) {
    s_where[t] = c0 + t;
    s_what[t] = c0 + t;
  }
  __syncthreads();
  __shared__ int s_pl[HS_PB];
  const bool optimistic = (fuse & 4) != 0;
  if (!optimistic) {
    if (t < HS_PB) s_pl[t] = (t < w) ? nd.pivlist[t] : -1;
  } else if (t < 64) {
    // fuse & 4: OPTIMISTIC pivoting -- no tournament ran; partial pivoting among the block's own w rows decides the order
    // (one wave, lane i = row c0+i in registers).  panel_l21 checks the multipliers of the rows below against
    // HS_GROWTH_MAX; a violation, or a block that is singular on its own rows, raises nd.growth and the level is redone
    // with the tournament.  For the diagonally dominant fronts of elliptic problems the check never fires and the chain of
    // a panel step shrinks from (rounds + 2) dependent kernels to 2.
    const int i = t & 31;
    T x[HS_PB];
    bool alive = (t < w);
#pragma clang loop unroll(full)
    for (int j = 0; j < HS_PB; ++j) x[j] = (alive && j < w) ? nd.LF[(size_t)(c0 + i) + (size_t)(c0 + j) * nd.ldl] : Scal<T>::zero();
    bool bad = false;
#pragma clang loop unroll(full)
    for (int k = 0; k < HS_PB; ++k) {
      unsigned long long key = 0;
      if (alive && k < w) key = ((unsigned long long)__double_as_longlong(Scal<T>::abs1(x[k])) & ~0xffull) | (unsigned long long)(255 - t);
      const unsigned long long best = wave_max_u64(key);
      int win = -1;
      if (k < w && (best >> 8) != 0) win = 255 - (int)(best & 0xff);
      if (k < w && win < 0) bad = true;  // nothing left in this column on the block's rows
      if (t == 0) s_pl[k] = (k < w) ? (win >= 0 ? c0 + win : -1) : -1;
      if (win >= 0) {
        const T pk = lane_bcast(x[k], win);
        const T rp = Scal<T>::one() / pk;
        T l = Scal<T>::zero();
        if (alive && t != win) l = x[k] * rp;
        if (t == win) alive = false;
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j)
          if (j > k) x[j] = Scal<T>::fnma(l, lane_bcast(x[j], win), x[j]);
      }
    }
    if (bad && t == 0 && nd.growth) *nd.growth = 1;
  }
  __syncthreads();
  if (t == 0) {
    // winners (rows by their position at panel start, in elimination order) -> LAPACK-style swaps.
    // A row from below the top block sits where it started until it is picked; rows that started in
    // the top block are tracked through `where`; positions c0+j, j < k, are final.
    int first_bad = 0;
    for (int k = 0; k < w; ++k) {
      const int r = s_pl[k];
      const int target = c0 + k;
      int p = target;
      if (r < 0) {  // no pivot: leave the row, flag the front singular
        if (first_bad == 0) first_bad = c0 + k + 1;
      } else {
        p = (r >= c0 && r < c0 + w) ? s_where[r - c0] : r;
      }
      s_piv[k] = p;
      if (p != target) {
        const int q = s_what[k];  // always a row that started inside the top block
        s_what[k] = r;
        if (p >= c0 && p < c0 + w) s_what[p - c0] = q;
        if (r >= c0 && r < c0 + w) s_where[r - c0] = target;
        s_where[q - c0] = p;
      }
    }
    if (first_bad && !optimistic) {  // optimistic: the growth flag is already up, the redo decides about singularity
      int old = *nd.info;
      if (old == 0 || old > first_bad) *nd.info = first_bad;
    }
  }
  __syncthreads();
  if (t < w) nd.ipiv[c0 + t] = s_piv[t];
  // Apply the w swaps to the panel's own columns (thread j owns column c0+j) and, as one more
  // "column", to the accumulated row permutation -- as ONE gather/scatter of the <= 2w touched rows
  // (top position k receives the row s_what[k]; the row that started at top position i goes to
  // s_where[i]), so all loads are independent instead of 2w dependent global round trips.
  auto swap_column = [&](T* col) {
    T topv[HS_PB], pivv[HS_PB];
#pragma unroll
    for (int i = 0; i < HS_PB; ++i) {
      topv[i] = (i < w) ? col[c0 + i] : Scal<T>::zero();
      pivv[i] = (i < w) ? col[s_what[i]] : Scal<T>::zero();
    }
#pragma unroll
    for (int i = 0; i < HS_PB; ++i) {
      if (i < w) {
        col[c0 + i] = pivv[i];
        const int f = s_where[i];
        if (f < c0 || f >= c0 + w) col[f] = topv[i];
      }
    }
  };
  if (t < w) {
    swap_column(nd.LF + (size_t)(c0 + t) * nd.ldl);
  } else if (t >= 128 && t < 128 + w2) {
    swap_column(nd.LF + (size_t)(c0 + HS_PB + (t - 128)) * nd.ldl);
  } else if ((fuse & 2) && t >= 192 && t < 192 + HS_PB && c0 >= HS_PB) {
    swap_column(nd.LF + (size_t)(c0 - HS_PB + (t - 192)) * nd.ldl);
  } else if (t >= 64 && t < 64 + w) {
    const int i = t - 64;
    const int topv = nd.rperm[c0 + i], pivv = nd.rperm[s_what[i]];
    __builtin_amdgcn_s_waitcnt(0);  // both loads of every lane are complete before any lane stores (one wave)
    nd.rperm[c0 + i] = pivv;
    const int f = s_where[i];
    if (f < c0 || f >= c0 + w) nd.rperm[f] = topv;
  }
  __syncthreads();
  // load the top w x w block (identity-padded to 32); il = iu = I
  for (int e = t; e < HS_PB * HS_PB; e += 256) {
    int i = e & 31, j = e >> 5;
    T id = (i == j) ? Scal<T>::one() : Scal<T>::zero();
    T v = id;
    if (i < w && j < w) v = nd.LF[(size_t)(c0 + i) + (size_t)(c0 + j) * nd.ldl];
    s_a[i][j] = v;
    s_il[i][j] = id;
    s_iu[i][j] = id;
  }
  __syncthreads();
  // unpivoted LU (pivot order fixed by the tournament); the same row operations applied to I give inv(L).
  // ONE wave does the 32 dependent steps with the rows in registers (lane i = row i; the pivot row comes
  // from lane k through v_readlane, k a compile-time constant): no barriers, no LDS traffic inside the
  // chain -- the 128 workgroup barriers of the previous LDS formulation were 2/3 of this kernel's time.
  if constexpr (sizeof(T) == 16) {
    // ComplexF64: the register formulation needs 4x the instructions (270 KB of straight-line code, slower than
    // the barrier chain it replaces) -- keep the LDS formulation: 256 threads, two barriers per step
    for (int k = 0; k < HS_PB; ++k) {
      const T piv = s_a[k][k];
      const bool zero_piv = (Scal<T>::abs1(piv) == 0.0);
      if (t < HS_PB && t > k && !zero_piv) s_a[t][k] = s_a[t][k] / piv;
      if (zero_piv && t == 0 && k < w) {
        if (optimistic) {
          if (nd.growth) *nd.growth = 1;
        } else {
          int old = *nd.info;
          if (old == 0 || old > c0 + k + 1) *nd.info = c0 + k + 1;
        }
      }
      __syncthreads();
      if (!zero_piv) {
        for (int e = t; e < HS_PB * HS_PB; e += 256) {
          int i = e & 31, j = e >> 5;
          if (i > k) {
            if (j > k)
              s_a[i][j] = Scal<T>::fnma(s_a[i][k], s_a[k][j], s_a[i][j]);
            else
              s_il[i][j] = Scal<T>::fnma(s_a[i][k], s_il[k][j], s_il[i][j]);
          }
        }
      }
      __syncthreads();
    }
    for (int p = HS_PB - 1; p >= 0; --p) {
      T d = s_a[p][p];
      if (Scal<T>::abs1(d) == 0.0) d = Scal<T>::one();
      if (t < HS_PB && t >= p) s_iu[p][t] = s_iu[p][t] / d;
      __syncthreads();
      for (int e = t; e < HS_PB * HS_PB; e += 256) {
        int i = e & 31, j = e >> 5;
        if (i < p && j >= p) s_iu[i][j] = Scal<T>::fnma(s_a[i][p], s_iu[p][j], s_iu[i][j]);
      }
      __syncthreads();
    }
  } else
  if (t < 64) {
    const int i = t & 31;
    T ar[HS_PB], il[HS_PB];
#pragma clang loop unroll(full)
    for (int j = 0; j < HS_PB; ++j) {
      ar[j] = s_a[i][j];
      il[j] = (i == j) ? Scal<T>::one() : Scal<T>::zero();
    }
    int bad = 0;
#pragma clang loop unroll(full)
    for (int k = 0; k < HS_PB; ++k) {
      const T piv = lane_bcast(ar[k], k);
      if (Scal<T>::abs1(piv) == 0.0) {  // wave-uniform
        if (bad == 0 && k < w) bad = c0 + k + 1;
      } else {
        const T rp = Scal<T>::one() / piv;
        T l = Scal<T>::zero();
        if (i > k) {
          l = ar[k] * rp;
          ar[k] = l;
        }
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) {
          if (j > k)
            ar[j] = Scal<T>::fnma(l, lane_bcast(ar[j], k), ar[j]);
          else
            il[j] = Scal<T>::fnma(l, lane_bcast(il[j], k), il[j]);
        }
      }
    }
    if (bad && t == 0) {
      if (optimistic) {
        if (nd.growth) *nd.growth = 1;
      } else {
        int old = *nd.info;
        if (old == 0 || old > bad) *nd.info = bad;
      }
    }
    if (t < HS_PB) {
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) {
        s_a[i][j] = ar[j];
        s_il[i][j] = il[j];
      }
    }
    // inv(U) by back substitution in rank-1 form (zero pivots are treated as 1: the front is already flagged):
    // lane i owns row i of inv(U) and of U; row p comes from lane p
#pragma clang loop unroll(full)
    for (int j = 0; j < HS_PB; ++j) il[j] = (i == j) ? Scal<T>::one() : Scal<T>::zero();
#pragma clang loop unroll(full)
    for (int p = HS_PB - 1; p >= 0; --p) {
      T d = lane_bcast(ar[p], p);
      if (Scal<T>::abs1(d) == 0.0) d = Scal<T>::one();
      const T rd = Scal<T>::one() / d;
      const T u = (i < p) ? ar[p] : Scal<T>::zero();
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) {
        if (j >= p) {
          if (i == p) il[j] = il[j] * rd;
          const T rowp = lane_bcast(il[j], p);
          il[j] = Scal<T>::fnma(u, rowp, il[j]);
        }
      }
    }
    if (t < HS_PB) {
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) s_iu[i][j] = il[j];
    }
  }
  __syncthreads();
  for (int e = t; e < HS_PB * HS_PB; e += 256) {
    int i = e & 31, j = e >> 5;
    if (i < w && j < w) nd.LF[(size_t)(c0 + i) + (size_t)(c0 + j) * nd.ldl] = s_a[i][j];
    nd.invL[(size_t)pb * HS_PB * HS_PB + e] = s_il[i][j];
    nd.invU[(size_t)pb * HS_PB * HS_PB + e] = s_iu[i][j];
  }
  if (w2 > 0) {  // U12 = inv(L11) * (P*A)[c0:c0+w, next 32 columns], in place
    __syncthreads();
    for (int e = t; e < HS_PB * HS_PB; e += 256) {
      int i = e & 31, j = e >> 5;
      T v = Scal<T>::zero();
      if (i < w && j < w2) v = nd.LF[(size_t)(c0 + i) + (size_t)(c0 + HS_PB + j) * nd.ldl];
      s_a[i][j] = v;
    }
    __syncthreads();
    for (int e = t; e < HS_PB * HS_PB; e += 256) {
      int i = e & 31, j = e >> 5;
      if (i < w && j < w2) {
        T u = Scal<T>::zero();
        for (int q = 0; q <= i; ++q) u = Scal<T>::fma(s_il[i][q], s_a[q][j], u);
        nd.LF[(size_t)(c0 + i) + (size_t)(c0 + HS_PB + j) * nd.ldl] = u;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// panel_l21: rows below the diagonal block, one row per thread:  x <- x * inv(U11)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void panel_l21_kernel(const NodeDesc<T>* __restrict__ nodes, int pb, int fuse, int rpt, int rlim) {
  __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win instruction issue over co-resident GEMM waves
  const NodeDesc<T> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  const int r0 = c0 + w;
  // rpt rows per thread, one after the other (rows t, t + 256, ... of the workgroup's 256 * rpt; HS_L21_ROWS, default 1): fewer, longer
  // workgroups for a lone front next to a running GEMM were tried and measured no gain (launch_panel_l21)
  const int mrows = min(nd.m, rlim);  // rlim: only the rows of the 256-wide diagonal block (Sched::lu_rec, the rows below follow as one product)
  if ((int)blockIdx.x * 256 * rpt >= mrows - r0) return;
  // fuse & 1: first panel of a 64-column pair -- the thread that owns a row also applies the rank-32 update
  // A[row, next 32 columns] -= L21[row, :] * U12 (U12 left in place by panel_pivot), with L21[row, :] still in registers
  const int w2 = (fuse & 1) ? max(0, min(HS_PB, nd.ni - (c0 + HS_PB))) : 0;
  __shared__ T s_iu[HS_PB * HS_PB];  // column-major, ld 32
  __shared__ T s_u[HS_PB * HS_PB];   // U12, column-major, ld 32
  for (int e = threadIdx.x; e < HS_PB * HS_PB; e += 256) {
    s_iu[e] = nd.invU[(size_t)pb * HS_PB * HS_PB + e];
    if (w2 > 0) {
      const int i = e & 31, j = e >> 5;
      s_u[e] = (i < w && j < w2) ? nd.LF[(size_t)(c0 + i) + (size_t)(c0 + HS_PB + j) * nd.ldl] : Scal<T>::zero();
    }
  }
  __syncthreads();
  for (int rr = 0; rr < rpt; ++rr) {
  const int row = r0 + (blockIdx.x * rpt + rr) * 256 + threadIdx.x;
  if (row >= mrows) return;
  T* base = nd.LF + (size_t)row + (size_t)c0 * nd.ldl;
  T a[HS_PB];
#pragma unroll
  for (int j = 0; j < HS_PB; ++j) a[j] = (j < w) ? base[(size_t)j * nd.ldl] : Scal<T>::zero();
  // x <- x * inv(U11) IN PLACE, last column first: l[j] needs a[0..j] only, so a[j] can take it.  One register row instead of two: the ComplexF64
  // instance held 310 registers (256 + 54 AGPRs) -- more than ONE retiring GEMM workgroup frees on a CU (4 waves x 256), so next to a running
  // trailing update its workgroups waited for BOTH GEMM workgroups of some CU to retire (340 us per launch in the profile of the metric's workload)
  T(&l)[HS_PB] = a;
  double lmax = 0.0;
#pragma unroll
  for (int j = HS_PB - 1; j >= 0; --j) {
    T s = Scal<T>::zero();
#pragma unroll
    for (int i = 0; i < HS_PB; ++i)
      if (i <= j) s = Scal<T>::fma(a[i], s_iu[i + j * HS_PB], s);
    if (!(Scal<T>::abs1(s) <= HS_GROWTH_MAX)) lmax = 2.0 * HS_GROWTH_MAX;  // not fmax(): it drops NaN operands, and a NaN multiplier is a violation
    if (j < w) base[(size_t)j * nd.ldl] = s;
    a[j] = s;
  }
  // fuse & 4: optimistic pivoting -- a row that partial pivoting could have picked (row < pivrows) must not need a
  // multiplier beyond HS_GROWTH_MAX; otherwise the level is redone with tournament pivoting (NaN counts as a violation)
  if ((fuse & 4) && row < nd.pivrows && !(lmax <= HS_GROWTH_MAX) && nd.growth) *nd.growth = 1;
  if (w2 > 0) {
    T* nxt = base + (size_t)HS_PB * nd.ldl;
#pragma unroll 4
    for (int j = 0; j < HS_PB; ++j) {
      if (j < w2) {
        T v = nxt[(size_t)j * nd.ldl];
#pragma unroll
        for (int q = 0; q < HS_PB; ++q) v = Scal<T>::fnma(l[q], s_u[q + j * HS_PB], v);
        nxt[(size_t)j * nd.ldl] = v;
      }
    }
  }
  }
}

// ------------------------------------------------------------------------------------------------
// panel_pivot, OPTIMISTIC pivoting, Float64 -- the chain link of a panel step, cut to one elimination.
// The general kernel above searches the pivots with one register elimination, turns the winners into swaps, applies them in
// global memory, re-reads the swapped block and eliminates it AGAIN for L, U and both inverses: four dependent 32-step loops
// and seven dependent memory round trips in ONE wave (59 us alone on the device, 140 us next to a running GEMM -- 40 % of
// the panel chain a lone front waits for, tools/factor_trace.sh).  With optimistic pivoting every pivot comes from the block's
// own 32 rows, so nothing outside the 32 x 32 block moves:
//   wave 0   ONE elimination with the pivot search inside it (lane = row, implicit permutation: a row that wins step k keeps its
//            lane and takes position k); it leaves L\U in LDS in pivoted order;
//   then, concurrently,  wave 1: inv(L)   wave 2: inv(U)
//            wave 0: swaps for `laswp` (ipiv), the pivoted rows of the previous 32 columns (fuse & 2), rperm, the L\U block;
//   and U12 = inv(L) * (P*A12) by all threads once inv(L) is in LDS (fuse & 1);
//   the side blocks were loaded into LDS by waves 1-3 WHILE wave 0 eliminated.  Two barriers, two memory round trips on the chain.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void panel_pivot_opt_kernel(const NodeDesc<double>* __restrict__ nodes, int pb, int fuse) {
  __builtin_amdgcn_s_setprio(3);
  const NodeDesc<double> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int w2 = (fuse & 1) ? max(0, min(HS_PB, nd.ni - (c0 + HS_PB))) : 0;
  const bool prev = (fuse & 2) && c0 >= HS_PB;
  const size_t ld = nd.ldl;
  double* const LF = nd.LF;

  __shared__ int s_pl[HS_PB];     // row (position in the front at panel start) that ends at top position c0+k
  __shared__ int s_piv[HS_PB], s_where[HS_PB], s_what[HS_PB], s_r[HS_PB];
  __shared__ double s_a[HS_PB][HS_PB + 1];   // L\U in pivoted order
  __shared__ double s_n[HS_PB][HS_PB + 1];   // rows c0.. of the NEXT 32 columns as loaded (fuse & 1)
  __shared__ double s_p[HS_PB][HS_PB + 1];   // rows c0.. of the PREVIOUS 32 columns as loaded (fuse & 2)
  __shared__ double s_il[HS_PB][HS_PB + 1];  // inv(L), for U12

  if (wave == 0) {
    const int i = lane & 31;
    bool alive = lane < w;
    int pos = (lane >= w && lane < HS_PB) ? lane : -1;  // padding rows keep their place
    double x[HS_PB];
#pragma clang loop unroll(full)
    for (int j = 0; j < HS_PB; ++j) x[j] = (lane < w && j < w) ? gld(LF + (size_t)(c0 + i) + (size_t)(c0 + j) * ld) : ((lane < HS_PB && i == j) ? 1.0 : 0.0);
    bool bad = false;
#pragma clang loop unroll(full)
    for (int k = 0; k < HS_PB; ++k) {
      if (k < w) {  // wave-uniform
        unsigned long long key = 0;
        if (alive) key = ((unsigned long long)__double_as_longlong(fabs(x[k])) & ~0xffull) | (unsigned long long)(255 - lane);
        const unsigned long long best = wave_max_u64(key);
        int win;
        double rp;
        if ((best >> 8) != 0) {
          win = 255 - (int)(best & 0xff);
          rp = 1.0 / lane_bcast(x[k], win);
        } else {  // the column is zero on every row still in play: the block is singular on its own rows -> the level is redone
          bad = true;
          const unsigned long long m = __ballot(alive);
          win = m ? (__ffsll((long long)m) - 1) : 0;
          rp = 0.0;
        }
        win = __builtin_amdgcn_readfirstlane(win);
        if (lane == 0) s_pl[k] = c0 + win;
        double l = 0.0;
        if (alive && lane != win) {
          l = x[k] * rp;
          x[k] = l;
        }
        if (lane == win) {
          alive = false;
          pos = k;
        }
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j)
          if (j > k) x[j] = fma(-l, lane_bcast(x[j], win), x[j]);
      } else if (lane == 0) {
        s_pl[k] = c0 + k;
      }
    }
    if (bad && lane == 0 && nd.growth) *nd.growth = 1;
    if (pos >= 0) {
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) s_a[pos][j] = x[j];
    }
  } else if (wave == 1) {
    if (w2 > 0 && lane < HS_PB) {  // lane = column of the next block
#pragma clang loop unroll(full)
      for (int i = 0; i < HS_PB; ++i) s_n[i][lane] = (i < w && lane < w2) ? gld(LF + (size_t)(c0 + i) + (size_t)(c0 + HS_PB + lane) * ld) : 0.0;
    }
  } else if (wave == 2) {
    if (prev && lane < HS_PB) {  // lane = column of the previous block
#pragma clang loop unroll(full)
      for (int i = 0; i < HS_PB; ++i) s_p[i][lane] = (i < w) ? gld(LF + (size_t)(c0 + i) + (size_t)(c0 - HS_PB + lane) * ld) : 0.0;
    }
  } else {
    if (lane < HS_PB) s_r[lane] = (lane < w) ? nd.rperm[c0 + lane] : 0;
  }
  __syncthreads();

  if (wave == 0) {
    if (lane == 0) {
      // winners (rows by their position at panel start, in elimination order) -> LAPACK-style swaps for laswp on the other columns
      for (int k = 0; k < HS_PB; ++k) {
        s_where[k] = c0 + k;
        s_what[k] = c0 + k;
      }
      for (int k = 0; k < w; ++k) {
        const int r = s_pl[k];
        const int target = c0 + k;
        const int p = s_where[r - c0];
        s_piv[k] = p;
        if (p != target) {
          const int q = s_what[k];
          s_what[k] = r;
          s_what[p - c0] = q;
          s_where[r - c0] = target;
          s_where[q - c0] = p;
        }
      }
    }
    // (same wave: the LDS writes of lane 0 are visible to the wave's later reads after the waitcnt the compiler places)
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
    if (lane < w) {
      nd.ipiv[c0 + lane] = s_piv[lane];
      nd.rperm[c0 + lane] = s_r[s_pl[lane] - c0];
    }
    // the L\U block: lane -> row, two columns per pass
    {
      const int i = lane & 31, jh = lane >> 5;
#pragma clang loop unroll(full)
      for (int jj = 0; jj < HS_PB / 2; ++jj) {
        const int j = 2 * jj + jh;
        if (i < w && j < w) gst(LF + (size_t)(c0 + i) + (size_t)(c0 + j) * ld, s_a[i][j]);
      }
      if (prev) {  // pivoted rows of the previous 32 columns (the left-looking swap of the pair)
#pragma clang loop unroll(full)
        for (int jj = 0; jj < HS_PB / 2; ++jj) {
          const int j = 2 * jj + jh;
          if (i < w) gst(LF + (size_t)(c0 + i) + (size_t)(c0 - HS_PB + j) * ld, s_p[s_pl[i] - c0][j]);
        }
      }
    }
  } else if (wave == 1) {
    // inv(L): lane i owns row i of L and of inv(L); the same row operations that reduce L to I applied to I
    if (lane < HS_PB) {
      const int i = lane;
      double lr[HS_PB], il[HS_PB];
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) {
        lr[j] = s_a[i][j];
        il[j] = (i == j) ? 1.0 : 0.0;
      }
#pragma clang loop unroll(full)
      for (int k = 0; k < HS_PB - 1; ++k) {
        const double l = (i > k) ? lr[k] : 0.0;
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j)
          if (j <= k) il[j] = fma(-l, lane_bcast(il[j], k), il[j]);
      }
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) {
        gst(nd.invL + (size_t)pb * HS_PB * HS_PB + i + j * HS_PB, il[j]);
        s_il[i][j] = il[j];
      }
    }
  } else if (wave == 2) {
    // inv(U) by back substitution in rank-1 form (zero pivots are treated as 1: the front is already flagged)
    if (lane < HS_PB) {
      const int i = lane;
      double ar[HS_PB], iu[HS_PB];
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) {
        ar[j] = s_a[i][j];
        iu[j] = (i == j) ? 1.0 : 0.0;
      }
#pragma clang loop unroll(full)
      for (int p = HS_PB - 1; p >= 0; --p) {
        double d = lane_bcast(ar[p], p);
        if (d == 0.0) d = 1.0;
        const double rd = 1.0 / d;
        const double u = (i < p) ? ar[p] : 0.0;
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) {
          if (j >= p) {
            if (i == p) iu[j] = iu[j] * rd;
            const double rowp = lane_bcast(iu[j], p);
            iu[j] = fma(-u, rowp, iu[j]);
          }
        }
      }
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) gst(nd.invU + (size_t)pb * HS_PB * HS_PB + i + j * HS_PB, iu[j]);
    }
  }
  if (w2 > 0) {  // U12 = inv(L11) * (P*A12) (fuse & 1), every thread four entries; inv(L) reaches LDS through wave 1
    __syncthreads();
    for (int e = t; e < HS_PB * HS_PB; e += 256) {
      const int i = e & 31, j = e >> 5;
      if (i < w && j < w2) {
        double u = 0.0;
        for (int q = 0; q <= i; ++q) u = fma(s_il[i][q], s_n[s_pl[q] - c0][j], u);
        gst(LF + (size_t)(c0 + i) + (size_t)(c0 + HS_PB + j) * ld, u);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// panel_pivot, OPTIMISTIC pivoting, ComplexF64 -- the merged kernel of the complex path (round 3).
// The general kernel (above) cost 190 us alone and 260-490 us next to a GEMM per 32-column panel -- the longest link of the complex panel
// chain, 0.5 s of device time per factorization of Helmholtz 112^3 (profiles/r03_*_before_zchain.csv) -- because its two 32-step loops run on 256
// threads with two workgroup barriers per step, after a separate pivot-search elimination and a trip of the swaps through global memory.
// panel_pivot_opt_kernel's register formulation is 4x the straight-line code for complex numbers (270 KB, slower than what it replaces), so
// here the rows live in LDS and ONE wave walks each dependent loop -- LDS operations of a wave complete in order, no barrier inside a loop --
// with the two halves of the wave sharing a row's columns:
//   wave 0   one elimination with the pivot search inside it (lane & 31 = row, implicit permutation); meanwhile waves 1-3 load the side blocks
//   then     wave 1: inv(L)    wave 2: inv(U)    wave 0: ipiv, rperm, the L\U block, the pivoted rows of the previous 32 columns
//   and      U12 = inv(L) * (P A12) by all threads (fuse & 1).     Two workgroup barriers per panel.
// ------------------------------------------------------------------------------------------------
#define HS_ZP_LD (HS_PB + 1)
#define HS_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")  // a wave's LDS writes have landed before its next (cross-lane) reads
__global__ __launch_bounds__(256) void panel_pivot_opt_z_kernel(const NodeDesc<cplx>* __restrict__ nodes, int pb, int fuse) {
  __builtin_amdgcn_s_setprio(3);
  const NodeDesc<cplx> nd = nodes[blockIdx.y];
  const int c0 = pb * HS_PB;
  if (c0 >= nd.ni) return;
  const int w = min(HS_PB, nd.ni - c0);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int w2 = (fuse & 1) ? max(0, min(HS_PB, nd.ni - (c0 + HS_PB))) : 0;
  const bool prev = (fuse & 2) && c0 >= HS_PB;
  const size_t ld = nd.ldl;
  cplx* const LF = nd.LF;
  extern __shared__ __attribute__((aligned(16))) unsigned char zp_smem[];
  cplx* s_a = reinterpret_cast<cplx*>(zp_smem);  // the block, rows in ORIGINAL order; after the elimination L\U, row of pivot k = s_pl[k] - c0
  cplx* s_n = s_a + HS_PB * HS_ZP_LD;            // rows c0.. of the NEXT 32 columns as loaded (fuse & 1)
  cplx* s_p = s_n + HS_PB * HS_ZP_LD;            // rows c0.. of the PREVIOUS 32 columns as loaded (fuse & 2)
  cplx* s_il = s_p + HS_PB * HS_ZP_LD;           // inv(L), pivoted order
  cplx* s_iu = s_il + HS_PB * HS_ZP_LD;          // inv(U)
  int* s_pl = reinterpret_cast<int*>(s_iu + HS_PB * HS_ZP_LD);
  int* s_piv = s_pl + HS_PB;
  int* s_where = s_piv + HS_PB;
  int* s_what = s_where + HS_PB;
  int* s_r = s_what + HS_PB;
#define ZA(i, j) s_a[(i) * HS_ZP_LD + (j)]
  const int i = lane & 31, hf = lane >> 5;

  // ---- load: wave 0 the diagonal block (identity-padded), waves 1-3 the side blocks and rperm -------------------------------------------------
  if (wave == 0) {
    for (int j = hf; j < HS_PB; j += 2) ZA(i, j) = (i < w && j < w) ? gld(LF + (size_t)(c0 + i) + (size_t)(c0 + j) * ld) : (i == j ? Scal<cplx>::one() : Scal<cplx>::zero());
  } else if (wave == 1) {
    if (w2 > 0)
      for (int r = hf; r < HS_PB; r += 2) s_n[r * HS_ZP_LD + i] = (r < w && i < w2) ? gld(LF + (size_t)(c0 + r) + (size_t)(c0 + HS_PB + i) * ld) : Scal<cplx>::zero();
  } else if (wave == 2) {
    if (prev)
      for (int r = hf; r < HS_PB; r += 2) s_p[r * HS_ZP_LD + i] = (r < w) ? gld(LF + (size_t)(c0 + r) + (size_t)(c0 - HS_PB + i) * ld) : Scal<cplx>::zero();
  } else {
    if (lane < HS_PB) s_r[lane] = (lane < w) ? nd.rperm[c0 + lane] : 0;
  }
  // ---- 1. elimination with the pivot search inside (wave 0; both halves of the wave work on row i, columns of their parity) -----------------
  if (wave == 0) {
    HS_LDS_ORDER();
    bool alive = i < w;
    bool bad = false;
    for (int k = 0; k < HS_PB; ++k) {
      if (k < w) {
        unsigned long long key = 0;
        if (alive && hf == 0) key = ((unsigned long long)__double_as_longlong(Scal<cplx>::abs1(ZA(i, k))) & ~0xffull) | (unsigned long long)(255 - i);
        const unsigned long long best = wave_max_u64(key);
        int win;
        if ((best >> 8) != 0) {
          win = 255 - (int)(best & 0xff);
        } else {  // the column is zero on every row still in play: the block is singular on its own rows -> the level is redone
          bad = true;
          const unsigned long long m = __ballot(alive && hf == 0);
          win = m ? (__ffsll((long long)m) - 1) : 0;
        }
        win = __builtin_amdgcn_readfirstlane(win);
        if (lane == 0) s_pl[k] = c0 + win;
        const cplx piv = ZA(win, k);
        const bool zero_piv = (best >> 8) == 0;
        if (alive && i != win && !zero_piv) {
          const cplx l = ZA(i, k) / piv;
          for (int j = k + 1 + hf; j < HS_PB; j += 2) ZA(i, j) = Scal<cplx>::fnma(l, ZA(win, j), ZA(i, j));
          HS_LDS_ORDER();
          if (hf == 0) ZA(i, k) = l;  // (after both halves have read the old value)
        }
        if (i == win) alive = false;
        HS_LDS_ORDER();
      } else if (lane == 0) {
        s_pl[k] = c0 + k;
      }
    }
    if (bad && lane == 0 && nd.growth) *nd.growth = 1;
  }
  __syncthreads();
#define ZLU(k, j) s_a[(s_pl[k] - c0) * HS_ZP_LD + (j)]  // L\U in pivoted order
  // ---- 2. inverses (waves 1, 2); swaps / rperm / the L\U block / the previous columns (wave 0) ------------------------------------------------
  if (wave == 0) {
    if (lane == 0) {
      for (int k = 0; k < HS_PB; ++k) {
        s_where[k] = c0 + k;
        s_what[k] = c0 + k;
      }
      for (int k = 0; k < w; ++k) {
        const int r = s_pl[k];
        const int target = c0 + k;
        const int p = s_where[r - c0];
        s_piv[k] = p;
        if (p != target) {
          const int q = s_what[k];
          s_what[k] = r;
          s_what[p - c0] = q;
          s_where[r - c0] = target;
          s_where[q - c0] = p;
        }
      }
    }
    HS_LDS_ORDER();
    if (lane < w) {
      nd.ipiv[c0 + lane] = s_piv[lane];
      nd.rperm[c0 + lane] = s_r[s_pl[lane] - c0];
    }
    for (int j = hf; j < HS_PB; j += 2) {
      if (i < w && j < w) gst(LF + (size_t)(c0 + i) + (size_t)(c0 + j) * ld, ZLU(i, j));
      if (prev && i < w) gst(LF + (size_t)(c0 + i) + (size_t)(c0 - HS_PB + j) * ld, s_p[(s_pl[i] - c0) * HS_ZP_LD + j]);
    }
  } else if (wave == 1) {
    // inv(L): the row operations that reduce L to I applied to I (row i of pivoted order in LDS; columns split over the two halves)
    for (int j = hf; j < HS_PB; j += 2) s_il[i * HS_ZP_LD + j] = (i == j) ? Scal<cplx>::one() : Scal<cplx>::zero();
    HS_LDS_ORDER();
    for (int k = 0; k < HS_PB - 1; ++k) {
      if (i > k) {
        const cplx l = ZLU(i, k);
        for (int j = hf; j <= k; j += 2) s_il[i * HS_ZP_LD + j] = Scal<cplx>::fnma(l, s_il[k * HS_ZP_LD + j], s_il[i * HS_ZP_LD + j]);
      }
      HS_LDS_ORDER();
    }
    for (int j = hf; j < HS_PB; j += 2) gst(nd.invL + (size_t)pb * HS_PB * HS_PB + i + j * HS_PB, s_il[i * HS_ZP_LD + j]);
  } else if (wave == 2) {
    // inv(U) by back substitution in rank-1 form (a zero pivot counts as 1: the front is already flagged)
    for (int j = hf; j < HS_PB; j += 2) s_iu[i * HS_ZP_LD + j] = (i == j) ? Scal<cplx>::one() : Scal<cplx>::zero();
    HS_LDS_ORDER();
    for (int p = HS_PB - 1; p >= 0; --p) {
      cplx d = ZLU(p, p);
      if (Scal<cplx>::abs1(d) == 0.0) d = Scal<cplx>::one();
      if (i == p)
        for (int j = p + hf; j < HS_PB; j += 2) s_iu[p * HS_ZP_LD + j] = s_iu[p * HS_ZP_LD + j] / d;
      HS_LDS_ORDER();
      if (i < p) {
        const cplx u = ZLU(i, p);
        for (int j = p + hf; j < HS_PB; j += 2) s_iu[i * HS_ZP_LD + j] = Scal<cplx>::fnma(u, s_iu[p * HS_ZP_LD + j], s_iu[i * HS_ZP_LD + j]);
      }
      HS_LDS_ORDER();
    }
    for (int j = hf; j < HS_PB; j += 2) gst(nd.invU + (size_t)pb * HS_PB * HS_PB + i + j * HS_PB, s_iu[i * HS_ZP_LD + j]);
  }
  if (w2 > 0) {  // U12 = inv(L11) * (P*A12) (fuse & 1), every thread four entries
    __syncthreads();
    for (int e = t; e < HS_PB * HS_PB; e += 256) {
      const int ii = e & 31, j = e >> 5;
      if (ii < w && j < w2) {
        cplx u = Scal<cplx>::zero();
        for (int q = 0; q <= ii; ++q) u = Scal<cplx>::fma(s_il[ii * HS_ZP_LD + q], s_n[(s_pl[q] - c0) * HS_ZP_LD + j], u);
        gst(LF + (size_t)(c0 + ii) + (size_t)(c0 + HS_PB + j) * ld, u);
      }
    }
  }
#undef ZA
#undef ZLU
}

// ------------------------------------------------------------------------------------------------
// group256: the WHOLE panel chain of a 256-column group on its 256 x 256 diagonal block in ONE launch (Float64, optimistic pivoting).
// Sched::lu_rec (diagonal-block-first groups) ran it as 28 dependent launches -- 8 x (pivot, L21), the swaps, the 64 / 128-row solves and the
// K <= 128 updates of the recursion, every one a single workgroup -- which next to a running GEMM cost 1.9 ms per group (tools/factor_trace.sh:
// each launch waits for a slot, runs 2x slower beside MFMA waves, and leaves the chip a launch gap): the 32,768 root of Poisson 128^3 spent
// 463 ms on 370 ms of GEMM.  Here one workgroup walks the eight 32-column panels of the block right-looking:
//   wave 0         LU of the 32 x 32 diagonal block with the pivot search inside (rows in registers, lane = row: panel_pivot_opt's elimination)
//   waves 1, 2     inv(L11), inv(U11) (registers -> LDS and the stored inverse blocks); wave 0 meanwhile: ipiv, rperm, the L\U block
//   all threads    one OTHER column of the block each: its 32 rows gathered in pivot order (the swaps), and right of the panel
//                  U12 = inv(L11) * (P A12), kept in LDS
//   all threads    one row BELOW the panel each: L21 = A21 * inv(U11) (growth bound checked), then the rank-32 update of that row over all
//                  trailing columns of the block from U12 in LDS
// Four barriers and ~20 us per panel, nothing but this workgroup's own L2-resident block touched; 77 KB of LDS (fits next to one GEMM
// workgroup).  The rows below the block follow as before: inv256 of the group, then L_below = A_below * inv(U_group) (GemmOp::ainv 7, 8).
// ------------------------------------------------------------------------------------------------
#define HS_G256_LDU 34  // U12 image: [column][k], 32 + 2 padding (16-byte reads of a k pair, rows of consecutive columns on different banks)
__global__ __launch_bounds__(256) void group256_kernel(const NodeDesc<double>* __restrict__ nodes, int grp) {
  __builtin_amdgcn_s_setprio(3);
  const NodeDesc<double> nd = nodes[blockIdx.y];
  const int g0 = grp * 256;
  if (g0 >= nd.ni) return;
  const int wl = min(256, nd.ni - g0);        // order of the diagonal block
  const int g1 = g0 + wl;
  const int rend = min(nd.m, g0 + 256);       // rows this kernel owns: the block's, and under a PARTIAL last group the boundary rows up to g0 + 256
                                              // (the products of the rows below start at g0 + 256, Sched::lu_rec)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const size_t ld = nd.ldl;
  double* const LF = nd.LF;
  extern __shared__ __attribute__((aligned(16))) double g256_smem[];
  double* s_u = g256_smem;                                   // [224][HS_G256_LDU]
  double (*s_a)[HS_PB + 1] = reinterpret_cast<double (*)[HS_PB + 1]>(s_u + 224 * HS_G256_LDU);  // L\U of the diagonal block, pivoted order
  double (*s_il)[HS_PB + 1] = s_a + HS_PB;
  double (*s_iu)[HS_PB + 1] = s_il + HS_PB;
  int* s_pl = reinterpret_cast<int*>(s_iu + HS_PB);          // row (front position) that ends at top position c+k
  int* s_piv = s_pl + HS_PB;
  int* s_where = s_piv + HS_PB;
  int* s_what = s_where + HS_PB;
  int* s_r = s_what + HS_PB;

  for (int c = g0; c < g1; c += HS_PB) {
    const int w = min(HS_PB, g1 - c);
    const int pb = c / HS_PB;
    // ---- 1. the diagonal block: one elimination with the pivot search inside (wave 0); rperm of its rows (wave 3) ---------------------------
    if (wave == 0) {
      const int i = lane & 31;
      bool alive = lane < w;
      int pos = (lane >= w && lane < HS_PB) ? lane : -1;  // padding rows keep their place
      double x[HS_PB];
#pragma clang loop unroll(full)
      for (int j = 0; j < HS_PB; ++j) x[j] = (lane < w && j < w) ? gld(LF + (size_t)(c + i) + (size_t)(c + j) * ld) : ((lane < HS_PB && i == j) ? 1.0 : 0.0);
      bool bad = false;
#pragma clang loop unroll(full)
      for (int k = 0; k < HS_PB; ++k) {
        if (k < w) {  // wave-uniform
          unsigned long long key = 0;
          if (alive) key = ((unsigned long long)__double_as_longlong(fabs(x[k])) & ~0xffull) | (unsigned long long)(255 - lane);
          const unsigned long long best = wave_max_u64(key);
          int win;
          double rp;
          if ((best >> 8) != 0) {
            win = 255 - (int)(best & 0xff);
            rp = 1.0 / lane_bcast(x[k], win);
          } else {  // the column is zero on every row still in play: singular on its own rows -> the level is redone with the tournament
            bad = true;
            const unsigned long long m = __ballot(alive);
            win = m ? (__ffsll((long long)m) - 1) : 0;
            rp = 0.0;
          }
          win = __builtin_amdgcn_readfirstlane(win);
          if (lane == 0) s_pl[k] = c + win;
          double l = 0.0;
          if (alive && lane != win) {
            l = x[k] * rp;
            x[k] = l;
          }
          if (lane == win) {
            alive = false;
            pos = k;
          }
#pragma clang loop unroll(full)
          for (int j = 0; j < HS_PB; ++j)
            if (j > k) x[j] = fma(-l, lane_bcast(x[j], win), x[j]);
        } else if (lane == 0) {
          s_pl[k] = c + k;
        }
      }
      if (bad && lane == 0 && nd.growth) *nd.growth = 1;
      if (pos >= 0) {
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) s_a[pos][j] = x[j];
      }
    } else if (wave == 3) {
      if (lane < HS_PB) s_r[lane] = (lane < w) ? nd.rperm[c + lane] : 0;
    }
    __syncthreads();
    // ---- 2. inverses of L11, U11 (waves 1, 2); swaps for laswp on the columns outside the group, rperm, the L\U block (wave 0) --------------
    if (wave == 0) {
      if (lane == 0) {
        for (int k = 0; k < HS_PB; ++k) {
          s_where[k] = c + k;
          s_what[k] = c + k;
        }
        for (int k = 0; k < w; ++k) {
          const int r = s_pl[k];
          const int target = c + k;
          const int p = s_where[r - c];
          s_piv[k] = p;
          if (p != target) {
            const int q = s_what[k];
            s_what[k] = r;
            s_what[p - c] = q;
            s_where[r - c] = target;
            s_where[q - c] = p;
          }
        }
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): lane 0's LDS writes are visible to the wave's later reads
      if (lane < w) {
        nd.ipiv[c + lane] = s_piv[lane];
        nd.rperm[c + lane] = s_r[s_pl[lane] - c];
      }
      const int i = lane & 31, jh = lane >> 5;
#pragma clang loop unroll(full)
      for (int jj = 0; jj < HS_PB / 2; ++jj) {
        const int j = 2 * jj + jh;
        if (i < w && j < w) gst(LF + (size_t)(c + i) + (size_t)(c + j) * ld, s_a[i][j]);
      }
    } else if (wave == 1) {
      if (lane < HS_PB) {  // inv(L): the row operations that reduce L to I applied to I; lane i owns row i
        const int i = lane;
        double lr[HS_PB], il[HS_PB];
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) {
          lr[j] = s_a[i][j];
          il[j] = (i == j) ? 1.0 : 0.0;
        }
#pragma clang loop unroll(full)
        for (int k = 0; k < HS_PB - 1; ++k) {
          const double l = (i > k) ? lr[k] : 0.0;
#pragma clang loop unroll(full)
          for (int j = 0; j < HS_PB; ++j)
            if (j <= k) il[j] = fma(-l, lane_bcast(il[j], k), il[j]);
        }
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) {
          gst(nd.invL + (size_t)pb * HS_PB * HS_PB + i + j * HS_PB, il[j]);
          s_il[i][j] = il[j];
        }
      }
    } else if (wave == 2) {
      if (lane < HS_PB) {  // inv(U) by back substitution in rank-1 form (a zero pivot counts as 1: the front is already flagged)
        const int i = lane;
        double ar[HS_PB], iu[HS_PB];
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) {
          ar[j] = s_a[i][j];
          iu[j] = (i == j) ? 1.0 : 0.0;
        }
#pragma clang loop unroll(full)
        for (int p = HS_PB - 1; p >= 0; --p) {
          double d = lane_bcast(ar[p], p);
          if (d == 0.0) d = 1.0;
          const double rd = 1.0 / d;
          const double u = (i < p) ? ar[p] : 0.0;
#pragma clang loop unroll(full)
          for (int j = 0; j < HS_PB; ++j) {
            if (j >= p) {
              if (i == p) iu[j] = iu[j] * rd;
              const double rowp = lane_bcast(iu[j], p);
              iu[j] = fma(-u, rowp, iu[j]);
            }
          }
        }
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) {
          gst(nd.invU + (size_t)pb * HS_PB * HS_PB + i + j * HS_PB, iu[j]);
          s_iu[i][j] = iu[j];
        }
      }
    }
    __syncthreads();
    // ---- 3. every other column of the block: its 32 rows in pivot order; right of the panel U12 = inv(L11) * (P A12), kept in LDS ---------------
    {
      const int nother = wl - w;  // columns of the block outside the panel
      if (t < nother) {
        const int col = (g0 + t < c) ? g0 + t : g0 + t + w;  // left of the panel, then right of it
        double* cp = LF + (size_t)col * ld;
        double xp[HS_PB];
#pragma clang loop unroll(full)
        for (int k = 0; k < HS_PB; ++k) xp[k] = (k < w) ? gld(cp + s_pl[k]) : 0.0;
        if (col < c) {
#pragma clang loop unroll(full)
          for (int k = 0; k < HS_PB; ++k)
            if (k < w) gst(cp + c + k, xp[k]);
        } else {
          const int j = col - (c + w);
          double* su = s_u + (size_t)j * HS_G256_LDU;
#pragma clang loop unroll(full)
          for (int i = 0; i < HS_PB; ++i) {
            double u = 0.0;
#pragma clang loop unroll(full)
            for (int q = 0; q < HS_PB; ++q)
              if (q <= i) u = fma(s_il[i][q], xp[q], u);
            if (i < w) gst(cp + c + i, u);
            su[i] = (i < w) ? u : 0.0;
          }
        }
      }
    }
    __syncthreads();
    // ---- 4. every row below the panel: L21 = A21 * inv(U11), then its rank-32 update over the trailing columns of the block -------------------
    {
      const int r = c + w + t;
      if (r < rend) {
        double* rp_ = LF + (size_t)r;
        double a[HS_PB], l[HS_PB];
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) a[j] = (j < w) ? gld(rp_ + (size_t)(c + j) * ld) : 0.0;
        bool big = false;
#pragma clang loop unroll(full)
        for (int j = 0; j < HS_PB; ++j) {
          double sacc = 0.0;
#pragma clang loop unroll(full)
          for (int i = 0; i < HS_PB; ++i)
            if (i <= j) sacc = fma(a[i], s_iu[i][j], sacc);
          l[j] = sacc;
          if (!(fabs(sacc) <= HS_GROWTH_MAX)) big = true;  // NaN counts
          if (j < w) gst(rp_ + (size_t)(c + j) * ld, sacc);
        }
        if (big && r < nd.pivrows && nd.growth) *nd.growth = 1;
        const int ntr = g1 - (c + w);
        double* tp = rp_ + (size_t)(c + w) * ld;
        // eight columns per pass, the next eight loaded while these are updated: one global round trip per 256 FMAs, eight independent FMA
        // chains (a lone wave per SIMD hides neither the ~1 us of an L2 load nor the FMA latency by itself)
        constexpr int CH = 8;
        double v[CH], vn[CH];
#pragma clang loop unroll(full)
        for (int q = 0; q < CH; ++q) v[q] = gld(tp + (size_t)min(q, max(ntr - 1, 0)) * ld);  // (branch-free: past the end the last column is re-read, never stored)
        for (int j = 0; j < ntr; j += CH) {
#pragma clang loop unroll(full)
          for (int q = 0; q < CH; ++q) vn[q] = gld(tp + (size_t)min(j + CH + q, ntr - 1) * ld);
          const double* ub = s_u + (size_t)j * HS_G256_LDU;  // (columns past ntr read LDS that belongs to this workgroup and are never stored)
#pragma clang loop unroll(full)
          for (int k = 0; k < HS_PB; k += 2) {
#pragma clang loop unroll(full)
            for (int q = 0; q < CH; ++q) {
              const hs_d2u uu = *reinterpret_cast<const hs_d2u*>(ub + q * HS_G256_LDU + k);
              v[q] = fma(-l[k], uu.x, v[q]);
              v[q] = fma(-l[k + 1], uu.y, v[q]);
            }
          }
#pragma clang loop unroll(full)
          for (int q = 0; q < CH; ++q) {
            if (j + q < ntr) gst(tp + (size_t)(j + q) * ld, v[q]);
            v[q] = vn[q];
          }
        }
      }
    }
    __syncthreads();  // the next diagonal block and s_u are final / free
  }
}

template <class T>
bool launch_group256(const NodeDesc<T>* dnodes, int nbatch, int grp, hipStream_t s) {
  if constexpr (sizeof(T) != 8) {
    return false;  // ComplexF64 keeps the launch chain (the register eliminations are 4x the code)
  } else {
    // OFF by default (HS_GROUP_FUSED=1 turns it on): measured on MI355X it is correct (all parity tests) but no faster than the launch chain it
    // replaces -- 489 us per group ALONE on the device (rocprofv3, 2-D fronts of 258), because the eight 32 x 32 eliminations are serial single-wave
    // work of ~30 us each (v_readlane broadcasts), and 3.53 s instead of 3.35 s on Poisson 128^3: with 256 VGPRs + 66 AGPRs per lane its four waves
    // need BOTH GEMM workgroups of a CU to retire before they can be placed, where the small chain kernels slip in after one.
    static const bool on = getenv("HS_GROUP_FUSED") && getenv("HS_GROUP_FUSED")[0] == '1';
    if (!on || nbatch <= 0) return false;
    constexpr int lds = (224 * HS_G256_LDU + 3 * HS_PB * (HS_PB + 1)) * (int)sizeof(double) + 5 * HS_PB * (int)sizeof(int);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)group256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr = true;
    }
    hipLaunchKernelGGL(group256_kernel, dim3(1, nbatch), dim3(256), lds, s, dnodes, grp);
    return true;
  }
}

// ------------------------------------------------------------------------------------------------
// laswp: apply swaps ipiv[k0:k1) to columns [c0, c1) of LF or UR (one column per thread)
// ------------------------------------------------------------------------------------------------
template <class T>
__global__ __launch_bounds__(256) void laswp_kernel(const NodeDesc<T>* __restrict__ nodes, int mat, int c0, int c1, int k0, int k1) {
  __builtin_amdgcn_s_setprio(3);  // latency-critical chain: win instruction issue over co-resident GEMM waves
  const NodeDesc<T> nd = nodes[blockIdx.y];
  T* p;
  int ld, rows, cols;
  mat_of(nodes + blockIdx.y, mat, p, ld, rows, cols);
  c1 = min(c1, cols);
  k1 = min(k1, nd.ni);
  if (c0 + (int)blockIdx.x * 256 >= c1 || k0 >= k1) return;  // workgroup-uniform
  const int t = threadIdx.x;
  const int c = c0 + blockIdx.x * 256 + t;
  T* col = p + (size_t)min(c, c1 - 1) * ld;
  // Most pivots of a diagonally dominant front stay where they are: the workgroup first compacts the real swaps of up to
  // 1024 pivots (in order) into LDS, then every thread applies only those to its column.  The previous version walked all
  // k1-k0 pivots per column even when none moved (74 ms of the 32,768 root front).
  __shared__ int s_k[1024], s_p[1024];
  __shared__ int s_cnt[256];
  for (int kc = k0; kc < k1; kc += 1024) {
    const int kb = kc + 4 * t;  // this thread's four consecutive pivots
    int pv[4], n = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + u;
      pv[u] = (k < k1) ? nd.ipiv[k] : k;
      n += (pv[u] != k);
    }
    s_cnt[t] = n;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {  // inclusive scan of the counts
      const int v = (t >= off) ? s_cnt[t - off] : 0;
      __syncthreads();
      s_cnt[t] += v;
      __syncthreads();
    }
    const int total = s_cnt[255];
    int pos = s_cnt[t] - n;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (pv[u] != kb + u) {
        s_k[pos] = kb + u;
        s_p[pos] = pv[u];
        ++pos;
      }
    }
    __syncthreads();
    if (c < c1) {
      for (int q = 0; q < total; ++q) {
        const int k = s_k[q], pq = s_p[q];
        const T tmp = col[k];
        col[k] = col[pq];
        col[pq] = tmp;
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// trsm_small: X[r0 : r0+32*NBLK, c0:c1) <- L[r0.., r0..]^-1 * X for the 64- and 128-row solves inside a 256-column group of lu_rec
// (the recursion of Sched::trsm_rec turned each of them into 3 / 7 dependent launches -- 32-row inverse products and K = 32 / 64
// sub-tile GEMMs of ~16 / ~27 us each, 13 per 256 columns of every panel chain).  One workgroup per 32 columns of X: the chunk of X, the
// chunk of X arrives in LDS in one round trip, the stored 32 x 32 inverses and the off-diagonal blocks of L one block row ahead of their
// use, then block forward substitution.
// ------------------------------------------------------------------------------------------------
template <class T, int NBLK>
__global__ __launch_bounds__(256) void trsm_small_kernel(const NodeDesc<T>* __restrict__ nodes, int mat, int r0, int c0, int c1) {
  __builtin_amdgcn_s_setprio(3);  // latency-critical chain
  extern __shared__ __attribute__((aligned(16))) unsigned char trsm_small_raw[];
  constexpr int BB = HS_PB * HS_PB;
  // LDS budget: a chain kernel only ever finds room next to ONE resident GEMM workgroup (68 KB of the CU's 160), so the blocks of L
  // and the inverses are staged one block row at a time from registers that were loaded a step ahead (72 KB for 128 rows of Float64)
  T* Xs = reinterpret_cast<T*>(trsm_small_raw);  // NBLK blocks (column-major, ld 32): block row i of the chunk; becomes Y_i
  T* Is = Xs + NBLK * BB;                        // inverse diagonal block of the current block row
  T* Ls = Is + BB;                               // L_ik, k < i, of the current block row
  T* Ss = Ls + (NBLK - 1) * BB;                  // its right-hand side
  const NodeDesc<T> nd = nodes[blockIdx.y];
  if (r0 >= nd.ni) return;
  T* xp;
  int ldx, xrows, xcols;
  mat_of(nodes + blockIdx.y, mat, xp, ldx, xrows, xcols);
  c1 = min(c1, xcols);
  const int cc = c0 + blockIdx.x * HS_PB;
  if (cc >= c1) return;  // workgroup-uniform
  const int wc = min(HS_PB, c1 - cc);
  const int wr = min(NBLK * HS_PB, nd.ni - r0);
  const int nbv = (wr + HS_PB - 1) / HS_PB;
  const int t = threadIdx.x;
  const T* LF = nd.LF;
  const size_t ldl = nd.ldl;
  T pI[4], pL[NBLK - 1][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = t + 256 * u, a = e & 31, bcol = e >> 5;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
      const int row = i * HS_PB + a;
      Xs[i * BB + e] = (row < wr && bcol < wc) ? gld(xp + (size_t)(r0 + row) + (size_t)(cc + bcol) * ldx) : Scal<T>::zero();
    }
    pI[u] = gld(nd.invL + (size_t)(r0 / HS_PB) * BB + e);
  }
#pragma unroll
  for (int i = 0; i < NBLK; ++i) {
    if (i < nbv) {  // workgroup-uniform
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = t + 256 * u;
        Is[e] = pI[u];
#pragma unroll
        for (int k = 0; k < NBLK - 1; ++k)
          if (k < i) Ls[k * BB + e] = pL[k][u];
      }
      if (i + 1 < nbv) {  // the next block row's operands travel while this one is solved
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int e = t + 256 * u, a = e & 31, bcol = e >> 5;
          const int row = (i + 1) * HS_PB + a;
          pI[u] = gld(nd.invL + (size_t)(r0 / HS_PB + i + 1) * BB + e);
#pragma unroll
          for (int k = 0; k < NBLK - 1; ++k)
            if (k <= i) pL[k][u] = (row < wr) ? gld(LF + (size_t)(r0 + row) + (size_t)(r0 + k * HS_PB + bcol) * ldl) : Scal<T>::zero();
        }
      }
      __syncthreads();
      T acc[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = Xs[i * BB + t + 256 * u];
#pragma unroll
      for (int k = 0; k < NBLK - 1; ++k) {
        if (k < i) {
          const T* Lik = Ls + k * BB;
          const T* Yk = Xs + k * BB;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int e = t + 256 * u, a = e & 31, bcol = e >> 5;
            T v = acc[u];
#pragma unroll 8
            for (int q = 0; q < HS_PB; ++q) v = Scal<T>::fnma(Lik[a + q * HS_PB], Yk[q + bcol * HS_PB], v);
            acc[u] = v;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) Ss[t + 256 * u] = acc[u];
      __syncthreads();
      T y[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = t + 256 * u, a = e & 31, bcol = e >> 5;
        T v = Scal<T>::zero();
#pragma unroll 8
        for (int q = 0; q < HS_PB; ++q) v = Scal<T>::fma(Is[a + q * HS_PB], Ss[q + bcol * HS_PB], v);
        y[u] = v;
      }
      __syncthreads();  // every read of Is / Ls / Ss of this block row is done before the next one restages them
#pragma unroll
      for (int u = 0; u < 4; ++u) Xs[i * BB + t + 256 * u] = y[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int e = t + 256 * u, a = e & 31, bcol = e >> 5;
#pragma unroll
    for (int i = 0; i < NBLK; ++i) {
      const int row = i * HS_PB + a;
      if (row < wr && bcol < wc) gst(xp + (size_t)(r0 + row) + (size_t)(cc + bcol) * ldx, Xs[i * BB + e]);
    }
  }
}

// rows = 64 or 128 (ComplexF64: 64 only -- the blocks of 128 rows would not fit the LDS); false: the caller recurses as before
template <class T>
bool launch_trsm_small(const NodeDesc<T>* dnodes, int nbatch, int mat, int r0, int rows, int c0, int c1, int maxcols, hipStream_t s) {
  static const bool on = !(getenv("HS_TRSM_SMALL") && getenv("HS_TRSM_SMALL")[0] == '0');
  // ComplexF64: on since round 3 (HS_TRSM_SMALL_Z=0 turns it off).  Round 2 had switched it off when the matrix-free block flow at tol 1e-12
  // ended at 8.5e-8 instead of 1.3e-11: that was the rank-revealing orthogonalisation accepting round-off as rank (hs_hss.hip, HS_CHOL_COND /
  // HS_NOISE_REL; tests/test_mf_gpu.py::test_error_follows_the_tolerance_down_to_roundoff), which any round-off-level change of a kernel perturbed.
  static const bool on_z = !(getenv("HS_TRSM_SMALL_Z") && getenv("HS_TRSM_SMALL_Z")[0] == '0');
  if (!on || (sizeof(T) == 16 && !on_z) || nbatch <= 0 || maxcols <= 0) return false;
  const dim3 grid((maxcols + HS_PB - 1) / HS_PB, nbatch);
  if (rows == 2 * HS_PB) {
    constexpr int lds = (2 + 1 + 1 + 1) * HS_PB * HS_PB * (int)sizeof(T);
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)trsm_small_kernel<T, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      attr = true;
    }
    hipLaunchKernelGGL((trsm_small_kernel<T, 2>), grid, dim3(256), lds, s, dnodes, mat, r0, c0, c1);
    return true;
  }
  if constexpr (sizeof(T) == 8) {
    if (rows == 4 * HS_PB) {
      constexpr int lds = (4 + 1 + 3 + 1) * HS_PB * HS_PB * (int)sizeof(T);
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute((const void*)trsm_small_kernel<T, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
      }
      hipLaunchKernelGGL((trsm_small_kernel<T, 4>), grid, dim3(256), lds, s, dnodes, mat, r0, c0, c1);
      return true;
    }
  }
  return false;
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
void launch_panel_pivot(const NodeDesc<T>* dnodes, int nbatch, int pb, int fuse, hipStream_t s) {
  if (nbatch <= 0) return;
  if constexpr (sizeof(T) == 8) {
    static const bool merged = !(getenv("HS_PANEL_OPT") && getenv("HS_PANEL_OPT")[0] == '0');  // 0: the general kernel for optimistic panels too
    if ((fuse & 4) && merged) {
      hipLaunchKernelGGL(panel_pivot_opt_kernel, dim3(1, nbatch), dim3(256), 0, s, dnodes, pb, fuse);
      return;
    }
  }
  if constexpr (sizeof(T) == 16) {
    static const bool merged_z = !(getenv("HS_PANEL_OPT_Z") && getenv("HS_PANEL_OPT_Z")[0] == '0');  // 0: the general kernel for optimistic complex panels
    if ((fuse & 4) && merged_z) {
      constexpr int lds = 5 * HS_PB * HS_ZP_LD * (int)sizeof(cplx) + 5 * HS_PB * (int)sizeof(int);
      static bool attr = false;
      if (!attr) {  // 85 KB: above the 64 KB a kernel gets without the opt-in
        (void)hipFuncSetAttribute((const void*)panel_pivot_opt_z_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr = true;
      }
      hipLaunchKernelGGL(panel_pivot_opt_z_kernel, dim3(1, nbatch), dim3(256), lds, s, dnodes, pb, fuse);
      return;
    }
  }
  hipLaunchKernelGGL(panel_pivot_kernel<T>, dim3(1, nbatch), dim3(256), 0, s, dnodes, pb, fuse);
}
template <class T>
void launch_panel_l21(const NodeDesc<T>* dnodes, int nbatch, int pb, int maxrows, int fuse, hipStream_t s, int rlim) {
  if (nbatch <= 0 || maxrows <= 0) return;
  static const int rpt_env = getenv("HS_L21_ROWS") ? atoi(getenv("HS_L21_ROWS")) : 0;
  // measured at Poisson 128^3 (tools/env_sweep.sh): 2 rows per thread for the lone fronts 3.452 s, 1 row 3.449 s, 4 rows 3.484 s -- one row stays the default
  const int rpt = rpt_env > 0 ? rpt_env : 1;
  hipLaunchKernelGGL(panel_l21_kernel<T>, dim3((maxrows + 256 * rpt - 1) / (256 * rpt), nbatch), dim3(256), 0, s, dnodes, pb, fuse, rpt, rlim);
}
template <class T>
void launch_laswp(const NodeDesc<T>* dnodes, int nbatch, int mat, int c0, int c1, int k0, int k1, int maxcols, hipStream_t s) {
  if (nbatch <= 0 || maxcols <= 0 || k1 <= k0) return;
  hipLaunchKernelGGL(laswp_kernel<T>, dim3((maxcols + 255) / 256, nbatch), dim3(256), 0, s, dnodes, mat, c0, c1, k0, k1);
}

#define INST(T)                                                                                         \
  template void launch_tournament_round<T>(const NodeDesc<T>*, int, int, int, int, hipStream_t);        \
  template void launch_panel_pivot<T>(const NodeDesc<T>*, int, int, int, hipStream_t);                       \
  template void launch_panel_l21<T>(const NodeDesc<T>*, int, int, int, int, hipStream_t, int);               \
  template void launch_laswp<T>(const NodeDesc<T>*, int, int, int, int, int, int, int, hipStream_t);    \
  template bool launch_trsm_small<T>(const NodeDesc<T>*, int, int, int, int, int, int, int, hipStream_t); \
  template bool launch_group256<T>(const NodeDesc<T>*, int, int, hipStream_t);                             \

INST(double)
INST(cplx)
