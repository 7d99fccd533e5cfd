// kernels_tourn.hip -- tournament pivoting (CALU) for one 32-column panel of a batch of fronts.
//
// Replaces the pivot search of the `getrf` hidden in every `\` and `/` of the reference
// (src/factorization.jl:36-37, src/blockmatrix.jl:118,162-185).
//
// A workgroup of 512 threads runs Gaussian elimination with partial pivoting on a chunk of 512*R
// candidate rows held in registers (R rows per thread: 2 for Float64 = 1024 rows, 1 for ComplexF64 =
// 512 rows) and nominates the 32 pivot rows it picked; the nominees play off in the next stage until one
// workgroup is left, whose picks ARE partial pivoting on the survivors.  The chain of 32 dependent
// elimination steps per stage is the latency floor of the panel, so the chunk is as large as the
// register file allows: 32,768 rows need 2 stages (the previous 256-row chunks needed 4), 1,024 rows 1.
//
// Per step: 64-bit key = |a| bits with the candidate's index in the low 10 bits (ties go to the lowest
// candidate), DPP max inside the wave, one LDS slot per wave and ONE barrier; the winner publishes its
// row through LDS (second barrier); every thread forms the reciprocal itself (v_rcp_f64 + a Newton step:
// the values eliminated here only steer the pivot ORDER -- the factors are recomputed by panel_pivot /
// panel_l21 from the original rows).  A row that was picked, or never existed, is an exactly-zero row
// from then on (its multiplier is forced to 1), so there are no liveness flags or divergent updates.
#include <cstdlib>
#include "hs_common.h"

template <class T>
struct Tour;
template <>
struct Tour<double> {
  static constexpr int R = 2;
};
template <>
struct Tour<cplx> {
  static constexpr int R = 1;
};
#define HS_TOUR_NT 512

int hs_tour_block_rows(bool is_complex) { return HS_TOUR_NT * (is_complex ? Tour<cplx>::R : Tour<double>::R); }

void hs_create_lookahead_streams(hipStream_t* la, hipStream_t* side_masked, hipStream_t* side) {
  *la = nullptr;
  *side_masked = nullptr;
  *side = nullptr;
  int lo = 0, hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&lo, &hi);  // hi = numerically lowest = highest priority
  if (hipStreamCreateWithPriority(side, hipStreamNonBlocking, hi) != hipSuccess) {
    *side = nullptr;
    (void)hipGetLastError();
  }
  const char* e = getenv("HS_LA_SIDE_CUS");
  int ncu = e ? atoi(e) : 0;  // opt-in: 32 reserved CUs took 7 % off a lone 32,768 front, but rocprofv3 crashed on CU-masked queues
  hipDeviceProp_t prop;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
  if (ncu > 0 && ncu * 2 <= prop.multiProcessorCount && prop.multiProcessorCount <= 512) {
    uint32_t m_side[16], m_main[16];
    const int words = (prop.multiProcessorCount + 31) / 32;
    for (int i = 0; i < 16; ++i) m_side[i] = 0;
    for (int b = 0; b < ncu; ++b) m_side[b >> 5] |= 1u << (b & 31);
    for (int i = 0; i < 16; ++i) m_main[i] = ~m_side[i];
    if (hipExtStreamCreateWithCUMask(la, words, m_main) == hipSuccess) {
      if (hipExtStreamCreateWithCUMask(side_masked, words, m_side) == hipSuccess) return;
      (void)hipStreamDestroy(*la);
    }
    *la = nullptr;
    *side_masked = nullptr;
    (void)hipGetLastError();
  }
}

namespace {

__device__ inline unsigned long long dppmax(unsigned long long v, const int sel) {
  int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
  int olo, ohi;
  switch (sel) {
    case 0: olo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, false); break;    // quad_perm [1,0,3,2]
    case 1: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, false); break;    // quad_perm [2,3,0,1]
    case 2: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, false); break;  // row_half_mirror
    default: olo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, false); ohi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, false); break; // row_mirror
  }
  unsigned long long o = ((unsigned long long)(unsigned)ohi << 32) | (unsigned)olo;
  return o > v ? o : v;
}
__device__ inline unsigned long long wavemax(unsigned long long v) {
  v = dppmax(v, 0);
  v = dppmax(v, 1);
  v = dppmax(v, 2);
  v = dppmax(v, 3);  // every row of 16 lanes holds its maximum
  int lo = (int)(unsigned)(v & 0xffffffffull), hi = (int)(unsigned)(v >> 32);
  unsigned long long m = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    unsigned long long x = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, 16 * r) << 32) | (unsigned)__builtin_amdgcn_readlane(lo, 16 * r);
    m = x > m ? x : m;
  }
  return m;
}

__device__ inline double recip(double x) {
  double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}
__device__ inline cplx recip(cplx x) {
  double d = recip(fma(x.re, x.re, x.im * x.im));
  return {x.re * d, -x.im * d};
}

}  // namespace

// stage s reads the nominees of stage s-1 (stage 0: all rows [c0, pivrows)) and writes 32 nominees per workgroup;
// the stage that runs with a single workgroup writes pivlist.
template <class T>
__global__ __launch_bounds__(HS_TOUR_NT) void tournament_stage_kernel(const NodeDesc<T>* __restrict__ nodes, int pb, int stage) {
  constexpr int R = Tour<T>::R, NT = HS_TOUR_NT, CH = NT * R, NWV = NT / 64;
  const NodeDesc<T>* pn = nodes + blockIdx.y;
  const int ni = pn->ni, c0 = pb * HS_PB;
  if (c0 >= ni) return;
  const int w = min(HS_PB, ni - c0);
  int cnt = pn->pivrows - c0;
  int nch = (cnt + CH - 1) / CH;
  for (int s = 0; s < stage; ++s) {
    if (nch == 1) return;  // this front finished in an earlier stage
    cnt = nch * HS_PB;
    nch = (cnt + CH - 1) / CH;
  }
  const int chunk = blockIdx.x;
  if (chunk >= nch) return;
  const int* cin = (stage & 1) ? pn->cand0 : pn->cand1;  // written by stage-1
  int* cout = (nch == 1) ? pn->pivlist : (((stage & 1) ? pn->cand1 : pn->cand0) + (size_t)chunk * HS_PB);
  const T* LF = pn->LF;
  const int ldl = pn->ldl;

  const int t = threadIdx.x;
  int row[R];
  T a[R][HS_PB];
#pragma unroll
  for (int s = 0; s < R; ++s) {
    const int q = chunk * CH + s * NT + t;
    int r = -1;
    if (q < cnt) r = (stage == 0) ? (c0 + q) : cin[q];
    row[s] = r;
    const T* src = LF + (size_t)max(r, 0) + (size_t)c0 * ldl;
#pragma unroll
    for (int j = 0; j < HS_PB; ++j) {
      T v = Scal<T>::zero();
      if (r >= 0 && j < w) v = src[(size_t)j * ldl];
      a[s][j] = v;
    }
  }

  __shared__ T prow[2][HS_PB];                    // winner's row, double-buffered by k parity
  __shared__ int prid[2];                         // winner's row id
  __shared__ unsigned long long wkey[2][NWV];     // per-wave maxima, double-buffered by k parity

#pragma unroll
  for (int k = 0; k < HS_PB; ++k) {
    if (k >= w) {  // narrow last panel (workgroup-uniform)
      if (t == 0) cout[k] = -1;
      continue;
    }
    unsigned long long key = 0;
#pragma unroll
    for (int s = 0; s < R; ++s) {
      const unsigned long long ks = ((unsigned long long)__double_as_longlong(Scal<T>::abs1(a[s][k])) & ~0x3ffull) | (unsigned long long)(1023 - (s * NT + t));
      key = ks > key ? ks : key;
    }
    const unsigned long long wm = wavemax(key);
    if ((t & 63) == 0) wkey[k & 1][t >> 6] = wm;
    __syncthreads();
    unsigned long long best = wkey[k & 1][0];
#pragma unroll
    for (int v = 1; v < NWV; ++v) best = wkey[k & 1][v] > best ? wkey[k & 1][v] : best;
    if ((best >> 10) == 0) {  // the column is exactly zero on every remaining row: no pivot (workgroup-uniform)
      if (t == 0) cout[k] = -1;
      continue;
    }
    const int idx = 1023 - (int)(best & 0x3ff);
    const int ws = idx / NT, wt = idx - ws * NT;
    if (t == wt) {
#pragma unroll
      for (int s = 0; s < R; ++s) {
        if (ws == s) {
#pragma unroll
          for (int j = 0; j < HS_PB; ++j)
            if (j >= k) prow[k & 1][j] = a[s][j];
          cout[k] = row[s];
        }
      }
    }
    __syncthreads();
    if (k + 1 < HS_PB) {
      const T rc = recip(prow[k & 1][k]);
      T l[R];
#pragma unroll
      for (int s = 0; s < R; ++s) {
        l[s] = a[s][k] * rc;
        if (t == wt && ws == s) l[s] = Scal<T>::one();  // the winner becomes an exactly-zero row
      }
#pragma unroll
      for (int j = 0; j < HS_PB; ++j) {
        if (j > k) {
          const T p = prow[k & 1][j];
#pragma unroll
          for (int s = 0; s < R; ++s) a[s][j] = Scal<T>::fnma(l[s], p, a[s][j]);
        }
      }
    }
  }
}

template <class T>
void launch_tournament_stage(const NodeDesc<T>* dnodes, int nbatch, int pb, int stage, int maxblocks, hipStream_t s) {
  if (nbatch <= 0 || maxblocks <= 0) return;
  hipLaunchKernelGGL(tournament_stage_kernel<T>, dim3(maxblocks, nbatch), dim3(HS_TOUR_NT), 0, s, dnodes, pb, stage);
}
template void launch_tournament_stage<double>(const NodeDesc<double>*, int, int, int, int, hipStream_t);
template void launch_tournament_stage<cplx>(const NodeDesc<cplx>*, int, int, int, int, hipStream_t);
