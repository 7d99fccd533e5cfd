// kernels_gemm.hip -- FP64 / complex-FP64 MFMA GEMM for gfx950 (v_mfma_f64_16x16x4_f64).
//
// C (M x N) -= A (M x K) * B (K x N)   or   C = A * B, all column-major (Julia Matrix{T}).
// This is the dense contraction of the elimination: the Schur-complement update
// S = Abb - Abi*R (reference src/factorization.jl:40,72), the products inside blockfactor /
// blockldiv / blockrdiv (src/blockmatrix.jl:118,163-170,178-185) and every trailing update of the
// blocked interior LU that replaces the reference's `\` and `/`.
//
// Tiling (real): 128x128 output tile per 256-thread workgroup, 4 waves as 2(M) x 2(N), each wave
// 64x64 = 4x4 MFMA tiles of 16x16 (64 f64 accumulators = 128 VGPRs/lane, 2 workgroups per CU).
// K is consumed in steps of BK=16 staged through LDS ([k][m] / [k][n] images, row stride 144 so the
// two 16-lane halves of a ds_read_b64 land on disjoint bank halves); the next K-step's global
// loads are issued into registers before the MFMAs of the current one.
//
// The MFMA is issued "transposed": its A operand carries B's column index and its B operand A's
// row index, so that lane&15 runs along C's rows -- contiguous in column-major memory -- and every
// C load/store instruction touches four 128-byte row segments instead of sixteen 32-byte ones.
// f64 C/D map (cdna guide section 3): col = lane&15, row = (lane>>4) + 4*reg.
//
// Complex: planar split when staging (re / im images in LDS), 4 real MFMAs per k-step
// (re += ar*br - ai*bi, im += ar*bi + ai*br); 128x64 tile, wave tile 64x32.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "hs_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_u __attribute__((ext_vector_type(2), aligned(8)));  // 16-byte load, 8-byte aligned

#define BM 128
#define BK 16
#define LDS_LD 144  // 128 + 16: stride == 16 (mod 32) doubles => conflict-free ds_read_b64 across the two k rows of a half-wave

// bijective XCD-aware remap of a 1-D block id (cdna guide section 5, "XCD swizzle must be bijective"):
// blocks b and b+8 share an XCD; give each XCD a contiguous chunk of tile ids so neighbouring tiles
// (which share A row-panels / B column-panels) hit the same L2.
__device__ inline int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, x = bid & 7, o = bid >> 3;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + o;
}

// tile id -> (tile_m, tile_n): walk column-panels of GROUP_M tiles so a group shares B panels and re-uses A panels
__device__ inline void tile_coords(int t, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int GROUP_M = 8;
  int per_group = GROUP_M * tiles_n;
  int g = t / per_group;
  int first_m = g * GROUP_M;
  int gm = min(tiles_m - first_m, GROUP_M);
  int in_g = t - g * per_group;
  tm = first_m + in_g % gm;
  tn = in_g / gm;
}

template <class T>
__device__ inline bool resolve_op(const NodeDesc<T>* pn, const GemmOp& op, GemmProb<T>& p) {
  T *cp, *bp;
  int ldc, ldb, crows, ccols, brows, bcols;
  mat_of(pn, op.cmat, cp, ldc, crows, ccols);
  mat_of(pn, op.bmat, bp, ldb, brows, bcols);
  const int ni = pn->ni, ldl = pn->ldl;
  T* const LF = pn->LF;
  int r1 = min(op.r1, crows), c1 = min(op.c1, ccols), k1 = min(op.k1, ni);
  int M = r1 - op.r0, N = c1 - op.c0, K = k1 - op.k0;
  if (M <= 0 || N <= 0 || K <= 0) return false;
  int aoff = (op.cmat == HS_MAT_SB) ? ni : 0;
  p.A = LF + (size_t)(op.r0 + aoff) + (size_t)op.k0 * ldl;
  p.B = bp + (size_t)op.k0 + (size_t)op.c0 * ldb;
  p.C = cp + (size_t)op.r0 + (size_t)op.c0 * ldc;
  p.M = M; p.N = N; p.K = K;
  p.lda = ldl; p.ldb = ldb; p.ldc = ldc;
  return true;
}

// ------------------------------------------------------------------------------------------------
// real double
// ------------------------------------------------------------------------------------------------
template <int BN_>
__device__ inline void gemm_tile_d(const GemmProb<double>& p, int tile_m, int tile_n, bool minus, double* smem) {
  constexpr int BN = BN_;  // 128
  double* As = smem;                 // [BK][LDS_LD]
  double* Bs = smem + BK * LDS_LD;   // [BK][LDS_LD]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int M = p.M, N = p.N, K = p.K;
  const double* __restrict__ A = p.A;
  const double* __restrict__ B = p.B;

  // staging coordinates
  const int a_pair = tid & 63;  // rows 2*a_pair, 2*a_pair+1
  const int a_k = tid >> 6;     // + 4*i
  const int b_kp = tid & 7;     // k = 2*b_kp, 2*b_kp+1
  const int b_n = tid >> 3;     // + 32*i

  double2_u ra[4], rb[4];

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kk = k0 + a_k + 4 * i;
      int mm = m0 + 2 * a_pair;
      double2_u v = {0.0, 0.0};
      if (kk < K) {
        const double* src = A + (size_t)mm + (size_t)kk * p.lda;
        if (mm + 1 < M) {
          v = *reinterpret_cast<const double2_u*>(src);
        } else if (mm < M) {
          v.x = src[0];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int nn = n0 + b_n + 32 * i;
      int kk = k0 + 2 * b_kp;
      double2_u v = {0.0, 0.0};
      if (nn < N) {
        const double* src = B + (size_t)kk + (size_t)nn * p.ldb;
        if (kk + 1 < K) {
          v = *reinterpret_cast<const double2_u*>(src);
        } else if (kk < K) {
          v.x = src[0];
        }
      }
      rb[i] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double* dst = As + (a_k + 4 * i) * LDS_LD + 2 * a_pair;
      dst[0] = ra[i].x;
      dst[1] = ra[i].y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int nn = b_n + 32 * i;
      Bs[(2 * b_kp) * LDS_LD + nn] = rb[i].x;
      Bs[(2 * b_kp + 1) * LDS_LD + nn] = rb[i].y;
    }
  };

  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int l15 = lane & 15, l4 = lane >> 4;
  const double* a_rd = As + l4 * LDS_LD + wm * 64 + l15;
  const double* b_rd = Bs + l4 * LDS_LD + wn * 64 + l15;

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = (k0 + BK) < K;
    if (more) load_tile(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      double af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = a_rd[ks * LDS_LD + i * 16];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = b_rd[ks * LDS_LD + j * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // transposed issue: MFMA-A <- B data (n on the register/row axis), MFMA-B <- A data (m on lane&15)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j], af[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }

  // epilogue: acc[i][j][r] <-> C[m0 + wm*64 + i*16 + l15][n0 + wn*64 + j*16 + l4 + 4r]
  double* __restrict__ C = p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int mm = m0 + wm * 64 + i * 16 + l15;
    if (mm >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int nn = n0 + wn * 64 + j * 16 + l4 + 4 * r;
        if (nn < N) {
          double* c = C + (size_t)mm + (size_t)nn * p.ldc;
          *c = minus ? (*c - acc[i][j][r]) : acc[i][j][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// complex double (planar split in LDS)
// ------------------------------------------------------------------------------------------------
#define ZBN 64
#define ZLDB 80  // 64 + 16
__device__ inline void gemm_tile_z(const GemmProb<cplx>& p, int tile_m, int tile_n, bool minus, double* smem) {
  double* Ar = smem;
  double* Ai = Ar + BK * LDS_LD;
  double* Br = Ai + BK * LDS_LD;
  double* Bi = Br + BK * ZLDB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = tile_m * BM, n0 = tile_n * ZBN;
  const int M = p.M, N = p.N, K = p.K;
  const cplx* __restrict__ A = p.A;
  const cplx* __restrict__ B = p.B;

  // A tile 128 x 16 complex: thread -> row (tid&127), k = (tid>>7) + 2*i, i<8
  const int a_m = tid & 127, a_k = tid >> 7;
  // B tile 16 x 64 complex: thread -> k = tid&15, n = (tid>>4) + 16*i, i<4
  const int b_k = tid & 15, b_n = tid >> 4;
  double2_u ra[8], rb[4];

  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int kk = k0 + a_k + 2 * i, mm = m0 + a_m;
      double2_u v = {0.0, 0.0};
      if (kk < K && mm < M) v = *reinterpret_cast<const double2_u*>(A + (size_t)mm + (size_t)kk * p.lda);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int kk = k0 + b_k, nn = n0 + b_n + 16 * i;
      double2_u v = {0.0, 0.0};
      if (kk < K && nn < N) v = *reinterpret_cast<const double2_u*>(B + (size_t)kk + (size_t)nn * p.ldb);
      rb[i] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int o = (a_k + 2 * i) * LDS_LD + a_m;
      Ar[o] = ra[i].x;
      Ai[o] = ra[i].y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int o = b_k * ZLDB + b_n + 16 * i;
      Br[o] = rb[i].x;
      Bi[o] = rb[i].y;
    }
  };

  double4_t accr[4][2], acci[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      accr[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
      acci[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    }
  const int l15 = lane & 15, l4 = lane >> 4;
  const int a_off = l4 * LDS_LD + wm * 64 + l15;
  const int b_off = l4 * ZLDB + wn * 32 + l15;

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = (k0 + BK) < K;
    if (more) load_tile(k0 + BK);
#pragma unroll
    for (int ks = 0; ks < BK; ks += 4) {
      double afr[4], afi[4], bfr[2], bfi[2], bfin[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        afr[i] = Ar[a_off + ks * LDS_LD + i * 16];
        afi[i] = Ai[a_off + ks * LDS_LD + i * 16];
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bfr[j] = Br[b_off + ks * ZLDB + j * 16];
        bfi[j] = Bi[b_off + ks * ZLDB + j * 16];
        bfin[j] = -bfi[j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          accr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[j], afr[i], accr[i][j], 0, 0, 0);
          accr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfin[j], afi[i], accr[i][j], 0, 0, 0);
          acci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfi[j], afr[i], acci[i][j], 0, 0, 0);
          acci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[j], afi[i], acci[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (more) {
      store_tile();
      __syncthreads();
    }
  }
  cplx* __restrict__ C = p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int mm = m0 + wm * 64 + i * 16 + l15;
    if (mm >= M) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int nn = n0 + wn * 32 + j * 16 + l4 + 4 * r;
        if (nn < N) {
          double2_u* c = reinterpret_cast<double2_u*>(C + (size_t)mm + (size_t)nn * p.ldc);
          double2_u v = {accr[i][j][r], acci[i][j][r]};
          if (minus) {
            double2_u o = *c;
            v.x = o.x - v.x;
            v.y = o.y - v.y;
          }
          *c = v;
        }
      }
  }
}

template <class T>
struct TileCfg;
template <>
struct TileCfg<double> {
  static constexpr int bn = 128;
  static constexpr int smem_doubles = 2 * BK * LDS_LD;
};
template <>
struct TileCfg<cplx> {
  static constexpr int bn = ZBN;
  static constexpr int smem_doubles = 2 * BK * LDS_LD + 2 * BK * ZLDB;
};

template <class T>
__device__ inline void gemm_dispatch(const GemmProb<T>& p, bool minus, double* smem) {
  int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + TileCfg<T>::bn - 1) / TileCfg<T>::bn;
  int ntiles = tiles_m * tiles_n;
  if ((int)blockIdx.x >= ntiles) return;
  int t = xcd_remap(blockIdx.x, ntiles);
  int tm, tn;
  tile_coords(t, tiles_m, tiles_n, tm, tn);
  if constexpr (sizeof(T) == 8)
    gemm_tile_d<128>(p, tm, tn, minus, smem);
  else
    gemm_tile_z(p, tm, tn, minus, smem);
}

template <class T>
__global__ __launch_bounds__(256, 2) void gemm_op_kernel(const NodeDesc<T>* __restrict__ nodes, GemmOp op) {
  __shared__ double smem[TileCfg<T>::smem_doubles];
  GemmProb<T> p;
  if (!resolve_op(nodes + blockIdx.y, op, p)) return;
  gemm_dispatch<T>(p, true, smem);
}

template <class T>
__global__ __launch_bounds__(256, 2) void gemm_probs_kernel(const GemmProb<T>* __restrict__ probs, int minus) {
  __shared__ double smem[TileCfg<T>::smem_doubles];
  GemmProb<T> p = probs[blockIdx.y];
  if (p.M <= 0 || p.N <= 0) return;
  gemm_dispatch<T>(p, minus != 0, smem);
}

template <class T>
__global__ void resolve_dump_kernel(const NodeDesc<T>* __restrict__ nodes, GemmOp op, GemmProb<T>* out, int* ok) {
  GemmProb<T> p;
  memset(&p, 0, sizeof p);
  ok[blockIdx.x] = resolve_op(nodes + blockIdx.x, op, p) ? 1 : 0;
  out[blockIdx.x] = p;
}

template <class T>
void launch_gemm_op(const NodeDesc<T>* dnodes, int nbatch, int maxM, int maxN, const GemmOp& op, hipStream_t s) {
  if (nbatch <= 0 || maxM <= 0 || maxN <= 0) return;
  {
    static int dbg = -1;
    if (dbg < 0) {
      const char* e = getenv("HS_DEBUG_SYNC");
      dbg = (e && e[0] == '2') ? 1 : 0;
    }
    if (dbg) {  // diagnostics: print what every front resolves this op to (no arithmetic)
      GemmProb<T>* dp;
      int* dok;
      (void)hipMalloc((void**)&dp, sizeof(GemmProb<T>) * nbatch);
      (void)hipMalloc((void**)&dok, sizeof(int) * nbatch);
      hipLaunchKernelGGL(resolve_dump_kernel<T>, dim3(nbatch), dim3(1), 0, s, dnodes, op, dp, dok);
      (void)hipStreamSynchronize(s);
      GemmProb<T> hp;
      int hok;
      NodeDesc<T> hn;
      for (int i = 0; i < nbatch && i < 4; ++i) {
        (void)hipMemcpy(&hp, dp + i, sizeof hp, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hok, dok + i, sizeof hok, hipMemcpyDeviceToHost);
        (void)hipMemcpy(&hn, dnodes + i, sizeof hn, hipMemcpyDeviceToHost);
        fprintf(stderr, "[hs debug] gemm op c%d b%d r[%d,%d) c[%d,%d) k[%d,%d) node%d: ok=%d A=%p B=%p C=%p M=%d N=%d K=%d ld=%d,%d,%d | LF=%p UR=%p SB=%p ni=%d nb=%d m=%d ld=%d,%d,%d maxM=%d maxN=%d\n",
                op.cmat, op.bmat, op.r0, op.r1, op.c0, op.c1, op.k0, op.k1, i, hok, (void*)hp.A, (void*)hp.B, (void*)hp.C, hp.M, hp.N, hp.K, hp.lda,
                hp.ldb, hp.ldc, (void*)hn.LF, (void*)hn.UR, (void*)hn.SB, hn.ni, hn.nb, hn.m, hn.ldl, hn.ldu, hn.lds, maxM, maxN);
      }
      (void)hipFree(dp);
      (void)hipFree(dok);
    }
  }
  int tiles = ((maxM + BM - 1) / BM) * ((maxN + TileCfg<T>::bn - 1) / TileCfg<T>::bn);
  hipLaunchKernelGGL(gemm_op_kernel<T>, dim3(tiles, nbatch), dim3(256), 0, s, dnodes, op);
}
template <class T>
void launch_gemm_probs(const GemmProb<T>* dprobs, int nprob, int maxM, int maxN, int minus, hipStream_t s) {
  if (nprob <= 0 || maxM <= 0 || maxN <= 0) return;
  int tiles = ((maxM + BM - 1) / BM) * ((maxN + TileCfg<T>::bn - 1) / TileCfg<T>::bn);
  hipLaunchKernelGGL(gemm_probs_kernel<T>, dim3(tiles, nprob), dim3(256), 0, s, dprobs, minus);
}

template void launch_gemm_op<double>(const NodeDesc<double>*, int, int, int, const GemmOp&, hipStream_t);
template void launch_gemm_op<cplx>(const NodeDesc<cplx>*, int, int, int, const GemmOp&, hipStream_t);
template void launch_gemm_probs<double>(const GemmProb<double>*, int, int, int, int, hipStream_t);
template void launch_gemm_probs<cplx>(const GemmProb<cplx>*, int, int, int, int, hipStream_t);

// ------------------------------------------------------------------------------------------------
// FP64 MFMA issue-rate microbenchmark: the guides list no f64 MFMA peak (MI355X_MICROARCH.md
// "Matrix cores" has no f64 row), so the roofline denominator is measured (bench.py --mfma-peak).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_f64_rate_kernel(double* out, int iters) {
  double4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// returns measured TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 (every CU, waves_per_simd waves per SIMD)
extern "C" double hsk_mfma_f64_peak(int waves_per_simd, int iters) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1.0;
  int cus = prop.multiProcessorCount;
  int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  double* out = nullptr;
  if (hipMalloc(&out, sizeof(double) * blocks * 256) != hipSuccess) return -1.0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, 0, out, iters / 10 + 1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(out);
  double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
  return flops / (ms * 1e-3) / 1e12;
}
