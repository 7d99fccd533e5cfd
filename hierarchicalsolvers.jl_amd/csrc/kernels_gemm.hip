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
#include <mutex>
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "hs_common.h"

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef hs_d2u double2_u;

#define BM 128
#define BK 16
#define LDB_S 18   // B image is [n][k] with 16 + 2 padding: ds_write_b128 of a k-pair row segment and the MFMA-operand ds_read_b64 are both conflict-free
#define LDS_LD 144  // 128 + 16: stride == 16 (mod 32) doubles => conflict-free ds_read_b64 across the two k rows of a half-wave

// bijective XCD-aware remap of a 1-D block id (cdna guide section 5, "XCD swizzle must be bijective"):
// blocks b and b+8 share an XCD; give each XCD a contiguous chunk of tile ids so neighbouring tiles
// (which share A row-panels / B column-panels) hit the same L2.
__device__ inline int xcd_remap(int bid, int nwg) {
  int q = nwg >> 3, r = nwg & 7, x = bid & 7, o = bid >> 3;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + o;
}
// The same for a front of a BATCHED launch (grid = tiles x fronts): the hardware deals workgroups to the XCDs round-robin over the LINEAR
// workgroup id blockIdx.x + gridDim.x * blockIdx.y, so block `bid` of front y sits on XCD (bid + shift) & 7 with shift = (gridDim.x * y) & 7.
// Round 2 assumed bid & 7 for every front: with gridDim.x not a multiple of 8 the chunks of every front but the first grouped tiles of
// DIFFERENT L2s.  Bijective on [0, nwg): XCD x owns the bids first_x, first_x + 8, ... and gets a contiguous chunk.
__device__ inline int xcd_remap_shift(int bid, int nwg, int shift) {
  if (shift == 0) return xcd_remap(bid, nwg);
  const int x = (bid + shift) & 7;
  int base = 0;
#pragma unroll
  for (int xx = 0; xx < 8; ++xx) {
    const int first = (xx - shift) & 7;                       // smallest bid on XCD xx
    const int cnt = first < nwg ? (nwg - first + 7) >> 3 : 0;  // bids of this front on XCD xx
    base += (xx < x) ? cnt : 0;
  }
  return base + ((bid - ((x - shift) & 7)) >> 3);
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
  if (op.ainv >= 16) {
    // The same product for tiles that are 64 columns wide (ComplexF64): column block q = ainv - 16 of the group, from the right (3, 2, 1, 0):
    //   L[r0.., k0+64q : k0+64q+64) <- A[r0.., k0 : k0+64(q+1)) * V[0 : 64(q+1), 64q : 64q+64)      (V = inv(U_group) is upper triangular)
    // One tile column per launch: a workgroup has read all of its K before it stores, no other workgroup touches its rows, and the blocks to the
    // LEFT of q -- the only columns block q reads besides its own -- are still the original A when block q runs.
    const int q = op.ainv - 16;
    const int wl = min(256, ni - op.k0);
    const int M = min(op.r1, crows) - op.r0;
    const int N = min(64, wl - 64 * q);
    if (wl <= 0 || M <= 0 || N <= 0 || q > 3) return false;
    const T* V = pn->inv256U + (size_t)(op.k0 / 256) * 65536;
    T* X = LF + (size_t)op.r0 + (size_t)op.k0 * ldl;
    p.A = X; p.B = V + (size_t)(64 * q) * 256; p.C = X + (size_t)(64 * q) * ldl;
    p.M = M; p.N = N; p.K = min(wl, 64 * (q + 1));
    p.lda = ldl; p.ldb = 256; p.ldc = ldl;
    p.flag = pn->growth;
    p.flag_rows = pn->pivrows - op.r0;
    return true;
  }
  if (op.ainv >= 7) {
    // L[r0.., k0 : k0+wl) <- A[r0.., k0 : k0+wl) * inv(U[k0 : k0+wl, k0 : k0+wl)), in place as two products that read every column they
    // overwrite before their stores (one tile column each: N <= 128): 7 = columns 128.. from all wl columns (first), 8 = columns 0..127
    const int wl = min(256, ni - op.k0);
    const int M = min(op.r1, crows) - op.r0;
    if (wl <= 0 || M <= 0) return false;
    const T* V = pn->inv256U + (size_t)(op.k0 / 256) * 65536;
    T* X = LF + (size_t)op.r0 + (size_t)op.k0 * ldl;
    if (op.ainv == 7) {
      if (wl <= 128) return false;
      p.A = X; p.B = V + (size_t)128 * 256; p.C = X + (size_t)128 * ldl;
      p.M = M; p.N = wl - 128; p.K = wl;
    } else {
      p.A = X; p.B = V; p.C = X;
      p.M = M; p.N = min(128, wl); p.K = min(128, wl);
    }
    p.lda = ldl; p.ldb = 256; p.ldc = ldl;
    p.flag = pn->growth;
    p.flag_rows = pn->pivrows - op.r0;
    return true;
  }
  if (op.ainv >= 3) {
    // 256-row TRSM base case X[r0:r0+256, c0:c1) <- inv256 * X, in place, as two half products that never read a row
    // another workgroup may already have overwritten (each has one tile row and reads all of K before its stores):
    //   lower (L): 3 = rows 128.. <- inv[128:256, 0:256] * X[0:256]   (first),  4 = rows 0..127 <- inv[0:128, 0:128] * X[0:128]
    //   upper (U): 5 = rows 0..127 <- inv[0:128, 0:256] * X[0:256]    (first),  6 = rows 128.. <- inv[128:256, 128:256] * X[128:256]
    const int wl = min(256, ni - op.r0);
    const int N = min(op.c1, ccols) - op.c0;
    if (wl <= 0 || N <= 0) return false;
    const bool up = op.ainv >= 5;
    const T* inv = (up ? pn->inv256U : pn->inv256L) + (size_t)(op.r0 / 256) * 65536;
    const bool second_half_rows = (op.ainv == 3 || op.ainv == 6);  // output rows 128..
    const bool full_k = (op.ainv == 3 || op.ainv == 5);
    const int ro = second_half_rows ? 128 : 0;
    const int M = min(128, wl - ro);
    if (M <= 0) return false;
    const int ko = (op.ainv == 6) ? 128 : 0;
    const int K = full_k ? wl : (op.ainv == 4 ? min(128, wl) : wl - 128);
    if (K <= 0) return false;
    p.A = inv + ro + (size_t)ko * 256;
    p.B = cp + (size_t)(op.r0 + ko) + (size_t)op.c0 * ldc;
    p.C = cp + (size_t)(op.r0 + ro) + (size_t)op.c0 * ldc;
    p.M = M; p.N = N; p.K = K;
    p.lda = 256; p.ldb = ldc; p.ldc = ldc;
    return true;
  }
  if (op.ainv) {  // X[r0:r0+w, c0:c1) <- inv(L11[r0/32]) * X, in place (one tile row, all of K is read before the stores)
    const int w = min(HS_PB, ni - op.r0);
    const int N = min(op.c1, ccols) - op.c0;
    if (w <= 0 || N <= 0) return false;
    p.A = (op.ainv == 2 ? pn->invU : pn->invL) + (size_t)(op.r0 / HS_PB) * HS_PB * HS_PB;  // 2: inverse of the UPPER diagonal block
    p.B = cp + (size_t)op.r0 + (size_t)op.c0 * ldc;
    p.C = cp + (size_t)op.r0 + (size_t)op.c0 * ldc;
    p.M = w; p.N = N; p.K = w;
    p.lda = HS_PB; p.ldb = ldc; p.ldc = ldc;
    return true;
  }
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
  constexpr int STAGE = 2 * BK * LDS_LD;  // one LDS stage: As [BK][LDS_LD] then Bs [BK][LDS_LD]; two stages
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

  // Global -> register staging.  Interior tiles take 16-byte loads with no guards.  Edge tiles use
  // clamped 8-byte loads + selects: a per-lane `if (in range) load` compiles to a branch with an
  // s_waitcnt vmcnt(0) at every join, i.e. eight fully serialised memory round trips per K-step
  // (that cost 33 % of every wave's life in SQ_WAIT_ANY before this was restructured).
  // Ragged batches (fronts of one level differ in ni / nb) give EVERY front an edge tile in n: at the leaf level of Poisson 128^3 9-17 %
  // of all tiles, which the clamped 8-byte path made 4-5x slower than interior ones (the level ran at the rate of its bounding box).
  // The operands are independent: a tile whose rows are all inside takes the 16-byte A loads whatever its columns do, and for a full
  // K-step the B loads stay 16 bytes wide with the COLUMN clamped -- what lands in columns >= N only feeds accumulators that are
  // never stored (C[:, n] depends on B[:, n] alone).  Only a partial K-step (the last one of a tile) needs the zero-filling path.
  const bool rows_in = (m0 + BM <= M), rows_even = (M & 1) == 0;  // M >= 2 when even (M > 0)
  auto load_tile = [&](int k0) {
    const bool kfull = k0 + BK <= K;
    if ((rows_in || rows_even) && kfull) {  // an even row count: a row pair is inside or outside as a whole, outside pairs re-read the last one
#pragma unroll
      for (int i = 0; i < 4; ++i)
        ra[i] = gld2(A + (size_t)min(m0 + 2 * a_pair, M - 2) + (size_t)(k0 + a_k + 4 * i) * p.lda);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = k0 + a_k + 4 * i, mm = m0 + 2 * a_pair;
        const double* col = A + (size_t)min(kk, K - 1) * p.lda;
        double x = gld(col + min(mm, M - 1)), y = gld(col + min(mm + 1, M - 1));
        const bool kok = kk < K;
        ra[i].x = (kok && mm < M) ? x : 0.0;
        ra[i].y = (kok && mm + 1 < M) ? y : 0.0;
      }
    }
    if (kfull) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        rb[i] = gld2(B + (size_t)(k0 + 2 * b_kp) + (size_t)min(n0 + b_n + 32 * i, N - 1) * p.ldb);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int nn = n0 + b_n + 32 * i, kk = k0 + 2 * b_kp;
        const double* col = B + (size_t)min(nn, N - 1) * p.ldb;
        double x = gld(col + min(kk, K - 1)), y = gld(col + min(kk + 1, K - 1));
        const bool nok = nn < N;
        rb[i].x = (nok && kk < K) ? x : 0.0;
        rb[i].y = (nok && kk + 1 < K) ? y : 0.0;
      }
    }
  };
  auto store_tile = [&](int stage) {
    double* As = smem + stage * STAGE;
    double* Bs = As + BK * LDS_LD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double* dst = As + (a_k + 4 * i) * LDS_LD + 2 * a_pair;
      dst[0] = ra[i].x;
      dst[1] = ra[i].y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      double* dst = Bs + (b_n + 32 * i) * LDB_S + 2 * b_kp;  // 16-byte aligned: 144 n + 16 kp bytes
      dst[0] = rb[i].x;
      dst[1] = rb[i].y;
    }
  };

  double4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

  const int l15 = lane & 15, l4 = lane >> 4;
  const int a_off = l4 * LDS_LD + wm * 64 + l15;
  const int b_off = BK * LDS_LD + (wn * 64 + l15) * LDB_S + l4;

  // Two LDS stages, ONE barrier per K-step: while the MFMAs of stage `cur` run, the next tile's global
  // loads are in flight; they are written to the other stage right after the MFMAs (every wave finished
  // reading that stage before the previous barrier).
  load_tile(0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = (k0 + BK) < K;
    if (more) load_tile(k0 + BK);
    const double* a_rd = smem + cur * STAGE + a_off;
    const double* b_rd = smem + cur * STAGE + b_off;
    // Chained issue: v_mfma_f64_16x16x4_f64 pays ~40 extra cycles whenever consecutive MFMAs use a
    // different accumulator (C read + D write through the register file); back-to-back MFMAs on the
    // SAME accumulator forward it inside the pipe (measured: 105 -> 74 cycles/MFMA at chain 4,
    // tools/mfma_f64_probe.hip).  So every accumulator takes all BK/4 k-steps of the tile in a row.
    // (Round 2: visiting all 16 accumulators per k-step instead measured the same -- 63.6 vs 64.4 TFLOP/s at 8192^3.)
    {
      double bf[4][BK / 4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) bf[j][ks] = b_rd[(j * 16) * LDB_S + ks * 4];
      double af[BK / 4];
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) af[ks] = a_rd[(ks * 4) * LDS_LD];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double afn[BK / 4];
        if (i < 3) {
#pragma unroll
          for (int ks = 0; ks < BK / 4; ++ks) afn[ks] = a_rd[(ks * 4) * LDS_LD + (i + 1) * 16];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int ks = 0; ks < BK / 4; ++ks)
            // transposed issue: MFMA-A <- B data (n on the register/row axis), MFMA-B <- A data (m on lane&15)
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j][ks], af[ks], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);  // keep the chain together: the scheduler must not interleave accumulators
        }
        if (i < 3) {
#pragma unroll
          for (int ks = 0; ks < BK / 4; ++ks) af[ks] = afn[ks];
        }
      }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: acc[i][j][r] <-> C[m0 + wm*64 + i*16 + l15][n0 + wn*64 + j*16 + l4 + 4r].
  // All 16 loads of a row block are issued before the first store (the compiler cannot prove the C
  // addresses distinct and would otherwise serialise load -> store -> load ...).
  double* __restrict__ C = p.C;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int mm = m0 + wm * 64 + i * 16 + l15;
    const bool rok = mm < M;
    double cv[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n0 + wn * 64 + j * 16 + l4 + 4 * r;
        cv[j][r] = (minus && rok && nn < N) ? gld(C + (size_t)mm + (size_t)nn * p.ldc) : 0.0;
      }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n0 + wn * 64 + j * 16 + l4 + 4 * r;
        if (rok && nn < N) {
          const double v = minus ? (cv[j][r] - acc[i][j][r]) : acc[i][j][r];
          gst(C + (size_t)mm + (size_t)nn * p.ldc, v);
          if (p.flag && mm < p.flag_rows && !(fabs(v) <= HS_GROWTH_MAX)) *p.flag = 1;  // uniform null test; NaN counts
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------
// real double, SKINNY problems (the grouped products of the HSS module: a 64-row window against 2,048 sample columns, 64 x 64 Gram
// matrices over a long inner dimension, ...).  The same pipeline with a 64 x 128 or 64 x 64 tile: a lone workgroup pays 64 MFMAs per wave and
// K-step of the 128 x 128 tile (1.7 us) whether its rows exist or not -- half or three quarters of them multiply padding when M, N <= 64 --
// and these launches are latency chains of K/16 such steps, not throughput.  2 x 2 waves, wave tile (BMs/2) x (BNs/2).
// ------------------------------------------------------------------------------------------------
template <int BMs, int BNs>
__device__ inline void gemm_tile_s_d(const GemmProb<double>& p, int tile_m, int tile_n, bool minus, double* smem) {
  constexpr int STAGE = 2 * BK * LDS_LD;
  constexpr int WM = BMs / 2, WN = BNs / 2, MI = WM / 16, NJ = WN / 16;
  constexpr int NP = BMs / 2, KA = 256 / NP, PA = BK / KA;  // A staging: row pairs, k rows per pass, passes
  constexpr int PB = BNs / 32;                               // B staging passes
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int m0 = tile_m * BMs, n0 = tile_n * BNs;
  const int M = p.M, N = p.N, K = p.K;
  const double* __restrict__ A = p.A;
  const double* __restrict__ B = p.B;
  const int a_pair = tid % NP, a_k = tid / NP;
  const int b_kp = tid & 7, b_n = tid >> 3;
  double2_u ra[PA], rb[PB];
  const bool interior = (m0 + BMs <= M) && (n0 + BNs <= N);
  auto load_tile = [&](int k0) {
    if (interior && k0 + BK <= K) {
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] = gld2(A + (size_t)(m0 + 2 * a_pair) + (size_t)(k0 + a_k + KA * i) * p.lda);
#pragma unroll
      for (int i = 0; i < PB; ++i) rb[i] = gld2(B + (size_t)(k0 + 2 * b_kp) + (size_t)(n0 + b_n + 32 * i) * p.ldb);
    } else {
#pragma unroll
      for (int i = 0; i < PA; ++i) {
        const int kk = k0 + a_k + KA * i, mm = m0 + 2 * a_pair;
        const double* col = A + (size_t)min(kk, K - 1) * p.lda;
        double x = gld(col + min(mm, M - 1)), y = gld(col + min(mm + 1, M - 1));
        const bool kok = kk < K;
        ra[i].x = (kok && mm < M) ? x : 0.0;
        ra[i].y = (kok && mm + 1 < M) ? y : 0.0;
      }
#pragma unroll
      for (int i = 0; i < PB; ++i) {
        const int nn = n0 + b_n + 32 * i, kk = k0 + 2 * b_kp;
        const double* col = B + (size_t)min(nn, N - 1) * p.ldb;
        double x = gld(col + min(kk, K - 1)), y = gld(col + min(kk + 1, K - 1));
        const bool nok = nn < N;
        rb[i].x = (nok && kk < K) ? x : 0.0;
        rb[i].y = (nok && kk + 1 < K) ? y : 0.0;
      }
    }
  };
  auto store_tile = [&](int stage) {
    double* As = smem + stage * STAGE;
    double* Bs = As + BK * LDS_LD;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      double* dst = As + (a_k + KA * i) * LDS_LD + 2 * a_pair;
      dst[0] = ra[i].x;
      dst[1] = ra[i].y;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      double* dst = Bs + (b_n + 32 * i) * LDB_S + 2 * b_kp;
      dst[0] = rb[i].x;
      dst[1] = rb[i].y;
    }
  };
  double4_t acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
  const int l15 = lane & 15, l4 = lane >> 4;
  const int a_off = l4 * LDS_LD + wm * WM + l15;
  const int b_off = BK * LDS_LD + (wn * WN + l15) * LDB_S + l4;
  load_tile(0);
  store_tile(0);
  __syncthreads();
  int cur = 0;
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = (k0 + BK) < K;
    if (more) load_tile(k0 + BK);
    const double* a_rd = smem + cur * STAGE + a_off;
    const double* b_rd = smem + cur * STAGE + b_off;
    {
      double bf[NJ][BK / 4], af[MI][BK / 4];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) bf[j][ks] = b_rd[(j * 16) * LDB_S + ks * 4];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) af[i][ks] = a_rd[(ks * 4) * LDS_LD + i * 16];
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int ks = 0; ks < BK / 4; ++ks) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (more) store_tile(cur ^ 1);
    __syncthreads();
    cur ^= 1;
  }
  double* __restrict__ C = p.C;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int mm = m0 + wm * WM + i * 16 + l15;
    const bool rok = mm < M;
    double cv[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n0 + wn * WN + j * 16 + l4 + 4 * r;
        cv[j][r] = (minus && rok && nn < N) ? gld(C + (size_t)mm + (size_t)nn * p.ldc) : 0.0;
      }
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n0 + wn * WN + j * 16 + l4 + 4 * r;
        if (rok && nn < N) gst(C + (size_t)mm + (size_t)nn * p.ldc, minus ? (cv[j][r] - acc[i][j][r]) : acc[i][j][r]);
      }
  }
}

// ------------------------------------------------------------------------------------------------
// complex double (planar split in LDS)
// ------------------------------------------------------------------------------------------------
#define ZBN 64
#define ZLDB 18  // B images are [n][k], 16 + 2 padding (see LDB_S)
__device__ inline void gemm_tile_z(const GemmProb<cplx>& p, int tile_m, int tile_n, bool minus, double* smem) {
  double* Ar = smem;
  double* Ai = Ar + BK * LDS_LD;
  double* Br = Ai + BK * LDS_LD;
  double* Bi = Br + ZBN * ZLDB;
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

  // branch-free staging (clamped address + select), see gemm_tile_d
  auto load_tile = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int kk = k0 + a_k + 2 * i, mm = m0 + a_m;
      double2_u v = gld2(A + (size_t)min(mm, M - 1) + (size_t)min(kk, K - 1) * p.lda);
      const bool ok = kk < K && mm < M;
      ra[i].x = ok ? v.x : 0.0;
      ra[i].y = ok ? v.y : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = k0 + b_k, nn = n0 + b_n + 16 * i;
      double2_u v = gld2(B + (size_t)min(kk, K - 1) + (size_t)min(nn, N - 1) * p.ldb);
      const bool ok = kk < K && nn < N;
      rb[i].x = ok ? v.x : 0.0;
      rb[i].y = ok ? v.y : 0.0;
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
      int o = (b_n + 16 * i) * ZLDB + b_k;
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
  const int b_off = (wn * 32 + l15) * ZLDB + l4;

  load_tile(0);
  store_tile();
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += BK) {
    const bool more = (k0 + BK) < K;
    if (more) load_tile(k0 + BK);
    {  // chained issue, see gemm_tile_d: each accumulator takes its 2 * BK/4 MFMAs back to back
      double bfr[2][BK / 4], bfi[2][BK / 4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
          bfr[j][ks] = Br[b_off + (j * 16) * ZLDB + ks * 4];
          bfi[j][ks] = Bi[b_off + (j * 16) * ZLDB + ks * 4];
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        double afr[BK / 4], afi[BK / 4];
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
          afr[ks] = Ar[a_off + (ks * 4) * LDS_LD + i * 16];
          afi[ks] = Ai[a_off + (ks * 4) * LDS_LD + i * 16];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int ks = 0; ks < BK / 4; ++ks) {
            accr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[j][ks], afr[ks], accr[i][j], 0, 0, 0);
            accr[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfi[j][ks], afi[ks], accr[i][j], 0, 0, 1);  // blgp bit 0 = NEG(src A) on f64 MFMA: re -= bi*ai
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < BK / 4; ++ks) {
            acci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfi[j][ks], afr[ks], acci[i][j], 0, 0, 0);
            acci[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(bfr[j][ks], afi[ks], acci[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
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
    const int mm = m0 + wm * 64 + i * 16 + l15;
    const bool rok = mm < M;
    double2_u cv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n0 + wn * 32 + j * 16 + l4 + 4 * r;
        double2_u o = {0.0, 0.0};
        if (minus && rok && nn < N) o = gld2(C + (size_t)mm + (size_t)nn * p.ldc);
        cv[j][r] = o;
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nn = n0 + wn * 32 + j * 16 + l4 + 4 * r;
        if (rok && nn < N) {
          double2_u v = {accr[i][j][r], acci[i][j][r]};
          if (minus) {
            v.x = cv[j][r].x - v.x;
            v.y = cv[j][r].y - v.y;
          }
          gst2(C + (size_t)mm + (size_t)nn * p.ldc, v);
          if (p.flag && mm < p.flag_rows && !(fabs(v.x) + fabs(v.y) <= HS_GROWTH_MAX)) *p.flag = 1;  // uniform null test; NaN counts (GemmOp::ainv 16..19: the values ARE multipliers)
        }
      }
  }
}

template <class T>
struct TileCfg;
template <>
struct TileCfg<double> {
  static constexpr int bn = 128;
  static constexpr int smem_doubles = 2 * (2 * BK * LDS_LD);  // two stages
};
template <>
struct TileCfg<cplx> {
  static constexpr int bn = ZBN;
  static constexpr int smem_doubles = 2 * BK * LDS_LD + 2 * ZBN * ZLDB;
};



template <class T>
__device__ inline void gemm_dispatch(const GemmProb<T>& p, bool minus, double* smem) {
  int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + TileCfg<T>::bn - 1) / TileCfg<T>::bn;
  int ntiles = tiles_m * tiles_n;
  // a launch may be capped to fewer workgroups than tiles (GemmOp::cap): each workgroup then walks the tiles
  // bid, bid + gridDim.x, ... -- with gridDim.x a multiple of 8 they all stay in the same XCD chunk of the remap
  const int shift = (int)((gridDim.x * blockIdx.y) & 7u);
  for (int bid = blockIdx.x; bid < ntiles; bid += gridDim.x) {
    int t = xcd_remap_shift(bid, ntiles, (gridDim.x & 7u) ? shift : 0);  // (a walking workgroup keeps its XCD only when gridDim.x is a multiple of 8: then shift == 0)
    int tm, tn;
    tile_coords(t, tiles_m, tiles_n, tm, tn);
    if constexpr (sizeof(T) == 8)
      gemm_tile_d<128>(p, tm, tn, minus, smem);
    else
      gemm_tile_z(p, tm, tn, minus, smem);
    if (bid + (int)gridDim.x < ntiles) __syncthreads();  // the next tile re-uses the LDS stages
  }
}

template <class T>
__global__ __launch_bounds__(256, 2) void gemm_op_kernel(const NodeDesc<T>* __restrict__ nodes, GemmOp op) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  GemmProb<T> p;
  if (op.prio) __builtin_amdgcn_s_setprio(2);  // look-ahead panel work sharing CUs with the big trailing update
  if (!resolve_op(nodes + blockIdx.y, op, p)) return;
  gemm_dispatch<T>(p, true, smem);
}
// The TRSM base case (X <- inv(diagonal block) * X, in place) runs the same tile code under its own name, so that
// profiles separate the trailing updates (gemm_op_kernel: the flops) from the 32-row solves (latency).
template <class T>
__global__ __launch_bounds__(256, 2) void trsm_inv_kernel(const NodeDesc<T>* __restrict__ nodes, GemmOp op) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  GemmProb<T> p;
  __builtin_amdgcn_s_setprio(2);
  if (!resolve_op(nodes + blockIdx.y, op, p)) return;
  gemm_dispatch<T>(p, false, smem);
}

// Accounting of the grouped products (hs_probs_stats, include/hs_kernels.h): the descriptors live on the device, so the kernel itself adds
// the real flops of every problem (one atomic per problem and launch, only while the accounting is on: bit 1 of `minus`)
__device__ double g_probs_flops;
template <class T>
__global__ __launch_bounds__(256, 2) void gemm_probs_kernel(const GemmProb<T>* __restrict__ probs, int minus) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  GemmProb<T> p = probs[blockIdx.y];
  if (p.M <= 0 || p.N <= 0) return;
  if ((minus & 2) && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_probs_flops, (sizeof(T) == 16 ? 8.0 : 2.0) * p.M * (double)p.N * p.K);
  gemm_dispatch<T>(p, (minus & 1) != 0, smem);
}
// Problem lists that cannot fill the chip with 128 x 128 tiles (launch_gemm_probs decides): what such a launch costs is the LENGTH of a
// tile's K loop (1.7 us per 16 columns for a lone workgroup), so the work is cut into 64 x 64 tiles -- a quarter of the MFMAs per K-step,
// four times the workgroups -- and 64 x 128 for a problem with M <= 64 < N.
__global__ __launch_bounds__(256, 2) void gemm_probs_skinny_kernel(const GemmProb<double>* __restrict__ probs, int minus) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  GemmProb<double> p = probs[blockIdx.y];
  if (p.M <= 0 || p.N <= 0) return;
  if ((minus & 2) && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&g_probs_flops, 2.0 * p.M * (double)p.N * p.K);
  minus &= 1;
  if (p.M <= 64 && p.N > 64) {
    const int tn_ = (p.N + 127) / 128;
    for (int bid = blockIdx.x; bid < tn_; bid += gridDim.x) {
      gemm_tile_s_d<64, 128>(p, 0, bid, minus != 0, smem);
      if (bid + (int)gridDim.x < tn_) __syncthreads();
    }
    return;
  }
  const int tm_ = (p.M + 63) / 64, tn_ = (p.N + 63) / 64, nt = tm_ * tn_;
  for (int bid = blockIdx.x; bid < nt; bid += gridDim.x) {
    gemm_tile_s_d<64, 64>(p, bid % tm_, bid / tm_, minus != 0, smem);
    if (bid + (int)gridDim.x < nt) __syncthreads();
  }
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
  if (op.cap > 0 && tiles > op.cap) tiles = std::max(8, op.cap / 8 * 8);
  constexpr int lds_bytes = TileCfg<T>::smem_doubles * 8;
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KiB of LDS per workgroup needs the opt-in
    (void)hipFuncSetAttribute((const void*)gemm_op_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void*)trsm_inv_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  if (op.ainv)
    hipLaunchKernelGGL(trsm_inv_kernel<T>, dim3(tiles, nbatch), dim3(256), lds_bytes, s, dnodes, op);
  else
    hipLaunchKernelGGL(gemm_op_kernel<T>, dim3(tiles, nbatch), dim3(256), lds_bytes, s, dnodes, op);
}

// hs_probs_stats: flops (counted by the kernels), launches and -- with timing on -- the summed launch durations of the grouped products,
// process-wide (they are issued from the HSS module, the low-rank compressions and the matrix-free fronts, on several streams)
namespace {
struct ProbsAcc {
  std::mutex mu;
  int mode = 0;  // 0: off, 1: flops + launches, 2: + a HIP event pair around every launch
  long long launches = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
ProbsAcc& probs_acc() {
  static ProbsAcc* a = new ProbsAcc();
  return *a;
}
}  // namespace
extern "C" int hs_probs_stats_mode(int mode) {  // resets the counters
  ProbsAcc& a = probs_acc();
  std::lock_guard<std::mutex> lk(a.mu);
  (void)hipDeviceSynchronize();
  for (auto& e : a.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  a.ev.clear();
  a.launches = 0;
  a.mode = mode;
  const double z = 0.0;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_probs_flops), &z, sizeof z) == hipSuccess ? 0 : -6;
}
extern "C" int hs_probs_stats(double* out3) {  // {real flops executed, launches, summed launch seconds (0 without timing)}
  if (!out3) return -1;
  ProbsAcc& a = probs_acc();
  std::lock_guard<std::mutex> lk(a.mu);
  if (hipDeviceSynchronize() != hipSuccess) return -6;
  double fl = 0.0, ms = 0.0;
  if (hipMemcpyFromSymbol(&fl, HIP_SYMBOL(g_probs_flops), sizeof fl) != hipSuccess) return -6;
  for (auto& e : a.ev) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, e.first, e.second) == hipSuccess) ms += t;
  }
  out3[0] = fl; out3[1] = (double)a.launches; out3[2] = ms * 1e-3;
  return 0;
}

template <class T>
void launch_gemm_probs(const GemmProb<T>* dprobs, int nprob, int maxM, int maxN, int minus, hipStream_t s) {
  if (nprob <= 0 || maxM <= 0 || maxN <= 0) return;
  minus = minus ? 1 : 0;
  hipEvent_t pe0 = nullptr, pe1 = nullptr;
  {
    ProbsAcc& a = probs_acc();
    if (a.mode) {  // (read without the lock: switched only between factorizations)
      std::lock_guard<std::mutex> lk(a.mu);
      a.launches++;
      minus |= 2;
      if (a.mode >= 2 && hipEventCreate(&pe0) == hipSuccess && hipEventCreate(&pe1) == hipSuccess) {
        (void)hipEventRecord(pe0, s);
        a.ev.push_back({pe0, pe1});
      } else {
        pe1 = nullptr;
      }
    }
  }
  struct Rec {  // the closing event is recorded on every exit path
    hipEvent_t e; hipStream_t s;
    ~Rec() { if (e) (void)hipEventRecord(e, s); }
  } rec{pe1, s};
  int tiles = ((maxM + BM - 1) / BM) * ((maxN + TileCfg<T>::bn - 1) / TileCfg<T>::bn);
  constexpr int lds_bytes = TileCfg<T>::smem_doubles * 8;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_probs_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  if constexpr (sizeof(T) == 8) {
    static const bool skinny = !(getenv("HS_GEMM_SKINNY") && getenv("HS_GEMM_SKINNY")[0] == '0');
    static const int small_tiles = getenv("HS_GEMM_SMALL_TILES") ? atoi(getenv("HS_GEMM_SMALL_TILES")) : 128;  // lists of at most this many 128 x 128 tiles
    if (skinny && (maxM <= 64 || (long long)tiles * nprob <= small_tiles)) {
      if (maxM > 64) tiles = ((maxM + 63) / 64) * ((maxN + 63) / 64);
      static bool attr_set_s = false;
      if (!attr_set_s) {
        (void)hipFuncSetAttribute((const void*)gemm_probs_skinny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        attr_set_s = true;
      }
      hipLaunchKernelGGL(gemm_probs_skinny_kernel, dim3(tiles, nprob), dim3(256), lds_bytes, s, (const GemmProb<double>*)dprobs, minus);
      return;
    }
  }
  hipLaunchKernelGGL(gemm_probs_kernel<T>, dim3(tiles, nprob), dim3(256), lds_bytes, s, dprobs, minus);
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

// The same loop on RANDOM operands that change every iteration: under such data the chip holds its clock below the 2.4 GHz the datasheet
// peak assumes (MI355X_MICROARCH.md, "DVFS give-back"), so this is the FP64 MFMA rate a real GEMM can approach.
__global__ __launch_bounds__(256) void mfma_f64_rate_random_kernel(double* out, int iters, unsigned long long seed) {
  double4_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (double4_t){0.0, 0.0, 0.0, 0.0};
  unsigned long long x = seed + (unsigned long long)(blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull;
  auto next = [&]() {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    return (double)(long long)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  };
  double a[4], b[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] = next(); b[q] = next(); }
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[(u + i) & 3], b[(u + 2 * i + (i >> 1)) & 3], acc[i], 0, 0, 0);
    if ((it & 1023) == 1020) {  // refresh the operands now and then (outside the hot issue pattern)
#pragma unroll
      for (int q = 0; q < 4; ++q) { a[q] = next(); b[q] = next(); }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
extern "C" double hsk_mfma_f64_peak_random(int waves_per_simd, int iters) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1.0;
  const int blocks = prop.multiProcessorCount * waves_per_simd;
  double* out = nullptr;
  if (hipMalloc(&out, sizeof(double) * blocks * 256) != hipSuccess) return -1.0;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_f64_rate_random_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 12345ull);  // warm-up: the clock settles under load
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL(mfma_f64_rate_random_kernel, dim3(blocks), dim3(256), 0, 0, out, iters, 67890ull);
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
  return flops / (ms * 1e-3) / 1e12;
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
