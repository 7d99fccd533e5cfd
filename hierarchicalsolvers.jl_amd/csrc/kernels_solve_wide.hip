// kernels_solve_wide.hip -- the triangular sweeps of ldiv!: inverses of the 256 x 256 diagonal blocks (inv256_kernel), the sweeps advancing 256
// columns per launch (fwd_wide / bwd_wide: HS_SOLVE_FLOW=0), and -- the default since round 3 -- the sweeps of a whole tree level as ONE
// dataflow launch (flow_sweep_kernel, second half of the file; DESIGN.md section 4a').
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
// A workgroup works on NC of the 32 columns of a block column: 32 for Float64 (72 KB of LDS), 16 for ComplexF64 -- with all 32 a ComplexF64
// workgroup needed 144 KB, more than ONE retiring GEMM workgroup frees on a CU (54 KB + the 52 KB two of them leave), so next to a running
// trailing update every workgroup waited for BOTH GEMM workgroups of some CU to retire and then had the CU to itself (1.1 - 5.5 ms per call in
// the profiles of the complex workloads); with 72 KB it fits beside one.
template <class T>
struct Inv256Cfg {
  static constexpr int NC = sizeof(T) == 8 ? HS_PB : HS_PB / 2;  // columns per workgroup
  static constexpr int H = HS_PB / NC;                            // workgroups per block column
  static constexpr int BS = HS_PB * NC;                           // elements of one X block
  static constexpr int NU = NC / 8;                               // outputs per thread (row t & 31, columns (t >> 5) + 8 u)
};
template <class T>
__global__ __launch_bounds__(256) void inv256_kernel(const SolveNode<T>* __restrict__ nodes, int first_block) {
  constexpr int NC = Inv256Cfg<T>::NC, H = Inv256Cfg<T>::H, BS = Inv256Cfg<T>::BS, NU = Inv256Cfg<T>::NU;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* X = reinterpret_cast<T*>(smem_raw);   // 8 blocks of 32 x NC (column-major, ld 32): X_kj of the current block column
  T* S = X + 8 * BS;                       // 32 x NC accumulator
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int upper = blockIdx.z;
  const int sub = blockIdx.x % H, bx = blockIdx.x / H;
  const int b256 = first_block + (bx >> 3), jc = bx & 7;
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
  for (int e = t; e < BS; e += 256) X[jc * BS + e] = inv32[(size_t)(sb0 + jc) * 1024 + sub * BS + e];
  __syncthreads();
  const int istep = upper ? -1 : 1;
  for (int i = jc + istep; i >= 0 && i < nsb; i += istep) {
    // S = sum_k A_ik * X_kj over the already known blocks k between j and i
    T acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) acc[u] = Scal<T>::zero();
    const int k0 = upper ? i + 1 : jc, k1 = upper ? jc : i - 1;
    // A_ik straight from global memory (round 3): thread t owns ROW a = t & 31 of the block and four of its columns (256 is a multiple of 32), so
    // one load of A_ik[a, q] -- coalesced over the 32 rows -- feeds four outputs, and the loads of all k are independent: no staging through
    // LDS and no barrier inside the k loop.  Before, every (i, k) pair staged its block and cost two workgroup barriers and a dependent
    // memory round trip: 28 of them in a row per block column (233 us per call for Float64, 1.4 ms for ComplexF64 next to a GEMM -- on the panel
    // chain of every 256-column group).
    {
      const int a = t & 31, wi = min(HS_PB, wl - i * HS_PB);
      for (int k = k0; k <= k1; ++k) {
        const T* Aik = nd.LF + (size_t)(c0 + i * HS_PB) + (size_t)(c0 + k * HS_PB) * nd.ldl;
        const int wk = min(HS_PB, wl - k * HS_PB);
        const T* Xk = X + k * BS;
        const T* arow = Aik + (size_t)min(a, wi - 1);  // (clamped: rows past the block's extent multiply by zero below)
        const bool rowok = a < wi;
        // Float64: all 32 loads of the row before the first use (a register array, fully unrolled) -- in a loop the compiler issued them in
        // dependent groups, the finding of the dataflow sweeps below (DESIGN.md section 4a'): 284 -> 231 us per call next to the GEMM.
        // ComplexF64 keeps the loop: batched it measured 2.7 ms instead of 1.1 (its 144 KB of LDS make it wait for both GEMM workgroups of a CU anyway)
        if constexpr (sizeof(T) == 8) {
          T av[HS_PB];
#pragma unroll
          for (int q = 0; q < HS_PB; ++q) av[q] = gld(arow + (size_t)min(q, wk - 1) * nd.ldl);
#pragma unroll
          for (int q = 0; q < HS_PB; ++q) {
            const T a_ = (rowok && q < wk) ? av[q] : Scal<T>::zero();
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = Scal<T>::fma(a_, Xk[q + ((t >> 5) + 8 * u) * HS_PB], acc[u]);
          }
        } else {
#pragma unroll 8
          for (int q = 0; q < HS_PB; ++q) {
            T a_ = gld(arow + (size_t)min(q, wk - 1) * nd.ldl);
            if (!(rowok && q < wk)) a_ = Scal<T>::zero();
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = Scal<T>::fma(a_, Xk[q + ((t >> 5) + 8 * u) * HS_PB], acc[u]);
          }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) S[t + 256 * u] = acc[u];
    __syncthreads();
    // X_ij = -inv32_i * S
    const T* Ii = inv32 + (size_t)(sb0 + i) * 1024;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = t + 256 * u, a = e & 31, b = e >> 5;
      T s = Scal<T>::zero();
      for (int q = 0; q < HS_PB; ++q) s = Scal<T>::fma(Ii[a + q * HS_PB], S[q + b * HS_PB], s);
      X[i * BS + e] = Scal<T>::zero() - s;
    }
    __syncthreads();
  }
  // write the WHOLE block column (256 rows x 32 columns): the blocks of the triangle, zeros elsewhere -- the TRSM base
  // case multiplies by the full 256 x 256 block (GemmOp::ainv 3..6)
  const int ilo = upper ? 0 : jc, ihi = upper ? jc : nsb - 1;
  for (int i = 0; i < HS_SW / HS_PB; ++i)
    for (int e = t; e < BS; e += 256) {
      const int a = e & 31, b = e >> 5;
      out[(size_t)(i * HS_PB + a) + (size_t)(jc * HS_PB + sub * NC + b) * HS_SW] = (i >= ilo && i <= ihi) ? X[i * BS + e] : Scal<T>::zero();
    }
}

// The sweeps below are chains of dependent launches (ni / 256 per front and sweep), so what matters for the large fronts is the latency of
// one step, i.e. the number of dependent memory round trips in it (a step costs the same 22 us for 2 and for 127 workgroups).
//  * A workgroup is 1024 threads over 256 rows.  Float64: a thread owns TWO adjacent rows and 32 columns, read as 32 16-byte loads that are
//    all in flight at once -- one round trip for the 256 x 256 block; ComplexF64: one row and 64 columns.  The partial sums of a row meet
//    in LDS.  (256 threads with one row each, loads dependent by eight: 74 us per forward step at Poisson 128^3.)
//  * The product with the stored inverse of the diagonal block is taken OFF the chain of all workgroups but one: the workgroup whose rows
//    are the next block of the sweep finishes them, multiplies them by that block's inverse and leaves y (x) of the next step behind; only
//    the first step of a front computes its own.
template <class T>
struct WideCfg {
  static constexpr int RP = sizeof(T) == 8 ? 2 : 1;  // rows per thread
  static constexpr int NG = 1024 / (HS_SW / RP);     // column groups
  static constexpr int GC = HS_SW / NG;              // columns per group
};

// s[0..RP) += sum_{j in [jlo, jhi)} A[row + {0..RP}, j] * sv[j]; `a` points at (row, 0); nrows = how many of the thread's rows exist
template <class T>
__device__ __forceinline__ void wide_dot(const T* a, size_t ld, int jlo, int jhi, const T* sv, int nrows, T* s) {
  if constexpr (WideCfg<T>::RP == 2) {
    if (nrows == 2) {
#pragma unroll 16
      for (int j = jlo; j < jhi; ++j) {
        const hs_d2u v = gld2(a + (size_t)j * ld);
        const T y = sv[j];
        s[0] = Scal<T>::fma(v.x, y, s[0]);
        s[1] = Scal<T>::fma(v.y, y, s[1]);
      }
      return;
    }
  }
  if (nrows >= 1) {
#pragma unroll 32
    for (int j = jlo; j < jhi; ++j) s[0] = Scal<T>::fma(gld(a + (size_t)j * ld), sv[j], s[0]);
  }
}
// thread coordinates: first row of the thread, first column of its group
#define WIDE_COORDS                                                     \
  constexpr int RP = WideCfg<T>::RP, NG = WideCfg<T>::NG, GC = WideCfg<T>::GC; \
  const int t = threadIdx.x;                                            \
  const int row = (t % (HS_SW / RP)) * RP, jg = (t / (HS_SW / RP)) * GC /* wave-uniform */
template <class T>
__device__ __forceinline__ void wide_put(T* s_red, int t, const T* s) {
  constexpr int RP = WideCfg<T>::RP;
  const int row = (t % (HS_SW / RP)) * RP, g = t / (HS_SW / RP);
#pragma unroll
  for (int q = 0; q < RP; ++q) s_red[g * HS_SW + row + q] = s[q];
}
template <class T>
__device__ __forceinline__ T wide_sum(const T* s_red, int t) {
  T v = s_red[t];
#pragma unroll
  for (int g = 1; g < WideCfg<T>::NG; ++g) v = v + s_red[g * HS_SW + t];
  return v;
}
// partial sums of inv * s_w for the thread's rows over the columns of its group (whole 32-blocks of the triangle)
template <class T, bool LOWER>
__device__ __forceinline__ void wide_inv_partial(const T* iv, int row, int jg, int wl, const T* s_w, T* s) {
  constexpr int RP = WideCfg<T>::RP, GC = WideCfg<T>::GC;
  const int jlo = LOWER ? jg : max(row / HS_PB * HS_PB, jg);
  const int jhi = LOWER ? min(min(wl, (row / HS_PB + 1) * HS_PB), jg + GC) : min(wl, jg + GC);
  wide_dot<T>(iv + row, HS_SW, jlo, jhi, s_w, min(RP, wl - row), s);
}

// forward: y_blk = L[blk,blk]^-1 * w_blk ; rows below -= L[:, blk] * y_blk   (rows >= ni are the Abi*U^-1 rows: they update rhs[bnd])
template <class T>
__global__ __launch_bounds__(1024) void fwd_wide_kernel(const SolveNode<T>* __restrict__ nodes, int blk, T* __restrict__ w, T* __restrict__ y,
                                                        T* __restrict__ b, int first_done) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int c0 = blk * HS_SW;
  if (c0 >= nd.ni) return;
  const int wl = min(HS_SW, nd.ni - c0);
  const int r0 = c0 + wl;
  if (blockIdx.x > 0 && (int)blockIdx.x * HS_SW >= nd.mrows - r0) return;
  __shared__ T s_w[HS_SW];
  __shared__ T s_y[HS_SW];
  __shared__ T s_red[HS_SW * WideCfg<T>::NG];
  WIDE_COORDS;
  const int rbase = r0 + blockIdx.x * HS_SW;
  const int r = rbase + t;  // the row thread t < 256 finishes
  const bool mine = t < HS_SW && r < nd.mrows;
  T wold = Scal<T>::zero();
  int gi = 0;
  if (mine) {  // issued ahead of the panel loads
    if (r < nd.ni) {
      wold = w[nd.woff + r];
    } else {
      gi = gld(nd.fidx + r);
      wold = b[gi];
    }
  }
  if (blk > 0 || first_done) {  // left behind by the previous step (block 0: by wide_first_kernel)
    if (t < HS_SW) s_y[t] = (t < wl) ? y[nd.woff + c0 + t] : Scal<T>::zero();
  } else {
    if (t < HS_SW) s_w[t] = (t < wl) ? w[nd.woff + c0 + t] : Scal<T>::zero();
    __syncthreads();
    T s[RP] = {};
    wide_inv_partial<T, true>(nd.inv256L + (size_t)blk * HS_SW * HS_SW, row, jg, wl, s_w, s);
    wide_put<T>(s_red, t, s);
    __syncthreads();
    if (t < HS_SW) {
      const T v = wide_sum<T>(s_red, t);
      s_y[t] = v;
      if (blockIdx.x == 0 && t < wl) y[nd.woff + c0 + t] = v;
    }
    __syncthreads();  // s_red is written again below
  }
  __syncthreads();
  {
    T s[RP] = {};
    wide_dot<T>(nd.LF + (size_t)(rbase + row) + (size_t)c0 * nd.ldl, nd.ldl, jg, min(jg + GC, wl), s_y, min(RP, nd.mrows - (rbase + row)), s);
    wide_put<T>(s_red, t, s);
  }
  __syncthreads();
  T wnew = Scal<T>::zero();
  if (mine) {
    wnew = wold - wide_sum<T>(s_red, t);
    if (r < nd.ni)
      w[nd.woff + r] = wnew;
    else
      b[gi] = wnew;
  }
  // the rows of this workgroup are the next block of the sweep: finish its step's product with the stored inverse
  if (blockIdx.x != 0 || r0 >= nd.ni) return;
  const int wl2 = min(HS_SW, nd.ni - r0);
  __syncthreads();  // every thread has read s_red
  if (t < HS_SW) s_w[t] = (t < wl2) ? wnew : Scal<T>::zero();
  __syncthreads();
  {
    T s[RP] = {};
    wide_inv_partial<T, true>(nd.inv256L + (size_t)(blk + 1) * HS_SW * HS_SW, row, jg, wl2, s_w, s);
    wide_put<T>(s_red, t, s);
  }
  __syncthreads();
  if (t < wl2) y[nd.woff + r0 + t] = wide_sum<T>(s_red, t);
}

// backward: x_blk = U[blk,blk]^-1 * w_blk ; rows above -= U[:, blk] * x_blk
template <class T>
__global__ __launch_bounds__(1024) void bwd_wide_kernel(const SolveNode<T>* __restrict__ nodes, int blk, T* __restrict__ w, T* __restrict__ x, int first_done) {
  const SolveNode<T> nd = nodes[blockIdx.y];
  const int c0 = blk * HS_SW;
  if (c0 >= nd.ni) return;
  const int wl = min(HS_SW, nd.ni - c0);
  if (blockIdx.x > 0 && (int)blockIdx.x * HS_SW >= c0) return;
  const bool have_x = c0 + HS_SW < nd.ni || first_done;  // not the first step of this front's sweep: the step before left x_blk behind (the first: wide_first_kernel)
  if (have_x && c0 == 0) return;
  __shared__ T s_w[HS_SW];
  __shared__ T s_x[HS_SW];
  __shared__ T s_red[HS_SW * WideCfg<T>::NG];
  WIDE_COORDS;
  const int rbase = blockIdx.x * HS_SW;
  const int r = rbase + t;
  const bool mine = t < HS_SW && r < c0;
  T wold = Scal<T>::zero();
  if (mine) wold = w[nd.woff + r];
  if (have_x) {
    if (t < HS_SW) s_x[t] = (t < wl) ? x[nd.woff + c0 + t] : Scal<T>::zero();
  } else {
    if (t < HS_SW) s_w[t] = (t < wl) ? w[nd.woff + c0 + t] : Scal<T>::zero();
    __syncthreads();
    T s[RP] = {};
    wide_inv_partial<T, false>(nd.inv256U + (size_t)blk * HS_SW * HS_SW, row, jg, wl, s_w, s);
    wide_put<T>(s_red, t, s);
    __syncthreads();
    if (t < HS_SW) {
      const T v = wide_sum<T>(s_red, t);
      s_x[t] = v;
      if (blockIdx.x == 0 && t < wl) x[nd.woff + c0 + t] = v;
    }
    __syncthreads();
  }
  __syncthreads();
  {
    T s[RP] = {};
    wide_dot<T>(nd.LF + (size_t)(rbase + row) + (size_t)c0 * nd.ldl, nd.ldl, jg, min(jg + GC, wl), s_x, min(RP, c0 - (rbase + row)), s);
    wide_put<T>(s_red, t, s);
  }
  __syncthreads();
  T wnew = Scal<T>::zero();
  if (mine) {
    wnew = wold - wide_sum<T>(s_red, t);
    w[nd.woff + r] = wnew;
  }
  // rows [c0 - 256, c0) are the next block of the sweep
  if (rbase + HS_SW != c0) return;
  __syncthreads();
  if (t < HS_SW) s_w[t] = wnew;
  __syncthreads();
  {
    T s[RP] = {};
    wide_inv_partial<T, false>(nd.inv256U + (size_t)(blk - 1) * HS_SW * HS_SW, row, jg, HS_SW, s_w, s);
    wide_put<T>(s_red, t, s);
  }
  __syncthreads();
  if (t < HS_SW) x[nd.woff + c0 - HS_SW + t] = wide_sum<T>(s_red, t);
}

// the product of the FIRST block of a sweep with its stored inverse, once per front (forward: block 0, backward: the last block) -- inside the sweep
// kernels every workgroup of the step would compute it again (16 workgroups per leaf front of Poisson 128^3)
template <class T, bool LOWER>
__global__ __launch_bounds__(1024) void wide_first_kernel(const SolveNode<T>* __restrict__ nodes, const T* __restrict__ w, T* __restrict__ out) {
  const SolveNode<T> nd = nodes[blockIdx.x];
  if (nd.ni <= 0) return;
  const int blk = LOWER ? 0 : (nd.ni - 1) / HS_SW;
  const int c0 = blk * HS_SW, wl = min(HS_SW, nd.ni - c0);
  __shared__ T s_w[HS_SW];
  __shared__ T s_red[HS_SW * WideCfg<T>::NG];
  WIDE_COORDS;
  (void)NG;
  if (t < HS_SW) s_w[t] = (t < wl) ? w[nd.woff + c0 + t] : Scal<T>::zero();
  __syncthreads();
  T s[RP] = {};
  wide_inv_partial<T, LOWER>((LOWER ? nd.inv256L : nd.inv256U) + (size_t)blk * HS_SW * HS_SW, row, jg, wl, s_w, s);
  wide_put<T>(s_red, t, s);
  __syncthreads();
  if (t < wl) out[nd.woff + c0 + t] = wide_sum<T>(s_red, t);
}

// ------------------------------------------------------------------------------------------------
// Dataflow sweeps (round 3): ONE launch per tree level and sweep instead of ni/256 dependent launches per front.
//
// The launch-per-step sweeps above cost 18-24 us per 256 columns whatever the panel size (307 steps per sweep at Poisson 128^3): a step is a
// chain of two dependent 256 x 256 tile products by ONE workgroup between memory round trips.  Here
//  * a workgroup of 512 threads OWNS 64 rows (ComplexF64: 32) of a front for the whole sweep: it accumulates sum_k L[rows, blk k] * y_k in
//    registers as the y_k arrive, so a tile is 128 KB, and four (eight) workgroups share what one did per step;
//  * the tiles do not depend on what is waited for, so the loads of the NEXT tile -- at the end: of the slab of the stored inverse -- are
//    issued before this round's vector is polled (two register tiles, flow_load / flow_fma): on the chain a round is exchange + FMAs +
//    one LDS reduction, the memory round trip of its tile is already behind it;
//  * values are exchanged through two vectors that hold a SENTINEL (all bits set: no arithmetic produces that NaN) until the value is
//    published with an agent-scope atomic store; a consumer polls the values themselves with agent-scope atomic loads -- no flag, no
//    fence.  Measured (hsk_flow_pingpong_us): 0.5 us per exchange between any two workgroups of the chip, same XCD or not.  E1 carries
//    y (x in the backward sweep), E2 the finished w of a diagonal block: the rows q*64.. of y_j = inv256_j * w_j need w of the sub-blocks
//    <= q (>= q backward), so a 256-step is tile -> w -> inverse slab -> y;
//  * forward progress: a workgroup waits only for sub-blocks of the SAME front that come earlier in the sweep, and workgroup ids are
//    handed out by an atomic counter in the order the workgroups actually start (block-major, front-minor: all fronts of a level advance
//    together), so everything a workgroup waits for has started.  Every poll is bounded (HS_FLOW_SPIN rounds of s_sleep, seconds): a
//    workgroup that runs out raises *err and leaves without publishing, its dependents run out in turn, the grid drains, and the host
//    reports it at the next call instead of the GPU hanging.
// ------------------------------------------------------------------------------------------------
#define HS_FLOW_SPIN (1 << 22)
static constexpr unsigned long long HS_SENT = ~0ull;
// Thread layout of a tile (512 threads = 8 waves): the LANES run along the rows and a WAVE owns 32 consecutive columns, so the column of a
// load is wave-uniform -- its address is a scalar base plus one per-lane byte offset, and 32 (16) loads in flight cost their data registers
// only.  Float64: 64 rows, one 8-byte load per lane and column.  ComplexF64: 32 rows, the two half-waves take the two columns of a pair.
template <class T>
struct FlowCfg {
#ifndef HS_FLOW_NT
#define HS_FLOW_NT 512
#endif
  static constexpr int NT = HS_FLOW_NT;                 // threads per workgroup
  static constexpr int FB = (sizeof(T) == 8 ? 64 : 32) * NT / 512;  // rows a workgroup owns: 64 data registers per thread and tile either way
  static constexpr int SC = 64 / FB;                    // columns a wave covers per load
  static constexpr int CW = HS_SW / (NT / 64);          // columns a wave owns
  static constexpr int GC = CW / SC;                    // loads per thread and tile: 64 data registers
  static constexpr int NP = (NT / 64) * SC;             // partial sums per row
  static constexpr int Q = HS_SW / FB;                  // sub-blocks per 256-block
};
__device__ __forceinline__ unsigned long long flow_ldbits(const double* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flow_publish(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flow_publish(cplx* p, cplx v) {
  flow_publish(reinterpret_cast<double*>(p), v.re);
  flow_publish(reinterpret_cast<double*>(p) + 1, v.im);
}
// thread t < 256 fetches element t of a published vector (cnt <= 256 entries) into dst[t], zero beyond cnt; false when the wait ran out
__device__ __forceinline__ bool flow_poll(const double* src, int cnt, double* dst, int t) {
  if (t >= HS_SW) return true;
  if (t >= cnt) {
    dst[t] = 0.0;
    return true;
  }
  for (int it = 0; it < HS_FLOW_SPIN; ++it) {
    const unsigned long long v = flow_ldbits(src + t);
    if (v != HS_SENT) {
      dst[t] = __longlong_as_double((long long)v);
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
__device__ __forceinline__ bool flow_poll(const cplx* src, int cnt, cplx* dst, int t) {
  if (t >= HS_SW) return true;
  if (t >= cnt) {
    dst[t] = cplx{0.0, 0.0};
    return true;
  }
  const double* q = reinterpret_cast<const double*>(src + t);
  for (int it = 0; it < HS_FLOW_SPIN; ++it) {
    const unsigned long long a = flow_ldbits(q), b = flow_ldbits(q + 1);
    if (a != HS_SENT && b != HS_SENT) {
      dst[t] = cplx{__longlong_as_double((long long)a), __longlong_as_double((long long)b)};
      return true;
    }
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}
// The thread's share of a tile, in registers: every load issued before anything uses it and none depending on the data the workgroup waits
// for -- so the tile of the NEXT round is requested before this round's vector is polled (the loop below keeps two of them).  Measured on
// the way here: with the loads in a loop the compiler issued them in small dependent groups and a 64 x 256 tile cost ~8 us on an idle chip;
// as one batch ~2.5 us.  Addresses are clamped instead of guarded (rows past rl, columns past `ncol`): what they bring in is multiplied by
// a zero of the vector or lands in a row nobody stores.  `base` is workgroup-uniform.
// (buffer loads: the tile base goes into a 128-bit descriptor held in SGPRs, the lane's part of the address is ONE 32-bit byte offset and the
// column's a scalar offset -- with 64-bit flat addresses the 32 loads in flight of two tiles took 128 address registers and the kernel spilled)
typedef unsigned int flow_u2 __attribute__((ext_vector_type(2)));
typedef unsigned int flow_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double flow_bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const double*) {
  const flow_u2 x = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
  return __longlong_as_double((long long)(((unsigned long long)x.y << 32) | x.x));
}
__device__ __forceinline__ cplx flow_bld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, const cplx*) {
  const flow_u4 x = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  return cplx{__longlong_as_double((long long)(((unsigned long long)x.y << 32) | x.x)), __longlong_as_double((long long)(((unsigned long long)x.w << 32) | x.z))};
}
template <class T>
__device__ __forceinline__ void flow_load(T (&v)[FlowCfg<T>::GC], const T* base, unsigned ld, int ncol, int rl, int t) {
  constexpr int FB = FlowCfg<T>::FB, SC = FlowCfg<T>::SC, GC = FlowCfg<T>::GC;
  const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6);
  const unsigned row = (unsigned)min(lane % FB, rl - 1);
  // (the descriptor must be PROVABLY uniform or every load becomes a waterfall loop: readfirstlane its inputs)
  const unsigned long long pb = (unsigned long long)base;
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pb), phi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
  ld = __builtin_amdgcn_readfirstlane(ld);
  ncol = __builtin_amdgcn_readfirstlane(ncol);
  const __amdgpu_buffer_rsrc_t r =
      __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(((unsigned long long)phi << 32) | plo), 0, 0x7FFFFFFF, 0x00020000);  // (256 columns of a front: < 2^31 bytes)
  if constexpr (SC == 1) {
    const unsigned voff = row * (unsigned)sizeof(T);
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      const unsigned col = min((unsigned)(wv * FlowCfg<T>::CW + j), (unsigned)(ncol - 1));  // scalar
      v[j] = flow_bld(r, voff, col * ld * (unsigned)sizeof(T), (const T*)nullptr);
    }
  } else {
    const unsigned sc = (unsigned)(lane / FB);
#pragma unroll
    for (int j = 0; j < GC; ++j) {
      const unsigned col = min((unsigned)(wv * FlowCfg<T>::CW + j * SC) + sc, (unsigned)(ncol - 1));
      v[j] = flow_bld(r, (row + col * ld) * (unsigned)sizeof(T), 0u, (const T*)nullptr);
    }
  }
}
template <class T>
__device__ __forceinline__ void flow_fma(const T (&v)[FlowCfg<T>::GC], const T* sv, int t, T& s) {
  constexpr int FB = FlowCfg<T>::FB, SC = FlowCfg<T>::SC, GC = FlowCfg<T>::GC;
  const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), sc = lane / FB;
#pragma unroll
  for (int j = 0; j < GC; ++j) s = Scal<T>::fma(v[j], sv[wv * FlowCfg<T>::CW + j * SC + sc], s);
}
template <class T>
__device__ __forceinline__ void flow_put(T* s_red, int t, T s) {
  constexpr int FB = FlowCfg<T>::FB, SC = FlowCfg<T>::SC;
  const int lane = t & 63, wv = t >> 6;
  s_red[(wv * SC + lane / FB) * FB + lane % FB] = s;
}
template <class T>
__device__ __forceinline__ T flow_sum(const T* s_red, int t) {  // t < FB
  T v = s_red[t];
#pragma unroll
  for (int g = 1; g < FlowCfg<T>::NP; ++g) v = v + s_red[g * FlowCfg<T>::FB + t];
  return v;
}

// The rows of the Gauss transform (rows >= ni of the forward sweep: rhs[bnd] -= (Abi*U^-1) * y) depend on nothing but the y they multiply, and they
// are most of what the forward sweep reads.  They are walked in TALL tiles -- FBB = 256 (ComplexF64: 128) rows x 64 columns, 128 KB like the tiles of
// the chain -- because what limits these rounds is DRAM efficiency, and a column of a tall tile is 2 KB of consecutive addresses instead of 512 bytes
// (measured with 32-row blocks, i.e. 256-byte segments: 29.6 ms instead of 27.6).  Lanes along 64 rows, RG row groups, CG column groups per workgroup.
template <class T>
struct FlowBCfg {
  static constexpr int FBB = sizeof(T) == 8 ? 256 : 128;  // rows of a boundary block
  static constexpr int RG = FBB / 64;                      // row groups (waves along the rows)
  static constexpr int CG = (FlowCfg<T>::NT / 64) / RG;    // column groups
  static constexpr int CT = 64;                            // columns of a tile
  static constexpr int GC = CT / CG;                       // loads per thread and tile (32 / 16): 64 data registers
};
template <class T>
__device__ __forceinline__ void flow_load_b(T (&v)[FlowBCfg<T>::GC], const T* base, unsigned ld, int ncol, int rl, int t) {
  constexpr int RG = FlowBCfg<T>::RG, GC = FlowBCfg<T>::GC;
  const int lane = t & 63, wv = __builtin_amdgcn_readfirstlane(t >> 6), rg = wv % RG, cg = wv / RG;
  const unsigned row = (unsigned)min(rg * 64 + lane, rl - 1);
  const unsigned long long pb = (unsigned long long)base;
  const unsigned plo = __builtin_amdgcn_readfirstlane((unsigned)pb), phi = __builtin_amdgcn_readfirstlane((unsigned)(pb >> 32));
  ld = __builtin_amdgcn_readfirstlane(ld);
  ncol = __builtin_amdgcn_readfirstlane(ncol);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(((unsigned long long)phi << 32) | plo), 0, 0x7FFFFFFF, 0x00020000);
  const unsigned voff = row * (unsigned)sizeof(T);
#pragma unroll
  for (int j = 0; j < GC; ++j) {
    const unsigned col = min((unsigned)(cg * GC + j), (unsigned)(ncol - 1));  // scalar
    v[j] = flow_bld(r, voff, col * ld * (unsigned)sizeof(T), (const T*)nullptr);
  }
}
template <class T>
__device__ __forceinline__ void flow_fma_b(const T (&v)[FlowBCfg<T>::GC], const T* sv, int t, T& s) {
  constexpr int RG = FlowBCfg<T>::RG, GC = FlowBCfg<T>::GC;
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6), cg = wv / RG;
#pragma unroll
  for (int j = 0; j < GC; ++j) s = Scal<T>::fma(v[j], sv[cg * GC + j], s);
}

#ifdef HS_FLOW_TRACE  // timing experiment: device timestamps (100 MHz) of the chain phases of front 0, forward sweep: 8 slots per interior sub-block
__device__ unsigned long long g_flow_trace[8 * 4096];
extern "C" int hsk_flow_trace(unsigned long long* out, int n) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_flow_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -1; }
#define FLOW_T(slot) do { if (!UPPER && f == 0 && nbatch == 1 && interior && t == 0 && sb < 4096) g_flow_trace[8 * sb + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FLOW_T(slot) do { } while (0)
#endif
// One sweep of one level.  UPPER = false: forward (L below the diagonal, rows down to mrows: the Abi*U^-1 rows update rhs[bnd]);
// UPPER = true: backward (U above the diagonal).  w: the level's work vector (in: gathered / updated right-hand side), out: y (x).
template <class T, bool UPPER>
__global__ __launch_bounds__(HS_FLOW_NT) void flow_sweep_kernel(const SolveNode<T>* __restrict__ nodes, int nbatch, T* __restrict__ w, T* __restrict__ out,
                                                         T* __restrict__ b, T* __restrict__ E1, T* __restrict__ E2, int* __restrict__ counter,
                                                         int* __restrict__ err, int tall) {
  constexpr int FB = FlowCfg<T>::FB, Q = FlowCfg<T>::Q, GC = FlowCfg<T>::GC;
  using V = T;
  __shared__ int s_id;
  __shared__ T s_v[HS_SW];
  __shared__ T s_red[FB * FlowCfg<T>::NP];
  const int t = threadIdx.x;
  if (t == 0) s_id = atomicAdd(counter, 1);
  __syncthreads();
  const int id = __builtin_amdgcn_readfirstlane(s_id), f = id % nbatch, sb = id / nbatch;  // (scalar: the front's descriptor and every tile base stay in SGPRs)
  const SolveNode<T> nd = nodes[f];
  if (nd.ni <= 0) return;
  const int ncb = (nd.ni + HS_SW - 1) / HS_SW;
  int jb, q, rs, rl;
  bool interior;
  if (!UPPER) {
    interior = sb < Q * ncb;
    if (interior) {
      jb = sb / Q;
      q = sb % Q;
      rs = jb * HS_SW + q * FB;
      rl = min(FB, nd.ni - rs);
    } else {
      jb = ncb;  // all column blocks
      q = 0;
      const int fbb = tall ? FlowBCfg<T>::FBB : FB;  // tall boundary tiles only where the level has enough of them to fill the chip (launch_fwd_flow)
      rs = nd.ni + (sb - Q * ncb) * fbb;
      rl = min(fbb, nd.mrows - rs);
    }
  } else {
    interior = true;
    if (sb >= Q * ncb) return;
    jb = ncb - 1 - sb / Q;
    q = Q - 1 - sb % Q;
    rs = jb * HS_SW + q * FB;
    rl = min(FB, nd.ni - rs);
  }
  if (rl <= 0) return;
  if (!UPPER && !interior && tall) {  // tall tiles: FBB rows x 64 columns (FlowBCfg)
    constexpr int FBB = FlowBCfg<T>::FBB, RG = FlowBCfg<T>::RG, CG = FlowBCfg<T>::CG, CT = FlowBCfg<T>::CT, GB = FlowBCfg<T>::GC;
    static_assert(FBB * CG <= FlowCfg<T>::FB * FlowCfg<T>::NP, "the partial sums of a boundary block must fit s_red");
    T wold_b = Scal<T>::zero();
    int gi_b = 0;
    if (t < rl) {
      gi_b = gld(nd.fidx + rs + t);
      wold_b = b[gi_b];
    }
    const T* brow = nd.LF + (size_t)rs;
    const int nr = (nd.ni + CT - 1) / CT;  // rounds: 64 columns each
    T sb_ = Scal<T>::zero();
    T ua[GB], ub[GB];
    auto request_b = [&](T(&v)[GB], int c) {
      if (c >= nr) return;
      flow_load_b<T>(v, brow + (size_t)c * CT * nd.ldl, (unsigned)nd.ldl, min(CT, nd.ni - c * CT), rl, t);
    };
    auto round_b = [&](const T(&v)[GB], int c) -> bool {
      const int wl = min(CT, nd.ni - c * CT);
      const bool ok = t < CT ? flow_poll(E1 + nd.woff + (size_t)c * CT, wl, s_v, t) : true;
      if (__builtin_amdgcn_readfirstlane(__syncthreads_or(ok ? 0 : 1))) return false;
      flow_fma_b<T>(v, s_v, t, sb_);
      __syncthreads();
      return true;
    };
    bool alive_b = true;
    int c = 0;
    if (nr & 1) {
      request_b(ua, 0);
      alive_b = round_b(ua, 0);
      c = 1;
    }
    request_b(ua, c);
    while (c < nr && alive_b) {
      request_b(ub, c + 1);
      alive_b = round_b(ua, c);
      request_b(ua, c + 2);
      if (alive_b) alive_b = round_b(ub, c + 1);
      c += 2;
    }
    if (!alive_b) {
      if (t == 0) *(volatile int*)err = 1;
      return;
    }
    {  // partial sums of the CG column groups meet in LDS
      const int lane = t & 63, wv = t >> 6, rg = wv % RG, cg = wv / RG;
      s_red[cg * FBB + rg * 64 + lane] = sb_;
    }
    __syncthreads();
    if (t < rl) {
      T v = s_red[t];
#pragma unroll
      for (int g = 1; g < CG; ++g) v = v + s_red[g * FBB + t];
      b[gi_b] = wold_b - v;
    }
    return;
  }
  T wold = Scal<T>::zero();
  int gi = 0;
  if (t < rl) {  // ahead of the tiles
    if (interior) {
      wold = w[nd.woff + rs + t];
    } else {
      gi = gld(nd.fidx + rs + t);
      wold = b[gi];
    }
  }
  T s = Scal<T>::zero();
  const T* arow = nd.LF + (size_t)rs;
  const int kfirst = UPPER ? ncb - 1 : 0, kstep = UPPER ? -1 : 1, kcount = UPPER ? ncb - 1 - jb : jb;
  // what the diagonal block needs: columns [c_lo, c_hi) of the slab of the stored inverse
  const int wlj = interior ? min(HS_SW, nd.ni - jb * HS_SW) : 1;
  const int c_lo = UPPER ? q * FB : 0, c_hi = UPPER ? wlj : min(wlj, (q + 1) * FB);
  const T* inv = interior ? (UPPER ? nd.inv256U : nd.inv256L) + (size_t)jb * HS_SW * HS_SW + (size_t)(q * FB) : nullptr;
  V va[GC], vb[GC];
  // requests the tile of round c (c >= kcount: the slab of the inverse; a boundary block has none) into `v`:
  // one straight run of loads whatever c is -- branches here made the compiler keep four tiles alive and spill
  const T* tail_base = interior ? inv : arow;
  const unsigned tail_ld = interior ? (unsigned)HS_SW : (unsigned)nd.ldl;
  auto request = [&](V(&v)[GC], int c) {
    const bool last = c >= kcount;
    if (last && !interior) return;  // (uniform, and no loads at all behind it)
    const int k = kfirst + (last ? 0 : c) * kstep;
    const T* base = last ? tail_base : arow + (size_t)k * HS_SW * nd.ldl;
    const int ncol = (last && interior) ? HS_SW : min(HS_SW, nd.ni - k * HS_SW);
    flow_load<T>(v, base, last ? tail_ld : (unsigned)nd.ldl, ncol, rl, t);
  };
  // round c < kcount: wait for y_k, accumulate; returns false when the wait ran out (uniform)
  auto round = [&](const V(&v)[GC], int c) -> bool {
    const int k = kfirst + c * kstep;
    const int wl = min(HS_SW, nd.ni - k * HS_SW);
    const bool ok = flow_poll(E1 + nd.woff + (size_t)k * HS_SW, wl, s_v, t);
    if (__builtin_amdgcn_readfirstlane(__syncthreads_or(ok ? 0 : 1))) return false;  // (also publishes s_v to the workgroup; readfirstlane: the loop
                                                                                        // counter must stay provably uniform, or every buffer load turns into a waterfall loop)
    flow_fma<T>(v, s_v, t, s);
    __syncthreads();  // s_v is refilled by the next round
    return true;
  };
  bool alive = true;
  int c = 0;
  if (kcount & 1) {  // an odd count: one round on its own first, so that the pairs below always end with the inverse in `va`
    request(va, 0);
    alive = round(va, 0);
    c = 1;
  }
  request(va, c);
  while (c < kcount && alive) {
    request(vb, c + 1);
    alive = round(va, c);
    request(va, c + 2);  // (c + 2 == kcount: the slab of the inverse, requested before the last y is waited for)
    if (alive) alive = round(vb, c + 1);
    c += 2;
  }
  FLOW_T(0);  // last round done (y of the block before arrived, FMAs done)
  if (!alive) {
    if (t == 0) *(volatile int*)err = 1;  // pinned host memory: a plain store
    return;
  }
  flow_put<T>(s_red, t, s);
  __syncthreads();
  T wfin = Scal<T>::zero();
  if (t < rl) wfin = wold - flow_sum<T>(s_red, t);
  if (!interior) {
    if (t < rl) b[gi] = wfin;
    return;
  }
  // the diagonal block: publish the finished w of these rows, fetch the sub-blocks the inverse needs, multiply by the slab of the stored inverse
  if (t < rl) flow_publish(E2 + nd.woff + rs + t, wfin);
  FLOW_T(1);  // w published
  bool ok = true;
  if (t < HS_SW) {
    const int own_lo = q * FB;
    const bool own = t >= own_lo && t < own_lo + FB;
    if (!own) {
      if (t >= c_lo && t < c_hi)
        ok = flow_poll(E2 + nd.woff + (size_t)jb * HS_SW, c_hi, s_v, t);  // element t of the block's finished w
      else
        s_v[t] = Scal<T>::zero();
    }
    if (t < FB) s_v[own_lo + t] = wfin;  // own rows: no round trip (zero beyond rl)
  }
  if (__builtin_amdgcn_readfirstlane(__syncthreads_or(ok ? 0 : 1))) {
    if (t == 0) *(volatile int*)err = 1;
    return;
  }
  FLOW_T(2);  // w of the block arrived
  {
    T s2 = Scal<T>::zero();
    flow_fma<T>(va, s_v, t, s2);
    flow_put<T>(s_red, t, s2);
  }
  __syncthreads();
  if (t < rl) {
    const T v = flow_sum<T>(s_red, t);
    flow_publish(E1 + nd.woff + rs + t, v);
    out[nd.woff + rs + t] = v;
    FLOW_T(3);  // y published
    if (!UPPER) w[nd.woff + rs + t] = wfin;  // (what the launch-per-step sweep leaves behind: nothing reads it, kept for identical buffers)
  }
}

template <class T>
void launch_fwd_flow(const SolveNode<T>* dn, int nbatch, int maxni, int maxnb, T* w, T* y, T* b, T* E1, T* E2, int* counter, int* err, hipStream_t s) {
  if (nbatch <= 0 || maxni <= 0) return;
  constexpr int FB = FlowCfg<T>::FB, FBB = FlowBCfg<T>::FBB;
  // tall boundary tiles (2 KB column segments) where the level has enough boundary blocks to fill the chip several times; at the top of the tree a
  // tall block is a serial walk over ni/64 rounds with too few workgroups beside it (measured: levels 2-4 of Poisson 128^3 20 % slower, the leaf level 4 % faster)
  const int tall = (long long)nbatch * ((std::max(maxnb, 0) + FBB - 1) / FBB) >= 1536 ? 1 : 0;
  const int fbb = tall ? FBB : FB;
  const int nsb = (HS_SW / FB) * ((maxni + HS_SW - 1) / HS_SW) + (std::max(maxnb, 0) + fbb - 1) / fbb;
  hipLaunchKernelGGL((flow_sweep_kernel<T, false>), dim3((unsigned)nsb * (unsigned)nbatch), dim3(FlowCfg<T>::NT), 0, s, dn, nbatch, w, y, b, E1, E2, counter, err, tall);
}
template <class T>
void launch_bwd_flow(const SolveNode<T>* dn, int nbatch, int maxni, T* w, T* x, T* E1, T* E2, int* counter, int* err, hipStream_t s) {
  if (nbatch <= 0 || maxni <= 0) return;
  const int nsb = (HS_SW / FlowCfg<T>::FB) * ((maxni + HS_SW - 1) / HS_SW);
  hipLaunchKernelGGL((flow_sweep_kernel<T, true>), dim3((unsigned)nsb * (unsigned)nbatch), dim3(FlowCfg<T>::NT), 0, s, dn, nbatch, w, x, (T*)nullptr, E1, E2, counter, err, 0);
}
template void launch_fwd_flow<double>(const SolveNode<double>*, int, int, int, double*, double*, double*, double*, double*, int*, int*, hipStream_t);
template void launch_fwd_flow<cplx>(const SolveNode<cplx>*, int, int, int, cplx*, cplx*, cplx*, cplx*, cplx*, int*, int*, hipStream_t);
template void launch_bwd_flow<double>(const SolveNode<double>*, int, int, double*, double*, double*, double*, int*, int*, hipStream_t);
template void launch_bwd_flow<cplx>(const SolveNode<cplx>*, int, int, cplx*, cplx*, cplx*, cplx*, int*, int*, hipStream_t);

// Latency of ONE exchange of the dataflow sweeps, measured: workgroup 0 publishes a counter value, workgroup `peer` answers it; `iters`
// round trips with the publish / poll primitives above.  peer = 1: the next workgroup of the grid (another XCD: the hardware deals
// workgroups round-robin over the 8 XCDs), peer = 8: the same XCD, another CU.  Every poll is bounded.
__global__ __launch_bounds__(64) void flow_pingpong_kernel(double* A, double* B, int peer, int iters) {
  if ((int)blockIdx.x != 0 && (int)blockIdx.x != peer) return;
  if (threadIdx.x != 0) return;
  const bool first = blockIdx.x == 0;
  for (int i = 1; i <= iters; ++i) {
    const unsigned long long want = (unsigned long long)__double_as_longlong((double)i);
    if (first) {
      flow_publish(A, (double)i);
      int it = 0;
      while (flow_ldbits(B) != want && ++it < HS_FLOW_SPIN) __builtin_amdgcn_s_sleep(1);
      if (it >= HS_FLOW_SPIN) return;
    } else {
      int it = 0;
      while (flow_ldbits(A) != want && ++it < HS_FLOW_SPIN) __builtin_amdgcn_s_sleep(1);
      if (it >= HS_FLOW_SPIN) return;
      flow_publish(B, (double)i);
    }
  }
}
extern "C" double hsk_flow_pingpong_us(int peer, int iters) {
  double* d = nullptr;
  if (peer < 1 || iters < 1 || hipMalloc((void**)&d, 4096) != hipSuccess) return -1.0;
  (void)hipMemset(d, 0, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float ms = 0.f, ms1 = 0.f;
  for (int rep = 0; rep < 2; ++rep) {  // the second pair is the measurement: (iters) against (1) round trips, so the launch itself cancels
    (void)hipMemset(d, 0, 4096);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(flow_pingpong_kernel, dim3(peer + 1), dim3(64), 0, 0, d, d + 256, peer, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemset(d, 0, 4096);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(flow_pingpong_kernel, dim3(peer + 1), dim3(64), 0, 0, d, d + 256, peer, 1);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms1, e0, e1);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d);
  return iters > 1 ? (double)(ms - ms1) * 1e3 / (iters - 1) : (double)ms * 1e3;
}

template <class T>
void launch_inv256(const SolveNode<T>* dn, int nbatch, int maxni, hipStream_t s, int only_block) {
  if (nbatch <= 0 || maxni <= 0) return;
  const int first = only_block >= 0 ? only_block : 0;
  const int nb256 = only_block >= 0 ? 1 : (maxni + HS_SW - 1) / HS_SW;
  if (first * HS_SW >= maxni) return;
  constexpr int lds_bytes = (int)(sizeof(T) * 9 * Inv256Cfg<T>::BS);
  static bool attr_set = false;
  if (!attr_set) {  // 72 KiB of LDS per workgroup needs the opt-in
    (void)hipFuncSetAttribute((const void*)inv256_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    attr_set = true;
  }
  hipLaunchKernelGGL(inv256_kernel<T>, dim3(nb256 * 8 * Inv256Cfg<T>::H, nbatch, 2), dim3(256), lds_bytes, s, dn, first);
}
template <class T>
void launch_fwd_wide(const SolveNode<T>* dn, int nbatch, int blk, int maxm, T* w, T* y, T* b, hipStream_t s) {
  if (nbatch <= 0) return;
  const int rows = maxm - blk * HS_SW;
  const int gx = rows > 0 ? (rows + 255) / 256 : 1;
  static const bool first = !(getenv("HS_SOLVE_FIRST") && getenv("HS_SOLVE_FIRST")[0] == '0');
  if (first && blk == 0) hipLaunchKernelGGL((wide_first_kernel<T, true>), dim3(nbatch), dim3(1024), 0, s, dn, (const T*)w, y);
  hipLaunchKernelGGL(fwd_wide_kernel<T>, dim3(gx, nbatch), dim3(1024), 0, s, dn, blk, w, y, b, first ? 1 : 0);
}
template <class T>
void launch_bwd_wide(const SolveNode<T>* dn, int nbatch, int blk, T* w, T* x, hipStream_t s, bool first_call) {
  if (nbatch <= 0) return;
  const int rows = blk * HS_SW;
  const int gx = rows > 0 ? (rows + 255) / 256 : 1;
  static const bool first = !(getenv("HS_SOLVE_FIRST") && getenv("HS_SOLVE_FIRST")[0] == '0');
  if (first && first_call) hipLaunchKernelGGL((wide_first_kernel<T, false>), dim3(nbatch), dim3(1024), 0, s, dn, (const T*)w, x);  // every front's own last block
  hipLaunchKernelGGL(bwd_wide_kernel<T>, dim3(gx, nbatch), dim3(1024), 0, s, dn, blk, w, x, first ? 1 : 0);
}
int hs_solve_wide_cols() { return HS_SW; }

template void launch_inv256<double>(const SolveNode<double>*, int, int, hipStream_t, int);
template void launch_inv256<cplx>(const SolveNode<cplx>*, int, int, hipStream_t, int);
template void launch_fwd_wide<double>(const SolveNode<double>*, int, int, int, double*, double*, double*, hipStream_t);
template void launch_fwd_wide<cplx>(const SolveNode<cplx>*, int, int, int, cplx*, cplx*, cplx*, hipStream_t);
template void launch_bwd_wide<double>(const SolveNode<double>*, int, int, double*, double*, hipStream_t, bool);
template void launch_bwd_wide<cplx>(const SolveNode<cplx>*, int, int, cplx*, cplx*, hipStream_t, bool);
