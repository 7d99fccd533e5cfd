// hs_common.h -- shared types of the gfx950 nested-dissection elimination kernels.
//
// Data layout in HBM.  One front = one node of the elimination tree with
// ni eliminated ("int") and nb boundary ("bnd") DOFs, front order [int; bnd] -- the reference's
// BlockMatrix quadruple Aii, Aib, Abi, Abb (src/factorization.jl:115-123):
//
//   LF  (m x ni, ld = ldl, m = ni+nb)  = [Aii; Abi]   -> after factor: [L\U ; Abi*U^-1]
//   UR  (ni x nb, ld = ldu)            =  Aib         -> after factor:  L^-1 * P * Aib
//   SB  (nb x nb, ld = lds)            =  Abb         -> after factor:  Schur complement S (temporary)
//
// all column-major like Julia's Matrix{T}.  LF and UR live in the permanent factor arena, SB in a
// per-level scratch arena until the parent front has absorbed it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HS_PB 32          // panel width (columns factored per tournament-pivoting step)
#define HS_CHUNK 256      // rows per tournament chunk (= threads per workgroup)
#define HS_BIG (1 << 30)
#define HS_GROWTH_MAX 4.0  // optimistic diagonal-block pivoting is accepted while every multiplier |l_ij| <= this

struct cplx {
  double re, im;
};

__host__ __device__ inline cplx operator+(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__host__ __device__ inline cplx operator-(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__host__ __device__ inline cplx operator-(cplx a) { return {-a.re, -a.im}; }
__host__ __device__ inline cplx operator*(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__host__ __device__ inline cplx operator/(cplx a, cplx b) {
  // Smith's algorithm (what LAPACK's zladiv-free paths and Julia's `/` use for robustness)
  if (fabs(b.re) >= fabs(b.im)) {
    double r = b.im / b.re, d = b.re + b.im * r;
    return {(a.re + a.im * r) / d, (a.im - a.re * r) / d};
  } else {
    double r = b.re / b.im, d = b.re * r + b.im;
    return {(a.re * r + a.im) / d, (a.im * r - a.re) / d};
  }
}

// Global-address-space accessors.  Matrix pointers reach the kernels through descriptors read from memory (NodeDesc,
// GemmProb, ...), so the compiler only knows them as generic pointers and emits flat_load / flat_store.  A flat access
// counts on lgkmcnt as well as vmcnt: every `s_waitcnt lgkmcnt(0)` in front of an LDS operand read then also waits for
// the global prefetch issued just before it, which serialises the software pipeline of the tile kernels.  gld/gst cast
// to address space 1, the accesses become global_load / global_store (vmcnt only).
#define HS_AS_GLOBAL __attribute__((address_space(1)))
typedef double hs_d2u __attribute__((ext_vector_type(2), aligned(8)));  // 16-byte access, 8-byte aligned
template <class T>
__device__ __forceinline__ T gld(const T* p) {
  return *(const T HS_AS_GLOBAL*)p;
}
template <>
__device__ __forceinline__ cplx gld<cplx>(const cplx* p) {
  hs_d2u v = *(const hs_d2u HS_AS_GLOBAL*)p;
  return {v.x, v.y};
}
template <class T>
__device__ __forceinline__ void gst(T* p, T v) {
  *(T HS_AS_GLOBAL*)p = v;
}
template <>
__device__ __forceinline__ void gst<cplx>(cplx* p, cplx v) {
  hs_d2u w = {v.re, v.im};
  *(hs_d2u HS_AS_GLOBAL*)p = w;
}
__device__ __forceinline__ hs_d2u gld2(const double* p) { return *(const hs_d2u HS_AS_GLOBAL*)p; }
__device__ __forceinline__ hs_d2u gld2(const cplx* p) { return *(const hs_d2u HS_AS_GLOBAL*)p; }
__device__ __forceinline__ void gst2(cplx* p, hs_d2u v) { *(hs_d2u HS_AS_GLOBAL*)p = v; }

template <class T>
struct Scal;
template <>
struct Scal<double> {
  static __host__ __device__ inline double zero() { return 0.0; }
  static __host__ __device__ inline double one() { return 1.0; }
  static __host__ __device__ inline double abs1(double a) { return fabs(a); }  // idamax
  static __host__ __device__ inline double fma(double a, double b, double c) { return ::fma(a, b, c); }
  // c - a*b
  static __host__ __device__ inline double fnma(double a, double b, double c) { return ::fma(-a, b, c); }
};
template <>
struct Scal<cplx> {
  static __host__ __device__ inline cplx zero() { return {0.0, 0.0}; }
  static __host__ __device__ inline cplx one() { return {1.0, 0.0}; }
  static __host__ __device__ inline double abs1(cplx a) { return fabs(a.re) + fabs(a.im); }  // izamax (cabs1)
  static __host__ __device__ inline cplx fma(cplx a, cplx b, cplx c) {
    return {::fma(a.re, b.re, ::fma(-a.im, b.im, c.re)), ::fma(a.re, b.im, ::fma(a.im, b.re, c.im))};
  }
  static __host__ __device__ inline cplx fnma(cplx a, cplx b, cplx c) {
    return {::fma(-a.re, b.re, ::fma(a.im, b.im, c.re)), ::fma(-a.re, b.im, ::fma(-a.im, b.re, c.im))};
  }
};

// Device-visible description of one front; an array of these (one per node of the current batch)
// is uploaded once per level, every grouped kernel indexes it with blockIdx.y.
template <class T>
struct NodeDesc {
  T* LF;
  T* UR;
  T* SB;
  T* invL;       // ceil(ni/32) blocks of 32x32 (column-major, ld 32): inverse of the unit-lower diagonal block
  T* invU;       // same for the upper diagonal block
  T* inv256L;    // ceil(ni/256) blocks of 256x256 (ld 256): inverses of the 256x256 diagonal blocks of L (null: not kept)
  T* inv256U;    // same for U.  Built as soon as the 256 columns are factored (kernels_solve_wide.hip): TRSM base case + ldiv!
  int* ipiv;     // ni entries: LAPACK-style swap targets (0-based row inside the front)
  int* rperm;    // ni entries: accumulated row permutation, (P x)[i] = x[rperm[i]] (used by ldiv!)
  int* cand0;    // tournament candidate lists (ping-pong)
  int* cand1;
  int* pivlist;  // HS_PB selected rows of the current panel
  int* info;     // 0 = ok, else 1 + first column with an exactly zero pivot (SingularException)
  int* growth;   // optimistic pivoting (panel_pivot, fuse & 4): set to 1 when a multiplier exceeded HS_GROWTH_MAX or a block was
                 // singular on its own rows -- the level is then redone with tournament pivoting; may be null
  const int* fidx;  // m global DOF ids (0-based), front order [int; bnd]
  int ni, nb, m;
  int ldl, ldu, lds;
  int ni1, nb1;  // branch: sizes of the left child's contribution to int / bnd (front split points); leaf: ni, nb
  const int* spos;         // first slice of a split front with re-ordered interior: position < s_ni -> original position; else null
  int s_ni, s_ni1, s_nb1;  // the same three split points of the front the CHILDREN address (which child a front position came
                           // from, for the gather): ni, ni1, nb1 except for the first slice of a split front (hs_split.h)
  int pivrows;   // pivot candidates are rows [c0, pivrows): ni for a front (`\\` on Aii pivots inside Aii only), all rows for a sketch
  int isleaf;
  int node;      // post-order id
  // the same three matrices indexed by HS_MAT_*: kernels that pick a matrix at run time index these
  // tables (plain address arithmetic) instead of branching over LF/UR/SB -- hipcc (ROCm 7.2) was seen
  // to miscompile the three-way scalar select of field addresses ("compiler notes").
  T* mp[3];
  int mld[3], mrows[3], mcols[3];
  __host__ void finalize() {
    if (pivrows <= 0) pivrows = ni;
    if (s_ni <= 0) { s_ni = ni; s_ni1 = ni1; s_nb1 = nb1; }
    mp[0] = LF; mp[1] = UR; mp[2] = SB;
    mld[0] = ldl; mld[1] = ldu; mld[2] = lds;
    mrows[0] = m; mrows[1] = ni; mrows[2] = nb;
    mcols[0] = ni; mcols[1] = nb; mcols[2] = nb;
  }
};

enum { HS_MAT_LF = 0, HS_MAT_UR = 1, HS_MAT_SB = 2 };

// Sub-block op of the recursive LU on a batch of fronts: ranges are given in front coordinates and
// clipped per node to the extents of the matrices they address.
struct GemmOp {
  int cmat, bmat;  // C and B live in LF / UR / SB; A is LF (or a stored inverse diagonal block, see ainv)
  int r0, r1;      // C rows   (A rows are the same, shifted by ni when C is SB)
  int c0, c1;      // C cols = B cols
  int k0, k1;      // A cols = B rows (always inside [0, ni))
  int ainv;        // 1: A = invL[r0/32] (32x32), C = A*B in place on rows [r0, r0+32): the TRSM base case; 2: A = invU[r0/32];
                   // 3..6: the 256-row base case through inv256L / inv256U, as two in-place half products (resolve_op)
                   // 7, 8: rows [r0, ..) of the 256 columns from k0 on times inv256U[k0/256], in place: the multipliers of the rows BELOW a
                   //       256-wide diagonal block once that block is factored (Sched::lu_rec, optimistic pivoting): 7 = columns 128.. (first), 8 = columns 0..127
                   // 16 + q: the same for 64-column tiles (ComplexF64): column block q of the group from all blocks <= q, run for q = 3, 2, 1, 0
  int cap;         // > 0: launch at most this many workgroups per front (they walk the tiles): leaves CU slots free for a
                   // concurrent stream (the look-ahead panel chain); 0: one workgroup per tile
  int prio;        // 1: raise the waves' issue priority (s_setprio): panel work of the look-ahead side stream
};

// plain problem (test hooks, root Schur, compressed path)
template <class T>
struct GemmProb {
  const T* A;
  const T* B;
  T* C;
  int M, N, K;
  int lda, ldb, ldc;
  int* flag = nullptr;  // optimistic pivoting: raised when a stored value of rows < flag_rows exceeds HS_GROWTH_MAX (GemmOp::ainv 7, 8: the values ARE multipliers)
  int flag_rows = 0;
};

template <class T>
__device__ inline void mat_of(const NodeDesc<T>* pn, int which, T*& p, int& ld, int& rows, int& cols) {
  p = pn->mp[which];
  ld = pn->mld[which];
  rows = pn->mrows[which];
  cols = pn->mcols[which];
}

// ---- launch API (implemented in the kernels_*.hip files) -------------------------------------
template <class T>
void launch_gemm_op(const NodeDesc<T>* dnodes, int nbatch, int maxM, int maxN, const GemmOp& op, hipStream_t s);
template <class T>
void launch_gemm_probs(const GemmProb<T>* dprobs, int nprob, int maxM, int maxN, int accumulate_minus, hipStream_t s);

template <class T>
void launch_tournament_round(const NodeDesc<T>* dnodes, int nbatch, int pb, int round, int maxchunks, hipStream_t s);
template <class T>
void launch_tournament_stage(const NodeDesc<T>* dnodes, int nbatch, int pb, int stage, int maxblocks, hipStream_t s);
int hs_tour_block_rows(bool is_complex);
// Streams of the look-ahead schedule (hs_sched.h): *side = high-priority stream on every CU; the masked pair
// (*side_masked owns HS_LA_SIDE_CUS compute units, default 32, *la every other one) is used for a lone front.
// Any of them may come back null (the schedule then falls back: no reservation / no look-ahead).
void hs_create_lookahead_streams(hipStream_t* la, hipStream_t* side_masked, hipStream_t* side);  // rows one workgroup of a tournament stage reduces to 32 nominees
template <class T>
void launch_panel_pivot(const NodeDesc<T>* dnodes, int nbatch, int pb, int fuse, hipStream_t s);
template <class T>
void launch_panel_l21(const NodeDesc<T>* dnodes, int nbatch, int pb, int maxrows, int fuse, hipStream_t s, int rlim = HS_BIG);  // rows < rlim only
template <class T>
bool launch_trsm_small(const NodeDesc<T>* dnodes, int nbatch, int mat, int r0, int rows, int c0, int c1, int maxcols, hipStream_t s);  // 64 / 128 rows in one launch
template <class T>
bool launch_group256(const NodeDesc<T>* dnodes, int nbatch, int grp, hipStream_t s);  // the panel chain of the 256 x 256 diagonal block of group `grp` in one launch (Float64; false: not available)
template <class T>
void launch_laswp(const NodeDesc<T>* dnodes, int nbatch, int mat, int c0, int c1, int k0, int k1, int maxcols, hipStream_t s);

template <class T>
void launch_init_fronts(const NodeDesc<T>* dnodes, int nbatch, int maxni, hipStream_t s);  // rperm = 0:ni-1, info = 0
template <class T>
void launch_mark(const NodeDesc<T>* dnodes, int nbatch, int maxm, int* own, int* pos, hipStream_t s);
template <class T>
void launch_gather(const NodeDesc<T>* dnodes, int nbatch, int maxm, const int64_t* colptr, const int32_t* rowval,
                   const T* nzval, const int* own, const int* pos, hipStream_t s);
template <class T>
struct ScatterDesc {  // child Schur complement -> parent front
  const T* S;         // child's SB (nbc x nbc, ld = lds)
  const int* cmap;    // nbc entries: position in the parent front, -1 = dropped
  int nbc, lds;
  int parent;         // index of the parent in the current batch's NodeDesc array
};
template <class T>
void launch_scatter(const NodeDesc<T>* dnodes, const ScatterDesc<T>* dsc, int nsc, int maxnbc, hipStream_t s);

// solve phase (kernels_solve.hip); one right-hand side per launch
template <class T>
struct SolveNode {
  const T* LF;
  const T* UR;
  const T* invL;
  const T* invU;
  T* inv256L;        // ceil(ni/256) blocks of 256 x 256 (column-major): inverses of the diagonal blocks of L (kernels_solve_wide.hip)
  T* inv256U;        // same for U
  const int* rperm;  // ni: (P x)[i] = x[rperm[i]]
  const int* fidx;   // m global ids (0-based), front order [int; bnd]
  int ni, nb, m, ldl, ldu;
  int mrows;         // rows the dense forward sweep covers: m, or ni when the Gauss transforms are low-rank
  int compressed;    // 1: L (Abi*Aii^-1) and R (Aii^-1*Aib) are applied through their low-rank factors
  long long woff;    // offset of this node's ni-segment in the work vectors
  long long poff;    // offset of this node's partial-sum scratch (ceil(nb/512) * ni entries)
};
template <class T>
void launch_fwd_gather(const SolveNode<T>* dn, int nbatch, int maxni, const T* b, T* w, hipStream_t s);
template <class T>
void launch_fwd_step(const SolveNode<T>* dn, int nbatch, int blk, int maxm, T* w, T* y, T* b, hipStream_t s);
template <class T>
void launch_int_update(const SolveNode<T>* dn, int nbatch, int maxni, int maxnb, const T* b, T* part, const T* y, T* w, hipStream_t s);
template <class T>
void launch_bwd_step(const SolveNode<T>* dn, int nbatch, int blk, T* w, T* x, hipStream_t s);
// the same sweeps 256 columns per launch (kernels_solve_wide.hip)
template <class T>
void launch_fwd_wide(const SolveNode<T>* dn, int nbatch, int blk, int maxm, T* w, T* y, T* b, hipStream_t s);
template <class T>
void launch_bwd_wide(const SolveNode<T>* dn, int nbatch, int blk, T* w, T* x, hipStream_t s, bool first_call);  // first_call: the first step of the level's sweep
template <class T>
void launch_inv256(const SolveNode<T>* dn, int nbatch, int maxni, hipStream_t s, int only_block = -1);
int hs_solve_wide_cols();
// the sweeps of a whole level in ONE launch (dataflow over published values; kernels_solve_wide.hip).  E1, E2: exchange vectors laid out like w,
// filled with 0xFF bytes (the sentinel) before the launch
template <class T>
void launch_fwd_flow(const SolveNode<T>* dn, int nbatch, int maxni, int maxnb, T* w, T* y, T* b, T* E1, T* E2, int* counter, int* err, hipStream_t s);
template <class T>
void launch_bwd_flow(const SolveNode<T>* dn, int nbatch, int maxni, T* w, T* x, T* E1, T* E2, int* counter, int* err, hipStream_t s);
template <class T>
void launch_bwd_scatter(const SolveNode<T>* dn, int nbatch, int maxni, T* b, const T* x, hipStream_t s);

// low-rank Gauss transform  X ~= P' * trap(Lp[:, :r]) * Z  applied to a vector (one front per call):
//   t = Z * x  (x contiguous, or gathered through xidx when xidx != nullptr)      -> launch_lr_zmul
//   u = trap(Lp) * t ;  dst[didx ? didx[rp[i]] : rp[i]] -= u[i]                    -> launch_lr_trap
template <class T>
void launch_lr_zmul(const T* Z, int ldz, int r, int cols, const T* x, const int* xidx, T* part, T* t, hipStream_t s);
template <class T>
void launch_lr_trap(const T* Lp, int ldp, int rows, int r, const int* rp, const T* t, T* dst, const int* didx, hipStream_t s);
void launch_pack_idx(const int* idx, int cnt, const void* b, void* buf, int esz, hipStream_t s);    // buf[i] = b[idx[i]]
void launch_unpack_idx(const int* idx, int cnt, void* b, const void* buf, int esz, hipStream_t s);  // b[idx[i]] = buf[i]
void launch_copy_idx(const int* idx, int cnt, const void* b, void* out, int esz, hipStream_t s);      // out[idx[i]] = b[idx[i]]

// matrix-free compressed fronts (kernels_mf.hip)
struct HsFillEntry {
  int row, col;  // destination in a zero-filled dense block
  long long e;   // >= 0: the value is nz[e]; < 0: one
};
template <class T>
void launch_fill_entries(const HsFillEntry* ent, int cnt, const T* nz, T* out, int ld, hipStream_t s);
template <class T>
void launch_identity_cols(T* X, int ld, int n, int c0, int cols, hipStream_t s);
template <class T>
void launch_perm_gather(const T* src, const int64_t* perm, T* dst, int64_t cnt, hipStream_t s);
template <class T>
void launch_transpose(const T* in, int ldi, T* out, int ldo, int rows, int cols, hipStream_t s);
template <class T>
void launch_negate(T* A, int ld, int rows, int cols, hipStream_t s);

void hs_set_error(int code, long long info, const char* fmt, ...);

#define HS_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t e__ = (call);                                                                  \
    if (e__ != hipSuccess) {                                                                  \
      hs_set_error(-6, 0, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
      throw (int)-6;                                                                          \
    }                                                                                         \
  } while (0)
