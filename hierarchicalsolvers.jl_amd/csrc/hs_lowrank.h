// hs_lowrank.h -- device low-rank factor X ~= P' * trap(Lp[:, :r]) * Z (see hs_lowrank.hip)
#pragma once
#include <cstdint>
#include "hs_common.h"

// Device blocks of the compressions (sketches, permutations, factors): recycled through a process-wide cache.  Every compression of a
// block allocated and released a handful of them, and a hipFree synchronises the whole device: 16 blocks per tree level of an HSS
// compression made the ALLOCATOR the cost of its pivoting stage, and serialised the host threads that compress fronts side by side.
// hs_lr_free accepts any device pointer: one it did not hand out goes to hipFree.
int hs_lr_alloc(void** out, size_t bytes);  // 0 on success (hipSuccess), else the hipError_t
void hs_lr_free(void* p);
int64_t hs_lr_trim();  // releases the cached blocks to the driver; returns their bytes

template <class T>
struct LowRank {
  T* Lp = nullptr;      // rows x k packed L\U of the pivoted sketch (ld = ldp); C = P' * unit-lower-trapezoid(Lp[:, :r])
  int* rperm = nullptr; // rows: pivoted row i of Lp is original row rperm[i]
  T* Z = nullptr;       // r x cols (ld = ldz)
  int rows = 0, cols = 0, k = 0, r = 0, ldp = 0, ldz = 0;
  double top = 0.0;     // |u_11| of the pivoted sketch: the scale the relative tolerance was applied to
  T* Cd = nullptr;      // optional dense C (rows x r, ld = ldc) in ORIGINAL row order; when set it replaces P'*trap(Lp)
  int ldc = 0;
  T* Y0 = nullptr;      // optional copy of the sketch X*Omega (rows x k, ld = ldp) taken before its pivoted LU (keep_sketch)
};

template <class T>
int lowrank_compress(T* X, int ldx, int rows, int cols, double atol, double rtol, int kinit, uint64_t seed, hipStream_t s, LowRank<T>* out);
template <class T>
void lowrank_free(LowRank<T>& lr);
// hs_lowrank_batch.hip: the same compression for a batch of independent blocks (one grouped launch per stage)
template <class T>
struct LowRankJob {
  T* X;            // rows x cols, ld ldx; destroyed
  int ldx, rows, cols;
  int kinit;       // initial sketch width (<= 0: 128)
  uint64_t seed;
  LowRank<T>* out;
  int k;           // work: current sketch width
};
template <class T>
// need_z = false: only the pivoted LU of the sketch, the row permutation and the rank are wanted (the HSS module takes its
// interpolation matrices from them): Z is not formed and X is left as it was
int lowrank_compress_batch(LowRankJob<T>* jobs, int njobs, double atol, double rtol, hipStream_t s, bool need_z = true, bool keep_sketch = false, bool sketch_only = false);  // sketch_only: Y0 = X * Omega and nothing else (the caller orders the rows itself: qr_refine with norm_select)
// hs_hss.hip: the interpolative form X ~= C*Z with Z = r ROWS of X and C = P'[I; T] -- pivot order from the tournament-pivoted LU of the
// sketch, rank (tolerance-stopped) and least-squares T from the windowed-pivoted orthogonalisation of the sketch rows (qr_refine): the role
// of `pqrfact(X; atol, rtol)` (rank-revealing QR) in `_lgauss_transform` / `_rgauss_transform` (src/factorization.jl:171-182).  X is left
// untouched; out->Cd (dense, original row order), out->Z, out->r, out->rperm are set, the packed LU is released.
template <class T>
int lowrank_id_batch(LowRankJob<T>* jobs, int njobs, double atol, double rtol, hipStream_t s);
// hs_lrdense.hip
template <class T>
void lowrank_expand(LowRank<T>& lr, hipStream_t s);  // fills Cd from the trapezoid form
template <class T>
void launch_lr_dense(const T* C, int ldc, int rows, int r, const T* t, T* dst, const int* didx, hipStream_t s);
template <class T>
struct RtrsmJob {
  T* X;        // r x n (ld ldx): right-hand side, consumed
  int ldx;
  T* Xout;     // r x n (ld ldo): X * U^-1
  int ldo;
  const T* LF;   // U = upper triangle of LF[0:n, 0:n] (ld ldl)
  int ldl;
  const T* invU;  // inverted 32x32 diagonal blocks of U
  int n, r;
  const T* inv256U;  // inverted 256x256 diagonal blocks of U (ld 256), or null: the base case is then 32 columns
};
template <class T>
int rtrsm_upper_batch(const RtrsmJob<T>* jobs, int nj, hipStream_t s, void** dprobs_out);
template <class T>
int rtrsm_upper(T* X, int ldx, T* Xout, int ldo, const T* LF, int ldl, const T* invU, int n, int r, hipStream_t s, void** dprobs_out);
