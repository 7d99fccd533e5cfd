/* C restatement of HierarchicalSolvers.jl's dense nested-dissection elimination: factor(A, nd, nd_loc; swlevel = 0) and ldiv!(F, b).
 *
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  Nothing under hierarchicalsolvers.jl_amd/ links, loads or calls this file; only tests/ and the
 * `cpu_baseline` leg of bench.py do (through oracle/hs_oracle_c.py).  It is the CPU baseline SURVEY.md section 8(d) names: "the build's C++ CPU
 * restatement of the same algorithm (LAPACK-free or linked to whatever BLAS the box has)".
 *
 * What it follows, statement by statement (file:line of /root/reference/src in the comments of hs_oracle_c_body.h):
 *   factorization.jl:14-27   _factor            the recursion (here: a loop over the nodes in post-order)
 *   factorization.jl:30-42   _factor_leaf       D, L = Abi / D, R = D \ Aib, S = Abb - Abi*R, S[perm, perm]
 *   factorization.jl:62-75   _factor_branch     _assemble_blocks (:115-123), blockfactor, blockrdiv, blockldiv, S = Abb - Abi*R
 *   blockmatrix.jl:115-187   blockfactor / blockldiv! / blockldiv / blockrdiv
 *   factornode.jl:62-99      ldiv!, _lsolve, _dsolve, _rsolve
 * including what makes the reference slow: every `\` and `/` on a Matrix is a fresh LU (13 getrf per dense branch, 3 more per branch and one per
 * leaf in every ldiv!), and L and R are formed explicitly.  The executed flops are counted call by call and returned.
 *
 * Dense kernels: the host's BLAS / LAPACK through function pointers (Fortran calling convention, 32-bit integers: what
 * scipy.linalg.cython_blas / cython_lapack export -- the OpenBLAS the image ships with SciPy, i.e. the same library class Julia's LinearAlgebra
 * would call), or, with a NULL table, the plain blocked loops in hs_oracle_c_body.h (OpenMP over columns): "LAPACK-free".
 *
 * Parity pinning: PARITY UNPINNED against the Julia reference (no Julia runtime, no golden vectors in the reference).  Pinned
 * instead against the NumPy restatement oracle/hs_oracle.py node by node (|S_i|_F of every front, the solution) and against SuperLU
 * (tests/test_oracle_c.py), on real and complex problems, with both kernel tables.
 *
 * Build: oracle/Makefile (gcc -O2 -fopenmp -shared -fPIC) -> oracle/_build/libhs_oracle_c.so */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef double _Complex zcplx;

typedef struct {
  void (*dgemm)(char*, char*, int*, int*, int*, double*, double*, int*, double*, int*, double*, double*, int*);
  void (*dgetrf)(int*, int*, double*, int*, int*, int*);
  void (*dgetrs)(char*, int*, int*, double*, int*, int*, double*, int*, int*);
  void (*zgemm)(char*, char*, int*, int*, int*, zcplx*, zcplx*, int*, zcplx*, int*, zcplx*, zcplx*, int*);
  void (*zgetrf)(int*, int*, zcplx*, int*, int*, int*);
  void (*zgetrs)(char*, int*, int*, zcplx*, int*, int*, zcplx*, int*, int*);
} hsc_blas;

/* The elimination tree with its index sets, nodes in POST-ORDER (children before parents, the root last), everything 0-based:
 *   left / right   children (-1: leaf)
 *   int, bnd       global DOF ids of the node's interior / boundary (NDNode.int, .bnd, nesteddissection.jl:7-17, after symfact! and permuted!)
 *   li, lb         the node's entry of the parallel tree `nd_loc` (nesteddissection.jl:35-69): positions inside its own `bnd` of the DOFs
 *                  that are interior / boundary for its PARENT */
typedef struct {
  int nnodes;
  const int* left;
  const int* right;
  const int64_t *int_ptr, *int_idx, *bnd_ptr, *bnd_idx, *li_ptr, *li_idx, *lb_ptr, *lb_idx;
} hsc_tree;

static double g_flops = 0.0, g_bytes = 0.0;
static long g_getrf = 0, g_singular = 0;

static double hsc_now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct { int64_t n; const int64_t* colptr; const int64_t* rowidx; const double* vals; } Csc_d;
typedef struct { int64_t n; const int64_t* colptr; const int64_t* rowidx; const zcplx* vals; } Csc_z;

#define T double
#define NAME(x) x##_d
#define BLASF(x) d##x
#define ABSF(v) fabs(v)
#define FLOPMUL 1.0
#include "hs_oracle_c_body.h"
#undef T
#undef NAME
#undef BLASF
#undef ABSF
#undef FLOPMUL

#define T zcplx
#define NAME(x) x##_z
#define BLASF(x) z##x
#define ABSF(v) cabs(v)
#define FLOPMUL 4.0
#include "hs_oracle_c_body.h"
#undef T
#undef NAME
#undef BLASF
#undef ABSF
#undef FLOPMUL

/* dense checks of the kernels themselves (tests/test_oracle_c.py: the plain loops against the BLAS table) */
int hsc_selftest_solve_d(const hsc_blas* bl, int n, int nrhs, const double* A, const double* B, double* X, double* Xr) {
  double* x = ldiv_new_d(bl, A, n, n, B, n, nrhs);
  memcpy(X, x, sizeof(double) * (size_t)n * (size_t)nrhs);
  free(x);
  x = rdiv_new_d(bl, B, nrhs, nrhs, A, n, n); /* B read as nrhs x n here */
  memcpy(Xr, x, sizeof(double) * (size_t)n * (size_t)nrhs);
  free(x);
  return 0;
}
int hsc_selftest_solve_z(const hsc_blas* bl, int n, int nrhs, const zcplx* A, const zcplx* B, zcplx* X, zcplx* Xr) {
  zcplx* x = ldiv_new_z(bl, A, n, n, B, n, nrhs);
  memcpy(X, x, sizeof(zcplx) * (size_t)n * (size_t)nrhs);
  free(x);
  x = rdiv_new_z(bl, B, nrhs, nrhs, A, n, n);
  memcpy(Xr, x, sizeof(zcplx) * (size_t)n * (size_t)nrhs);
  free(x);
  return 0;
}
