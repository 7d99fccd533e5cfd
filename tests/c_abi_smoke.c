/* c_abi_smoke.c -- the drop-in boundary exercised from plain C, with exactly the argument layout of the Julia shim in
 * INTEGRATION.md (no Python, no ctypes): 1-based Int64 CSC fields of a SparseMatrixCSC{ComplexF64,Int}, the values as
 * interleaved (re, im) doubles behind a typed pointer, the flat tree of (nd, nd_loc), the POD mirror of SolverOptions
 * (include/hs_solver.h), `ldiv!` in place (C === B), `maxrank`, the error path (status code + hs_last_error) and hs_free.
 *
 * Scenario = test/rungmres.jl:15-19,32 of the reference on a synthetic 2-D Helmholtz problem: tree -> symfact! -> postorder ->
 * permute -> factor(A, nd, nd_loc; swlevel=0) -> ldiv!(F, b), the symbolic steps through include/hs_symbolic.h.
 * The library is opened with dlopen, as Julia's ccall does.  Built by __graft_entry__.build(); run by tests/test_c_abi_gpu.py.
 *
 *   c_abi_smoke <path/to/libhs_solver.so> [grid points per side, default 40]
 * prints "C_ABI_SMOKE OK ..." and exits 0 when every residual is below 1e-10; exits 2 when the library reports no device. */
#define _GNU_SOURCE
#include <complex.h>
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "hs_solver.h"
#include "hs_symbolic.h"

#define LOAD(name)                                                  \
  do {                                                              \
    *(void**)(&p_##name) = dlsym(lib, #name);                       \
    if (!p_##name) {                                                \
      fprintf(stderr, "symbol %s missing: %s\n", #name, dlerror()); \
      return 1;                                                     \
    }                                                               \
  } while (0)

static void (*p_hs_options_default)(hs_options*);
static int (*p_hs_factor_z)(int64_t, const int64_t*, const int64_t*, const double*, const hs_tree*, const hs_options*, hs_handle**);
static int (*p_hs_factor_d)(int64_t, const int64_t*, const int64_t*, const double*, const hs_tree*, const hs_options*, hs_handle**);
static int (*p_hs_ldiv_z)(hs_handle*, double*, int64_t, const double*, int64_t, int64_t, int64_t);
static int (*p_hs_ldiv_d)(hs_handle*, double*, int64_t, const double*, int64_t, int64_t, int64_t);
static int64_t (*p_hs_maxrank)(const hs_handle*);
static int (*p_hs_is_complex)(const hs_handle*);
static int64_t (*p_hs_size)(const hs_handle*);
static void (*p_hs_free)(hs_handle*);
static const char* (*p_hs_last_error)(void);
static int (*p_hs_device_info)(char*, int64_t, int64_t*, int64_t*);
static int (*p_hs_symbolic_from_graph)(int64_t, const int64_t*, const int64_t*, int64_t, hs_symbolic**);
static const int64_t* (*p_hs_symbolic_perm)(const hs_symbolic*);
static int (*p_hs_symbolic_tree)(const hs_symbolic*, hs_tree*);
static void (*p_hs_symbolic_free)(hs_symbolic*);
static int (*p_hs_comm_unique_id)(void*);
static int (*p_hs_comm_create_rccl)(const void*, int64_t, int64_t, hs_comm**);
static int (*p_hs_comm_create_host)(hs_transfer_fn, void*, int64_t, int64_t, hs_comm**);
static void (*p_hs_comm_free)(hs_comm*);
static const char* (*p_hs_comm_kind)(const hs_comm*);
static int (*p_hs_comm_selftest)(hs_comm*, int64_t);
static int (*p_hs_set_comm)(hs_handle*, hs_comm*);

/* hs_transfer_fn of a one-rank host: the only peer is the rank itself -- every send is the matching receive */
static int self_transfer(void* user, int64_t nsend, const int64_t* send_peer, void* const* send_buf, const int64_t* send_bytes, int64_t nrecv,
                         const int64_t* recv_peer, void* const* recv_buf, const int64_t* recv_bytes) {
  int64_t* calls = (int64_t*)user;
  ++*calls;
  if (nsend != nrecv) return 1;
  for (int64_t k = 0; k < nsend; ++k) {
    if (send_peer[k] != 0 || recv_peer[k] != 0 || send_bytes[k] != recv_bytes[k]) return 2;
    memcpy(recv_buf[k], send_buf[k], (size_t)send_bytes[k]);
  }
  return 0;
}

/* 5-point Helmholtz on an N x N grid, CSC, 1-based: A = -Laplace_h - k^2 + i*sigma (complex symmetric, not Hermitian) */
static void build_matrix(int N, int64_t** colptr, int64_t** rowval, double complex** nz) {
  const int64_t n = (int64_t)N * N;
  *colptr = malloc(sizeof(int64_t) * (n + 1));
  *rowval = malloc(sizeof(int64_t) * 5 * n);
  *nz = malloc(sizeof(double complex) * 5 * n);
  const double h = 1.0 / (N + 1), k = 2.0 * M_PI * 2.5;
  int64_t e = 0;
  for (int j = 0; j < N; ++j)
    for (int i = 0; i < N; ++i) {
      const int64_t c = (int64_t)j * N + i;
      (*colptr)[c] = e + 1;
      /* rows in ascending order: (i, j-1), (i-1, j), (i, j), (i+1, j), (i, j+1) */
      if (j > 0) { (*rowval)[e] = c - N + 1; (*nz)[e++] = -1.0; }
      if (i > 0) { (*rowval)[e] = c - 1 + 1; (*nz)[e++] = -1.0; }
      (*rowval)[e] = c + 1;
      (*nz)[e++] = 4.0 - k * k * h * h + 0.3 * I * k * h;
      if (i < N - 1) { (*rowval)[e] = c + 1 + 1; (*nz)[e++] = -1.0; }
      if (j < N - 1) { (*rowval)[e] = c + N + 1; (*nz)[e++] = -1.0; }
    }
  (*colptr)[n] = e + 1;
}

/* B = A[perm, perm] (1-based perm), rows of every column sorted: what `permute(A, perm, perm)` returns */
static void permute_csc(int64_t n, const int64_t* cp, const int64_t* rv, const double complex* nz, const int64_t* perm, int64_t** cpo, int64_t** rvo,
                        double complex** nzo) {
  int64_t* inv = malloc(sizeof(int64_t) * n);
  for (int64_t i = 0; i < n; ++i) inv[perm[i] - 1] = i;
  const int64_t nnz = cp[n] - 1;
  *cpo = malloc(sizeof(int64_t) * (n + 1));
  *rvo = malloc(sizeof(int64_t) * nnz);
  *nzo = malloc(sizeof(double complex) * nnz);
  int64_t e = 0;
  for (int64_t j = 0; j < n; ++j) {
    const int64_t oj = perm[j] - 1, e0 = e;
    (*cpo)[j] = e + 1;
    for (int64_t a = cp[oj] - 1; a < cp[oj + 1] - 1; ++a) {
      (*rvo)[e] = inv[rv[a] - 1] + 1;
      (*nzo)[e++] = nz[a];
    }
    for (int64_t a = e0 + 1; a < e; ++a) { /* insertion sort of the column (<= 5 entries) */
      const int64_t r = (*rvo)[a];
      const double complex v = (*nzo)[a];
      int64_t b = a - 1;
      while (b >= e0 && (*rvo)[b] > r) {
        (*rvo)[b + 1] = (*rvo)[b];
        (*nzo)[b + 1] = (*nzo)[b];
        --b;
      }
      (*rvo)[b + 1] = r;
      (*nzo)[b + 1] = v;
    }
  }
  (*cpo)[n] = e + 1;
  free(inv);
}

static double residual(int64_t n, const int64_t* cp, const int64_t* rv, const double complex* nz, const double complex* x, const double complex* b) {
  double complex* r = malloc(sizeof(double complex) * n);
  for (int64_t i = 0; i < n; ++i) r[i] = -b[i];
  for (int64_t j = 0; j < n; ++j)
    for (int64_t a = cp[j] - 1; a < cp[j + 1] - 1; ++a) r[rv[a] - 1] += nz[a] * x[j];
  double nr = 0.0, nb = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    nr += creal(r[i]) * creal(r[i]) + cimag(r[i]) * cimag(r[i]);
    nb += creal(b[i]) * creal(b[i]) + cimag(b[i]) * cimag(b[i]);
  }
  free(r);
  return sqrt(nr / nb);
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s <libhs_solver.so> [N]\n", argv[0]);
    return 1;
  }
  const int N = argc > 2 ? atoi(argv[2]) : 40;
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "dlopen failed: %s\n", dlerror());
    return 1;
  }
  LOAD(hs_options_default); LOAD(hs_factor_z); LOAD(hs_factor_d); LOAD(hs_ldiv_z); LOAD(hs_ldiv_d); LOAD(hs_maxrank); LOAD(hs_is_complex);
  LOAD(hs_size); LOAD(hs_free); LOAD(hs_last_error); LOAD(hs_device_info); LOAD(hs_symbolic_from_graph); LOAD(hs_symbolic_perm);
  LOAD(hs_symbolic_tree); LOAD(hs_symbolic_free);
  LOAD(hs_comm_unique_id); LOAD(hs_comm_create_rccl); LOAD(hs_comm_create_host); LOAD(hs_comm_free); LOAD(hs_comm_kind); LOAD(hs_comm_selftest); LOAD(hs_set_comm);

  /* the option struct crosses the ABI with the reference's defaults (HierarchicalSolvers.jl:43-54) */
  hs_options o;
  p_hs_options_default(&o);
  if (o.swlevel != 5 || o.swsize != 1 || o.atol != 1e-6 || o.rtol != 1e-6 || o.c_tol != 0.5 || o.leafsize != 32 || o.kest != -1 || o.stepsize != 10 ||
      o.verbose != 0) {
    fprintf(stderr, "hs_options_default does not return SolverOptions()'s defaults\n");
    return 1;
  }
  char arch[64] = "";
  int64_t cus = 0, hbm = 0;
  if (p_hs_device_info(arch, sizeof arch, &cus, &hbm) <= 0) {
    fprintf(stderr, "no device: %s\n", p_hs_last_error());
    return 2;
  }

  const int64_t n = (int64_t)N * N;
  int64_t *cp0, *rv0, *cp, *rv;
  double complex *nz0, *nz;
  build_matrix(N, &cp0, &rv0, &nz0);
  hs_symbolic* S = NULL;
  int st = p_hs_symbolic_from_graph(n, cp0, rv0, 60, &S);
  if (st != HS_OK) {
    fprintf(stderr, "hs_symbolic_from_graph: %d %s\n", st, p_hs_last_error());
    return 1;
  }
  const int64_t* perm = p_hs_symbolic_perm(S);
  hs_tree tree;
  p_hs_symbolic_tree(S, &tree);
  permute_csc(n, cp0, rv0, nz0, perm, &cp, &rv, &nz);

  /* error path first: ArgumentError of chkopts! (HierarchicalSolvers.jl:74-78) -> HS_ERR_ARGUMENT + message, no handle */
  hs_handle* F = (hs_handle*)0x1;
  hs_options bad = o;
  bad.swsize = 0;
  st = p_hs_factor_z(n, cp, rv, (const double*)nz, &tree, &bad, &F);
  if (st != HS_ERR_ARGUMENT || F != NULL || strlen(p_hs_last_error()) == 0) {
    fprintf(stderr, "bad options: status %d, handle %p, message '%s'\n", st, (void*)F, p_hs_last_error());
    return 1;
  }

  /* factor(A, nd, nd_loc; swlevel=0): ComplexF64 values passed as the array Julia holds (interleaved re, im) */
  o.swlevel = 0;
  st = p_hs_factor_z(n, cp, rv, (const double*)nz, &tree, &o, &F);
  if (st != HS_OK) {
    fprintf(stderr, "hs_factor_z: %d %s\n", st, p_hs_last_error());
    return st == HS_ERR_DEVICE ? 2 : 1;
  }
  if (!p_hs_is_complex(F) || p_hs_size(F) != n || p_hs_maxrank(F) != 0) {
    fprintf(stderr, "eltype / size / maxrank of the exact factorization are wrong\n");
    return 1;
  }
  /* ldiv!(F, B) with two right-hand sides, in place (C === B), leading dimension > n like a view into a larger Matrix */
  const int64_t ld = n + 3, nrhs = 2;
  double complex* B = calloc((size_t)ld * nrhs, sizeof(double complex));
  double complex* B0 = calloc((size_t)ld * nrhs, sizeof(double complex));
  unsigned long long sd = 88172645463325252ull;
  for (int64_t c = 0; c < nrhs; ++c)
    for (int64_t i = 0; i < n; ++i) {
      sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17;
      const double u = (double)(sd >> 11) / 9007199254740992.0 - 0.5;
      sd ^= sd << 13; sd ^= sd >> 7; sd ^= sd << 17;
      const double v = (double)(sd >> 11) / 9007199254740992.0 - 0.5;
      B0[i + c * ld] = B[i + c * ld] = u + v * I;
    }
  st = p_hs_ldiv_z(F, (double*)B, ld, (const double*)B, ld, n, nrhs);
  if (st != HS_OK) {
    fprintf(stderr, "hs_ldiv_z: %d %s\n", st, p_hs_last_error());
    return 1;
  }
  double worst = 0.0;
  for (int64_t c = 0; c < nrhs; ++c) worst = fmax(worst, residual(n, cp, rv, nz, B + c * ld, B0 + c * ld));
  /* a Float64 right-hand side entry point on a ComplexF64 factorization is a MethodError in Julia: HS_ERR_ARGUMENT here */
  st = p_hs_ldiv_d(F, (double*)B, ld, (const double*)B, ld, n, 1);
  if (st != HS_ERR_ARGUMENT) {
    fprintf(stderr, "hs_ldiv_d on a complex factorization returned %d\n", st);
    return 1;
  }
  p_hs_free(F);
  p_hs_free(NULL); /* finalizers may see a null handle */

  /* the real entry points with the real part of the same matrix shifted to be definite */
  double* nzr = malloc(sizeof(double) * (size_t)(cp[n] - 1));
  for (int64_t e = 0; e < cp[n] - 1; ++e) nzr[e] = creal(nz[e]);
  for (int64_t j = 0; j < n; ++j)
    for (int64_t a = cp[j] - 1; a < cp[j + 1] - 1; ++a)
      if (rv[a] - 1 == j) nzr[a] = 4.0;
  st = p_hs_factor_d(n, cp, rv, nzr, &tree, &o, &F);
  if (st != HS_OK) {
    fprintf(stderr, "hs_factor_d: %d %s\n", st, p_hs_last_error());
    return 1;
  }
  double* x = malloc(sizeof(double) * n);
  double* b = malloc(sizeof(double) * n);
  for (int64_t i = 0; i < n; ++i) b[i] = x[i] = creal(B0[i]);
  st = p_hs_ldiv_d(F, x, n, x, n, n, 1);
  if (st != HS_OK) {
    fprintf(stderr, "hs_ldiv_d: %d %s\n", st, p_hs_last_error());
    return 1;
  }
  double nr = 0.0, nb = 0.0;
  {
    double* r = malloc(sizeof(double) * n);
    for (int64_t i = 0; i < n; ++i) r[i] = -b[i];
    for (int64_t j = 0; j < n; ++j)
      for (int64_t a = cp[j] - 1; a < cp[j + 1] - 1; ++a) r[rv[a] - 1] += nzr[a] * x[j];
    for (int64_t i = 0; i < n; ++i) { nr += r[i] * r[i]; nb += b[i] * b[i]; }
    free(r);
  }
  const double res_d = sqrt(nr / nb);
  /* the communicator part of the ABI ("Multi-GPU hosts") from C, at one rank: the host-staged transport through a C
   * callback, the RCCL transport from a fresh id; both must pass the library's self-test and attach to the handle */
  {
    hs_comm* hc = NULL;
    int64_t calls = 0;
    st = p_hs_comm_create_host(self_transfer, &calls, 0, 1, &hc);
    if (st != HS_OK || !hc || strcmp(p_hs_comm_kind(hc), "host") != 0 || p_hs_comm_selftest(hc, 100000) != HS_OK || calls < 1 || p_hs_set_comm(F, hc) != HS_OK) {
      fprintf(stderr, "host-staged communicator: %d '%s' (callback ran %lld times)\n", st, p_hs_last_error(), (long long)calls);
      return 1;
    }
    p_hs_set_comm(F, NULL);
    p_hs_comm_free(hc);
    unsigned char id[128];
    hs_comm* rc = NULL;
    st = p_hs_comm_unique_id(id);
    if (st == HS_OK) st = p_hs_comm_create_rccl(id, 0, 1, &rc);
    if (st != HS_OK || !rc || strcmp(p_hs_comm_kind(rc), "rccl") != 0 || p_hs_comm_selftest(rc, 1 << 20) != HS_OK) {
      fprintf(stderr, "RCCL communicator: %d '%s'\n", st, p_hs_last_error());
      return 1;
    }
    hs_comm* bad = NULL;
    if (p_hs_comm_create_rccl(id, 3, 2, &bad) != HS_ERR_ARGUMENT || bad != NULL) {  /* rank outside 0:nranks-1 */
      fprintf(stderr, "hs_comm_create_rccl accepted rank 3 of 2\n");
      return 1;
    }
    p_hs_comm_free(rc);
  }
  p_hs_free(F);
  p_hs_symbolic_free(S);
  printf("C_ABI_SMOKE %s n=%lld arch=%s cus=%lld  residual z=%.2e d=%.2e\n", (worst < 1e-10 && res_d < 1e-10) ? "OK" : "FAIL", (long long)n, arch,
         (long long)cus, worst, res_d);
  free(cp0); free(rv0); free(nz0); free(cp); free(rv); free(nz); free(B); free(B0); free(nzr); free(x); free(b);
  return (worst < 1e-10 && res_d < 1e-10) ? 0 : 1;
}
