/*
 * hs_kernels.h -- kernel-level test hooks of libhs_solver (used by tests/ only).
 *
 * These drive the same HIP kernels the hot path (hs_factor_* / hs_ldiv_* in hs_solver.h) launches,
 * on caller-supplied dense host data, so each kernel can be checked against the oracle alone.
 * Column-major everywhere; complex = interleaved (re, im) doubles (Julia ComplexF64).
 */
#ifndef HS_KERNELS_H
#define HS_KERNELS_H
#include <stdint.h>
#include "hs_solver.h"
#ifdef __cplusplus
extern "C" {
#endif

/* C = C - A*B (minus != 0) or C = A*B (minus == 0) with the MFMA GEMM kernel
 * (the contraction of src/factorization.jl:40,72 and src/blockmatrix.jl:97,118).
 * repeat > 0 additionally times `repeat` back-to-back launches (ms per launch in *ms_out). */
int hsk_gemm_d(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb,
               double* C, int64_t ldc, int minus, int repeat, double* ms_out);
int hsk_gemm_z(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb,
               double* C, int64_t ldc, int minus, int repeat, double* ms_out);

/* Eliminate the first ni DOFs of `count` dense fronts F[k] ((ni+nb) x (ni+nb), front order [int; bnd]):
 * the per-front work of _factor_leaf / _factor_branch (src/factorization.jl:30-42, 62-75).
 *   outLF  (ni+nb) x ni : [L\U of P*Aii ; Abi*U^-1]      outUR  ni x nb : L^-1*P*Aib
 *   outSB  nb x nb      : Abb - Abi*Aii^-1*Aib           out_rperm ni   : (P x)[i] = x[rperm[i]] (0-based)
 *   info[k]             : 0, or 1 + first column with an exactly zero pivot */
int hsk_front_factor_d(int64_t count, int64_t ni, int64_t nb, const double* F, double* outLF, double* outUR,
                       double* outSB, int64_t* out_rperm, int64_t* info, double* ms_out);
int hsk_front_factor_z(int64_t count, int64_t ni, int64_t nb, const double* F, double* outLF, double* outUR,
                       double* outSB, int64_t* out_rperm, int64_t* info, double* ms_out);

/* Low-rank compression X (rows x cols) ~= C (rows x r) * Z (r x cols) to tolerance max(atol, rtol*|u_11|): the
 * device primitive behind the compressed Gauss transforms (`_lgauss_transform` / `_rgauss_transform`,
 * src/factorization.jl:171-182, which call LowRankApprox.pqrfact).  cap = capacity (in columns of C / rows of Z)
 * of the output arrays; kinit = initial sketch width. */
int hsk_lowrank_d(int64_t rows, int64_t cols, const double* X, double atol, double rtol, int64_t kinit, int64_t seed,
                  int64_t* r_out, double* Cout, double* Zout, int64_t cap);
int hsk_lowrank_z(int64_t rows, int64_t cols, const double* X, double atol, double rtol, int64_t kinit, int64_t seed,
                  int64_t* r_out, double* Cout, double* Zout, int64_t cap);

/* Host-only: the order in which the HSS form of a front's interior block lists its DOFs (hs_options.hss_d): recursive bisection of
 * the graph of A (1-based CSC pattern colptr / rowval of the n x n matrix) restricted to the ni DOFs `ids` (1-based), split where the
 * HSS cluster tree splits its index range.  perm_out[new position] = position in `ids` (0-based). */
int hsk_bisect_perm(int64_t n, const int64_t* colptr, const int64_t* rowval, int64_t ni, const int64_t* ids, int64_t* perm_out);

/* Accounting of the GROUPED MFMA products (`gemm_probs_kernel`: every product of the HSS module, of the low-rank compressions and of the
 * matrix-free fronts -- the flops of the compressed branch that `hs_stats.gemm_flops` (the fronts' `gemm_op_kernel`) does not see).
 *   hs_probs_stats_mode(0 | 1 | 2) : off / count flops and launches / also time every launch with a HIP event pair; resets the counters
 *   hs_probs_stats(out3)           : {real flops executed (complex: 8 M N K), launches, summed launch seconds}; synchronises the device
 * Process-wide (the products are issued from several host threads and streams). */
int hs_probs_stats_mode(int mode);
int hs_probs_stats(double* out3);

/* Microseconds per ROUND TRIP (two exchanges) between workgroup 0 and workgroup `peer` of one launch through agent-scope atomic stores and polled
 * loads -- the exchange primitive of the dataflow sweeps of ldiv! (kernels_solve_wide.hip).  peer = 1: another XCD, peer = 8: the same XCD. */
double hsk_flow_pingpong_us(int peer, int iters);
/* Measured TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64 on every CU (roofline denominator). */
double hsk_mfma_f64_peak(int waves_per_simd, int iters);
/* The same issue loop on random operands that change while it runs: the rate at the clock the chip holds under such data (DVFS). */
double hsk_mfma_f64_peak_random(int waves_per_simd, int iters);

#ifdef __cplusplus
}
#endif
#endif
