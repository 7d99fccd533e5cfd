/*
 * hs_hss.h -- C ABI of the device HSS module of libhs_solver (gfx950).
 *
 * In the reference the Schur complement `S` of a compressed front and its interior block `D = Aii` are
 * `HssMatrix` objects of HssMatrices.jl (a dependency that is NOT part of the reference tree, Manifest.toml:263-269).
 * These entry points are what a Julia `ccall` shim binds in place of the HssMatrices calls the hot path makes:
 *
 *   hs_hss_compress_{d,z}  <- compress(S[perm,perm], cl, cl; atol, rtol)              src/factorization.jl:56-57
 *                             randcompress_adaptive(hssS, cl, cl; kest, atol, rtol)   src/factorization.jl:109-110
 *                             with cl = bisection_cluster((n1, n); leafsize)          src/factorization.jl:56,109
 *   hs_hss_rank            <- hssrank(F.S)                                            src/factornode.jl:53
 *   hs_hss_mul             <- `*` of an HssMatrix with a dense block                  src/factorization.jl:242, blockmatrix.jl:97
 *   hs_hss_factor / _ldiv  <- `\` with HssMatrix blocks inside blockfactor / blockldiv!  src/blockmatrix.jl:121-156
 *
 * Representation (oracle/hs_hss.py restates it on the CPU): binary cluster tree in breadth-first order (node 0 =
 * root); every non-root node keeps ONE skeleton for rows and columns -- local positions `p` (the r skeleton
 * positions first) and the interpolation matrix T ((m-r) x r): A(I[p_R], far) ~= T A(I[p_S], far),
 * A(far, I[p_R]) ~= A(far, I[p_S]) T^T (plain transpose) -- leaves keep their diagonal block D, inner nodes the
 * couplings B12 = A(sk_l, sk_r), B21 = A(sk_r, sk_l) of their children's skeletons.  hs_hss_factor eliminates the
 * redundant positions level by level (ID-based ULV elimination, "recursive skeletonization"); hs_hss_ldiv applies
 * the inverse.  Column-major everywhere; complex = interleaved (re, im) doubles (Julia ComplexF64).
 * `where`: 0 = the pointers are host memory, 1 = device memory (no PCIe traffic).
 */
#ifndef HS_HSS_H
#define HS_HSS_H
#include <stdint.h>
#include "hs_solver.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hs_hss hs_hss; /* opaque: owns the device-resident generators and factors */

typedef struct hs_hss_options {
  int64_t leafsize;    /* leaves of the cluster tree hold at most this many indices (SolverOptions.leafsize) */
  int64_t first_split; /* > 0: the root splits at this index, bisection_cluster((n1, n)); 0: in the middle */
  double atol, rtol;   /* an off-diagonal block is truncated at max(atol, rtol * |largest pivot of its samples|) */
  int64_t kest;        /* initial number of samples per side; doubled until every rank <= 0.8*samples - pad */
  int64_t pad;         /* oversampling (default 8) */
  int64_t seed;
  double level_scale;  /* tolerances of tree level l are multiplied by level_scale^(l-1) (root = level 0): deeper levels are
                          truncated more tightly so that their accumulated error stays below the truncation threshold of
                          the levels above (default 0.5; 1 = the same tolerance everywhere) */
} hs_hss_options;

void hs_hss_options_default(hs_hss_options* o); /* 64, 0, 1e-6, 1e-6, 64, 8, 123, 0.5 */

/* HSS form of the dense n x n matrix A (column-major, leading dimension lda). */
int hs_hss_compress_d(int64_t n, const double* A, int64_t lda, int where, const hs_hss_options* o, hs_hss** out);
int hs_hss_compress_z(int64_t n, const double* A, int64_t lda, int where, const hs_hss_options* o, hs_hss** out);

/* The same for H ~= A[perm, perm] (perm: n 0-based indices on the HOST, NULL = identity; products and solves keep the
 * caller's index order) on HIP stream `stream` (hipStream_t; NULL: a private stream).  The elimination uses it for the
 * interior block of a front, whose two separator layers are listed one after the other: the
 * permutation interleaves them so that an index range of the cluster tree is a compact patch of the separator. */
int hs_hss_compress_ex_d(int64_t n, const double* A, int64_t lda, int where, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out);
int hs_hss_compress_ex_z(int64_t n, const double* A, int64_t lda, int where, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out);

/* The same for an operator that is never formed: H ~= (B - C*M*Z)[perm, perm] with dense B (n x n), thin C (n x r1), small M (r1 x r2)
 * and thin Z (r2 x n) -- the Schur complement `S = Abb - Abi*R` of a compressed front as the reference compresses it, through products
 * (`_sample_schur!`, src/factorization.jl:238-244) and entries (`_getindex_schur`, :246-249) under `randcompress_adaptive` (:110).
 * r1 = 0 or r2 = 0: no update (the plain compression of B). */
int hs_hss_compress_lru_d(int64_t n, const double* B, int64_t ldb, const double* C, int64_t ldc, const double* M, int64_t ldm, const double* Z, int64_t ldz,
                          int64_t r1, int64_t r2, int where, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out);
int hs_hss_compress_lru_z(int64_t n, const double* B, int64_t ldb, const double* C, int64_t ldc, const double* M, int64_t ldm, const double* Z, int64_t ldz,
                          int64_t r1, int64_t r2, int where, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out);
/* The same for `count` matrices at once (all operands on the DEVICE): B[b] - C[b]*M[b]*Z[b] (r1[b] = 0: no update; M[b] = NULL with r1 = r2: the
   identity), perm[b] on the host or NULL, one options block per matrix (leafsize, first_split, seed, kest are its own; atol, rtol, pad,
   level_scale those of opts[0]; the sample count is common).  The compressions of the Schur complements of one tree level are independent
   chains of small dependent launches: batched, a stage of a cluster-tree level is ONE group of launches for all of them.  out[b] share the
   generators of one forest, which lives until the last of them is freed. */
int hs_hss_compress_lru_multi_d(int64_t count, const int64_t* n, const double* const* B, const int64_t* ldb, const double* const* C, const int64_t* ldc,
                                const double* const* M, const int64_t* ldm, const double* const* Z, const int64_t* ldz, const int64_t* r1, const int64_t* r2,
                                const int64_t* const* perm, const hs_hss_options* const* opts, void* stream, hs_hss** out);
int hs_hss_compress_lru_multi_z(int64_t count, const int64_t* n, const double* const* B, const int64_t* ldb, const double* const* C, const int64_t* ldc,
                                const double* const* M, const int64_t* ldm, const double* const* Z, const int64_t* ldz, const int64_t* r1, const int64_t* r2,
                                const int64_t* const* perm, const hs_hss_options* const* opts, void* stream, hs_hss** out);

/* later products / eliminations / solves run on `stream` (hipStream_t; NULL = the default stream); every call still returns only after its
 * work on that stream has completed */
int hs_hss_set_stream(hs_hss* H, void* stream);

int64_t hs_hss_rank(const hs_hss* H);      /* hssrank: largest rank of an off-diagonal block */
int64_t hs_hss_size(const hs_hss* H);      /* n */
int64_t hs_hss_samples(const hs_hss* H);   /* samples per side the adaptive compression ended with */
int64_t hs_hss_bytes(const hs_hss* H);     /* device bytes held by the generators and, once eliminated, the factors (views: 0) */
int64_t hs_hss_num_nodes(const hs_hss* H); /* nodes of the cluster tree (breadth-first numbering) */
/* out[0..7] = lo, hi (0-based half-open index range), left, right (-1: leaf), level, m (local size), r (rank), isleaf */
int hs_hss_node_info(const hs_hss* H, int64_t node, int64_t out[8]);
/* generators of one node copied to tightly packed host arrays (any pointer may be NULL):
 * p[m] (0-based local positions, skeleton first), T ((m-r) x r), D (m x m, leaves), B12 (r_l x r_r), B21 (r_r x r_l) */
int hs_hss_node_data(const hs_hss* H, int64_t node, int64_t* p, double* T, double* D, double* B12, double* B21);

/* out (ni x nj, leading dimension ldo) = H[I, J] for 0-based index lists on the HOST (in the caller's index order): the entry access an
 * operator assembled from HSS blocks is asked for (`S.A11[...]` inside `_getindex_schur`, src/factorization.jl:246-249; the blocks
 * `randcompress_adaptive` reads).  O((ni + nj) * rank) per tree level; `where` says where `out` lives. */
int hs_hss_getindex(hs_hss* H, const int64_t* I, int64_t ni, const int64_t* J, int64_t nj, double* out, int64_t ldo, int where);

/* out (size(node) x r, leading dimension ldo) = the EXPANDED basis U of one non-root node, A(I_node, far) ~= U * A(sk_node, far) (and, with
 * the plain transpose, A(far, I_node) ~= A(far, sk_node) * U^T): `generators(S.A11)` and the factors `U*B12`, `V` of a child's off-diagonal
 * blocks that a parent front assembles its low-rank couplings from (src/factorization.jl:129-137).  Nodes 1 and 2 are the two halves of the
 * top-level split: A12 = U_1 * B12 * U_2^T, A21 = U_2 * B21 * U_1^T with the root's B12, B21 (hs_hss_node_data). */
int hs_hss_basis(hs_hss* H, int64_t node, double* out, int64_t ldo, int where);
/* `Matrix(H)`: out (n x n, column-major, ldo >= n; where = 0 host / 1 device) = the matrix H represents in H's own index order
   (a compressed matrix: A[perm, perm]; the views of hs_hss_child carry no permutation).  2 n^2 r flops. */
int hs_hss_expand(hs_hss* H, double* out, int64_t ldo, int where);

/* Y = H * X for n x nrhs blocks */
int hs_hss_mul(hs_hss* H, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t nrhs, int where);
/* Y = H^T * X (plain transpose; the samples `X^T A` of an operator that contains H) */
int hs_hss_mul_t(hs_hss* H, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t nrhs, int where);
/* `H.A11` (which = 0) / `H.A22` (which = 1) of the top-level split as an HSS matrix that SHARES H's generators (H must outlive it;
 * which = 2: all of H as a view in cluster-tree order, i.e. without H's permutation):
 * what `_assemble_blocks` reads from a child's Schur complement (src/factorization.jl:127-135).  Its index space is the block's own,
 * 0 .. size-1 in cluster-tree order.  HS_ERR_HSS_LEAF when H is a single leaf (factorization.jl:164). */
int hs_hss_child(hs_hss* H, int which, hs_hss** out);
/* One contiguous DEVICE buffer for a whole HSS matrix: what crosses ranks at a join of the elimination tree when a child's Schur complement
 * travels as an HssMatrix (src/factorization.jl:78-112,126-140; SURVEY.md 8(e): "ship HSS generators instead of dense S").
 *   hs_hss_pack_size : bytes the packed form takes
 *   hs_hss_pack      : writes it to dev_buf (device pointer, at least that many bytes) on `stream` (NULL: the matrix's own); returns when done
 *   hs_hss_unpack    : a new matrix from a packed buffer (device pointer; it is copied, the caller keeps dev_buf); is_complex must match
 * Generators, permutation, cluster tree and options travel; the factors of hs_hss_factor do not (the receiver eliminates again if it needs to). */
int hs_hss_pack_size(const hs_hss* H, int64_t* bytes);
int hs_hss_pack(hs_hss* H, void* dev_buf, int64_t bytes, void* stream);
int hs_hss_unpack(const void* dev_buf, int64_t bytes, int is_complex, void* stream, hs_hss** out);
/* Order in which the rank-revealing orthogonalisation inside every compression takes its rows (the role of `pqrfact`,
 * src/factorization.jl:171-182, a column-pivoted QR stopped at the tolerance):
 *   0 (default)  the pivot order of a tournament-pivoted LU of a sketch, then windowed pivoted Cholesky-QR with re-orthogonalisation
 *   1            no LU: a blocked column-pivoted QR -- every window is the rows of largest residual norm (downdated, recomputed when they
 *                have lost digits), and a pivot is accepted only while it is at least HS_QR_THETA (0.5) times every residual outside the
 *                window: the greedy order of `pqrfact` up to that factor.  Same ranks and errors as 0 (tests/test_hss_gpu.py), 1.5-2x the
 *                time of the compressions; it exists to check 0 against.
 * mode < 0 only reads.  Returns the previous mode.  Process-wide; the environment variable HS_QR_ORDER=norm sets the initial value. */
int hs_hss_qr_order(int mode);
/* The off-diagonal blocks of the top-level split in low-rank form, A12 = C*Z (which = 0: C = U_1*B12 is n1 x r2, Z = U_2^T is r2 x n2) or
 * A21 = C*Z (which = 1: C = U_2*B21 is n2 x r1, Z = U_1^T is r1 x n1): the factors `Uint = generators(S.A11)[1]*S.B12`, `Vbnd` that a parent
 * front takes its low-rank couplings Aib, Abi from (src/factorization.jl:129-137).  Sizes and ranks: hs_hss_node_info of nodes 1 and 2. */
int hs_hss_offdiag(hs_hss* H, int which, double* C, int64_t ldc, double* Z, int64_t ldz, int where);

/* ---- operators that are never formed: two diagonal HSS blocks + sparse couplings (device only) -------------------------------------------
 * The reference assembles the blocks of a compressed branch from its children's HSS Schur complements without densifying them
 * (`_assemble_blocks`, src/factorization.jl:126-140): Aii = [S1.A11  A[int1,int2]; A[int2,int1]  S2.A11], Abb = [S1.A22  A[bnd1,bnd2];
 * A[bnd2,bnd1]  S2.A22], and compresses `S = P(Abb - Abi*R)P'` from products (`_sample_schur!`, :238-244) and entries
 * (`_getindex_schur`, :246-249).  hs_hss_blockop describes such an operator Op = [H1  A[g1,g2]; A[g2,g1]  H2] on the device. */
typedef struct hs_sparse_dev { /* the sparse matrix A on the DEVICE, 0-based, as CSC and as CSR (both gather-form products) */
  int64_t n;
  const int64_t* colptr; const int32_t* rowval; const void* nzval;   /* CSC */
  const int64_t* rowptr; const int32_t* colind; const void* nzval_r; /* CSR of the same matrix */
} hs_sparse_dev;
typedef struct hs_hss_blockop {
  int64_t n1, n2;      /* sizes of the two parts; the operator's index space is [part 1; part 2] */
  hs_hss* H1;          /* diagonal blocks (hs_hss_child views of the children's Schur complements); NULL for an empty part */
  hs_hss* H2;
  const int64_t* gid;  /* HOST: 0-based global DOF id (row/column of A) of every index, n1 + n2 entries */
  const hs_sparse_dev* A;
  int32_t* lpos;       /* DEVICE scratch of A->n ints, all -1 on entry; restored to -1 on return */
} hs_hss_blockop;
/* H ~= (Op - C*M*Z)[perm, perm] (C, M, Z on the device; r1 = 0 or r2 = 0: no update; M = NULL with r1 = r2: M is the identity;
   perm on the host or NULL) */
int hs_hss_compress_blockop_d(const hs_hss_blockop* op, const double* C, int64_t ldc, const double* M, int64_t ldm, const double* Z, int64_t ldz,
                              int64_t r1, int64_t r2, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out);
int hs_hss_compress_blockop_z(const hs_hss_blockop* op, const double* C, int64_t ldc, const double* M, int64_t ldm, const double* Z, int64_t ldz,
                              int64_t r1, int64_t r2, const int64_t* perm, const hs_hss_options* o, void* stream, hs_hss** out);
/* Y = Op*X (trans != 0: Op^T*X) on device blocks of nrhs columns, in the operator's index order */
int hs_hss_blockop_apply(const hs_hss_blockop* op, int is_complex, const double* X, int64_t ldx, double* Y, int64_t ldy, int64_t nrhs, int trans, void* stream);

/* `prune_leaves!`, `compatible`, `depth` -- the operations of `_equilibrate_clusters` (src/factorization.jl:143-168), which makes the cluster trees
 * of the two children's `S.A11` structurally equal before HSS-by-HSS arithmetic.  hs_hss_prune_leaves: every node whose two children are leaves
 * becomes a leaf (D = [D_l  U_l B12 U_r^T; U_r B21 U_l^T  D_r]; its basis re-expressed over the merged leaf); the result is a view that shares
 * the untouched generators of H (H must outlive it) and represents the SAME matrix.  HS_ERR_HSS_LEAF when H is a single leaf (:163-165).
 * hs_hss_compatible: 1 when the two trees have the same shape.  The elimination itself never needs them here (every compression samples an
 * operator, hs_mffront.h); they are provided for hosts that do HSS-by-HSS arithmetic on the handed-over Schur complements. */
int hs_hss_prune_leaves(hs_hss* H, hs_hss** out);
int hs_hss_compatible(const hs_hss* A, const hs_hss* B);
int64_t hs_hss_depth(const hs_hss* H); /* levels of the cluster tree (1 = a single leaf) */

/* ULV-type elimination of the HSS matrix (once), then B <- H^-1 B in place */
int hs_hss_factor(hs_hss* H);
int hs_hss_ldiv(hs_hss* H, double* B, int64_t ldb, int64_t nrhs, int where);
/* wall time of the last compress / factor on the device (seconds, host clock around a synchronised stream) */
double hs_hss_time(const hs_hss* H, int what); /* 0: compress, 1: factor */
/* The module recycles its device blocks (and those of the low-rank compressions of hs_factor_*) through process-wide caches, up to
   24 + 16 GiB: hs_hss_trim gives them back to the driver and returns the device bytes released.  Call it with nothing in flight. */
int64_t hs_hss_trim(void);

/* `F.S` of one front of a factorization (include/hs_solver.h; needs hs_options.keep_schur) as an HSS matrix:
 * compress(S[perm,perm], cl, cl; atol, rtol), perm = [nd_loc.int; nd_loc.bnd], cl = bisection_cluster((|nd_loc.int|, |nd.bnd|))
 * (src/factorization.jl:56-57,109-110).  o == NULL: leafsize, atol, rtol, kest, seed of the factorization's options. */
int hs_node_schur_hss(const hs_handle* F, int64_t node, const hs_hss_options* o, hs_hss** out);

void hs_hss_free(hs_hss* H);

#ifdef __cplusplus
}
#endif
#endif
