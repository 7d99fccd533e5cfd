/*
 * hs_solver.h -- C ABI of the MI355X-native nested-dissection elimination.
 *
 * Drop-in boundary for ONE hot path of bonevbs/HierarchicalSolvers.jl:
 *
 *   factor(A::SparseMatrixCSC{T}, nd, nd_loc, opts; kw...) -> FactorNode{T}
 *                                      (reference src/factorization.jl:5-11)
 *   ldiv!(C, F, B), ldiv!(F, B)        (reference src/factornode.jl:62-74)
 *   maxrank(F)                         (reference src/factornode.jl:49-57)
 *
 * The reference has no FFI of its own (it is pure Julia); these entry points
 * are what a Julia `ccall` shim for that path binds (INTEGRATION.md shows the
 * shim).  Plain pointers and sizes only; all index arrays are 1-based int64
 * exactly as the Julia host holds them (SparseMatrixCSC.colptr/rowval, the
 * index vectors of the NestedDissection trees), converted inside the library.
 *
 * Threading: a call blocks the calling thread.  One handle must not be used
 * from two threads at once; distinct handles may.  hs_free is idempotent per
 * handle pointer value being NULL-safe and may be called from any thread
 * (Julia finalizer).  Errors: every int-returning function returns HS_OK (0)
 * or a negative hs_status; hs_last_error() returns the thread-local message.
 */
#ifndef HS_SOLVER_H
#define HS_SOLVER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum hs_status {
  HS_OK = 0,
  HS_ERR_ARGUMENT = -1,  /* Julia ArgumentError   (HierarchicalSolvers.jl:74-78, nesteddissection.jl:111) */
  HS_ERR_DIMENSION = -2, /* Julia DimensionMismatch (blockmatrix.jl:13-16,116-117; nesteddissection.jl:107) */
  HS_ERR_TREE = -3,      /* ErrorException "Expected nested dissection to be a binary tree..." (factorization.jl:25) */
  HS_ERR_SINGULAR = -4,  /* LinearAlgebra.SingularException raised by `\` in the reference; info = node id */
  HS_ERR_HSS_LEAF = -5,  /* error("One of the Schur complements turned into a leaf. Aborting.") (factorization.jl:164) */
  HS_ERR_DEVICE = -6,    /* HIP runtime failure / no gfx950 device: the library never falls back to the CPU */
  HS_ERR_NOMEM = -7,     /* device or host allocation failed */
  HS_ERR_UNSUPPORTED = -8
} hs_status;

/* POD mirror of `mutable struct SolverOptions` (HierarchicalSolvers.jl:30-40), same order,
 * followed by extension fields (zero = default). */
typedef struct hs_options {
  int64_t swlevel;  /* compress the top `swlevel` levels; <0 counts from the leaves (factorization.jl:8) */
  int64_t swsize;   /* minimum |bnd| for compression */
  double atol;
  double rtol;
  double c_tol;     /* validated, never consumed by the reference (factorization.jl:97-100) */
  int64_t leafsize; /* HSS leaf size */
  int64_t kest;     /* rank estimate for randomized compression; <0 => ceil(rank(L)/2) (factorization.jl:102-104) */
  int64_t stepsize; /* validated, never consumed by the reference */
  uint8_t verbose;
  /* ---- extensions ---- */
  uint8_t keep_schur; /* debug: retain every node's Schur complement S for hs_node_export */
  uint8_t profile;    /* time every kernel launch with HIP events (fills hs_stats.t_gemm etc.; adds launch gaps) */
  uint8_t split;      /* slice width, in units of 256 columns, in which the interior block of a compressed front (level <=
                         swlevel, ni >= 2 slices) is eliminated: the role of the 2x2 BlockFactorization of D (blockmatrix.jl:106-130);
                         0 = off; single-rank factorizations only */
  uint8_t hss_d;      /* > 0: a front at a level <= swlevel whose interior block has at least hss_d*1024 DOFs keeps D = Aii as an HSS
                         matrix (include/hs_hss.h) instead of a dense LU -- the role of `D::BlockFactorization` over HssMatrix blocks
                         (blockmatrix.jl:121-130, factorization.jl:86-96); the root included, which in the reference receives its
                         children's HSS blocks although it is never flagged (factorization.jl:15,67,126).  Single rank only.  With such fronts
                         hs_ldiv_dev_* and hs_solve_*_levels return only after the HSS solves on the given stream have completed
                         (the HSS module recycles its workspaces per call). */
  uint8_t hss_dexp;   /* tolerance of that HSS form: atol, rtol * 10^-e with e = 2 for hss_dexp = 0 (default) and e = hss_dexp - 1
                         otherwise (1: the tolerance of the fronts, as the reference does).  D^-1 inherits cond(D) * tol, the low-rank
                         couplings only tol: measured on Poisson 128^3 the same tolerance for both leaves GMRES unconverged */
  uint8_t mf;         /* 1 or 2: the compressed branch with the reference's data flow (src/factorization.jl:78-112,126-140): the Schur complement of a
                         flagged front leaves as an HSS matrix compressed from the operator Abb - Abi*R (never formed where the children are HSS),
                         a parent with two such children is assembled matrix-free from their generators and the sparse couplings of A: Aib, Abi
                         and Abb are never dense, L / R come from the children's generators (hs_mffront.h).
                         3: D = Aii of such a parent is the reference's `blockfactor` over HSS blocks (src/blockmatrix.jl:121-130): A11 is the left
                            child's own HSS block, A12 / A21 the sparse couplings, S22 = A22 - A21*A11^-1*A12 recompressed from its operator
                            (hss_dexp sets its tolerance); `ldiv!` runs `blockldiv!` (two HSS solves).
                         2: D is ONE HSS matrix over a bisection order of the interior, compressed from the operator [A11 A12; A21 A22].
                         1: D is expanded from the generators and eliminated densely by the front kernels (ni x ni, the one dense block of the
                            front), except on the fronts hss_d selects: the dense LU of a 32,768 block takes 0.5 s, its HSS compression +
                            elimination at 1e-4 1.8 s.
                         With nranks > 1 (dist_top = 0) the joins ship the children's HSS matrices packed into one buffer (hs_exchange_kind,
                         hs_schur_pack / hs_schur_unpack); refused together with dist_top or split.  hs_maxrank includes hssrank(S). */
  uint8_t dist_top;   /* multi-rank factorizations (nranks > 1): 1 = every front ABOVE the rank cut is eliminated by all ranks of its group instead of
                         the group's first rank (the reference factors the two subtrees of a node one after the other although they are independent,
                         src/factorization.jl:20-21; above the cut the independent units are the block columns of one front): block columns of
                         HS_DIST_NB (512) interior DOFs dealt over the group in runs of HS_DIST_PERIOD (2), each factored by its owner and fanned out to the group,
                         every rank updating its own block columns and its slice of the boundary columns; the group ends holding the complete
                         factors and Schur complement, sibling groups swap Schur complements pairwise at the join.  Needs a communicator
                         (hs_set_comm).  Fronts above the cut are eliminated exactly in this mode (no compression there). */
  int64_t seed;       /* RNG seed of the randomized compression (reference: Random.seed!(123), test/rungmres.jl:7) */
} hs_options;

/* Defaults of the reference's kw-constructor (HierarchicalSolvers.jl:43-54): 5,1,1e-6,1e-6,0.5,32,-1,10,false */
void hs_options_default(hs_options* opts);

/* Flat post-ordered form of the two trees `symfact!` returns (nesteddissection.jl:29-69).
 * Node ids are 0-based post-order positions (children before parents, root = nnodes-1), -1 = no child.
 * *_idx hold 1-based values as Julia holds them; *_ptr are 0-based offsets, length nnodes+1.
 *   int_idx  : nd.int      (global DOF ids eliminated at the node)
 *   bnd_idx  : nd.bnd      (global DOF ids of the node's boundary)
 *   iloc_idx : nd_loc.int  (positions in the node's OWN bnd that land in the parent's int)
 *   bloc_idx : nd_loc.bnd  (positions in the node's OWN bnd that land in the parent's bnd)
 */
typedef struct hs_tree {
  int64_t nnodes;
  const int64_t* left;
  const int64_t* right;
  const int64_t* int_ptr;
  const int64_t* int_idx;
  const int64_t* bnd_ptr;
  const int64_t* bnd_idx;
  const int64_t* iloc_ptr;
  const int64_t* iloc_idx;
  const int64_t* bloc_ptr;
  const int64_t* bloc_idx;
} hs_tree;

typedef struct hs_handle hs_handle; /* opaque: owns every device allocation of one factorization */

/* factor(A, nd, nd_loc, opts) for T = Float64 / ComplexF64 (factorization.jl:5).
 * A is n x n CSC with 1-based colptr[n+1], rowval[nnz]; nzval_z is interleaved (re,im) = Julia ComplexF64.
 * Inputs are borrowed for the duration of the call.  On success *out owns the device-resident factors. */
int hs_factor_d(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                const hs_tree* tree, const hs_options* opts, hs_handle** out);
int hs_factor_z(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval_z,
                const hs_tree* tree, const hs_options* opts, hs_handle** out);

/* ldiv!(C, F, B): C = F^{-1} B for n x nrhs column-major host arrays (factornode.jl:62-74).
 * C may alias B (true in-place semantics; the reference's 2-arg form allocates). */
int hs_ldiv_d(hs_handle* F, double* C, int64_t ldc, const double* B, int64_t ldb, int64_t n, int64_t nrhs);
int hs_ldiv_z(hs_handle* F, double* C, int64_t ldc, const double* B, int64_t ldb, int64_t n, int64_t nrhs);

/* Same with DEVICE pointers on HIP stream `stream` (hipStream_t, NULL = default), asynchronous:
 * for an on-device Krylov caller (gmres(...; Pr=F), test/rungmres.jl:47-48). */
int hs_ldiv_dev_d(hs_handle* F, double* dC, int64_t ldc, const double* dB, int64_t ldb, int64_t n, int64_t nrhs, void* stream);
int hs_ldiv_dev_z(hs_handle* F, double* dC, int64_t ldc, const double* dB, int64_t ldb, int64_t n, int64_t nrhs, void* stream);

/* ---- phased form of factor (hs_factor_* = hs_analyze + hs_numeric_* over all levels) -------------------------
 * hs_analyze builds the plan and uploads the sparsity pattern, so a later numeric factorization starts with
 * every input resident in HBM; the pattern (colptr, rowval, tree) is reused for new values of A.
 * rank / nranks (a power of two): with nranks > 1 the 2^p subtrees rooted at tree level p+1 go one per rank
 * (factorization.jl:20-21 factors them one after the other; they are independent), and a front above the cut
 * is eliminated by the first rank of its group (hs_options.dist_top = 0) or by all ranks of its group (dist_top = 1).
 *   dist_top = 0: the library moves no data between ranks: the host layer does, with its own communication library
 *     (torch.distributed / RCCL here, MPI.jl for a Julia host), using hs_exchange_info for WHAT crosses ranks and
 *     hs_set_schur_buffer / hs_pack_bnd / hs_unpack_bnd for WHERE, and drives hs_numeric_levels / hs_solve_*_levels level by level.
 *   dist_top = 1: the library moves everything through the communicator given with hs_set_comm (below): one
 *     hs_numeric_levels(F, nlevels, 0) factors, hs_ldiv_* / hs_ldiv_dev_* solve (every rank passes the same right-hand side and
 *     receives the whole solution). */
int hs_analyze(int is_complex, int64_t n, const int64_t* colptr, const int64_t* rowval, const hs_tree* tree,
               const hs_options* opts, int64_t rank, int64_t nranks, hs_handle** out);
/* Host-side plan only (ownership, exchange list, sizes): touches no device, for schedule tests and sizing. */
int hs_plan(int is_complex, int64_t n, const hs_tree* tree, const hs_options* opts, int64_t rank, int64_t nranks,
            hs_handle** out);
int hs_numeric_begin(hs_handle* F, const void* nzval, int nzval_on_device);
int hs_numeric_levels(hs_handle* F, int64_t level_from, int64_t level_to); /* deepest-first: from >= to; root = 1 */
int hs_numeric_end(hs_handle* F); /* synchronise; HS_ERR_SINGULAR if a front hit an exactly zero pivot */

/* sweeps of ldiv! on a device vector b (n elements of T), restricted to a range of tree levels */
int hs_solve_fwd_levels(hs_handle* F, void* d_b, int64_t level_from, int64_t level_to, void* stream); /* from >= to */
int hs_solve_bwd_levels(hs_handle* F, void* d_b, int64_t level_from, int64_t level_to, void* stream); /* from <= to */

int64_t hs_nlevels(const hs_handle* F);   /* depth(nd) */
int64_t hs_cut_level(const hs_handle* F); /* levels > cut are rank-local; 1 when nranks == 1 */
int64_t hs_node_owner(const hs_handle* F, int64_t node);
int64_t hs_num_exchanges(const hs_handle* F);
/* out6 = {node, level of node, src rank, dst rank, nb, elements of T in the node's Schur buffer (lds*nb)}:
 * node's Schur complement goes src -> dst before dst eliminates node's parent; in ldiv! the vector b[bnd(node)]
 * goes src -> dst in the forward sweep and dst -> src in the backward sweep. */
int hs_exchange_info(const hs_handle* F, int64_t k, int64_t* out6);
/* Order the library's stream against a stream of the host layer without blocking the host: direction 0 = `other_stream` waits for all work
 * the library has enqueued (call before a send that reads a Schur / boundary buffer), 1 = the library's stream waits for all work enqueued on
 * `other_stream` (call after a receive into such a buffer).  NULL = the legacy default stream. */
int hs_stream_order(hs_handle* F, void* other_stream, int direction);
/* 0: exchange k moves a dense Schur complement (hs_set_schur_buffer); 1: it moves an HSS matrix packed into one buffer whose size is known
 * only after the sender compressed it (hs_options.mf with nranks > 1: src calls hs_schur_pack_size + hs_schur_pack after the node's level,
 * ships the byte count and the buffer, dst calls hs_schur_unpack before the parent's level).  This is the reference's data flow over ranks:
 * the parent reads its children's S as HssMatrix objects (src/factorization.jl:78-112,126-140); out6[5] is 0 for such an exchange. */
int64_t hs_exchange_kind(const hs_handle* F, int64_t k);
int hs_schur_pack_size(hs_handle* F, int64_t node, int64_t* bytes);
int hs_schur_pack(hs_handle* F, int64_t node, void* dev_buf, int64_t bytes, void* stream);   /* returns when the buffer is complete */
int hs_schur_unpack(hs_handle* F, int64_t node, const void* dev_buf, int64_t bytes, void* stream);
/* What a factorization does with its options on THIS rank: out8 = {hs_options.mf in effect (0 = the dense-S flow), matrix-free fronts, fronts whose S
 * leaves as an HSS matrix, fronts with low-rank L / R, fronts with an HSS D (hss_d), group fronts (dist_top), ranks, fronts eliminated in slices}. */
int hs_flow_info(const hs_handle* F, int64_t* out8);
/* make `dptr` (device memory owned by the caller, hs_exchange_info's element count) the node's Schur buffer */
int hs_set_schur_buffer(hs_handle* F, int64_t node, void* dptr);
int hs_pack_bnd(const hs_handle* F, int64_t node, const void* d_b, void* d_buf, void* stream);   /* buf[j] = b[bnd_j] */
int hs_unpack_bnd(const hs_handle* F, int64_t node, void* d_b, const void* d_buf, void* stream); /* b[bnd_j] = buf[j] */
int hs_extract_owned(const hs_handle* F, const void* d_b, void* d_out, void* stream); /* out[int(mine)] = b[int(mine)] */

/* ---- communicator: the one data-movement primitive of a multi-rank factorization (hs_options.dist_top) -------------------------
 * Every exchange is a set of point-to-point pieces ("send these device ranges to those ranks, receive those from these"), ordered on a
 * HIP stream.  Two transports:
 *   hs_comm_create_rccl : RCCL over xGMI (grouped ncclSend / ncclRecv on one world communicator the library creates; librccl is opened
 *                         with dlopen at this call).  Rank 0 obtains an id with hs_comm_unique_id and the host distributes its 128 bytes
 *                         with whatever it has (MPI.bcast in a Julia host, torch.distributed here), then every rank calls create.
 *   hs_comm_create_host : the same operation staged through host memory and moved by a callback of the host layer (MPI.jl; gloo in
 *                         the single-GPU rehearsals of this repository, where RCCL refuses several ranks on one device).
 * The callback receives one message per peer and direction (HOST pointers), must complete all of them and return 0. */
typedef struct hs_comm hs_comm;
typedef int (*hs_transfer_fn)(void* user, int64_t nsend, const int64_t* send_peer, void* const* send_buf, const int64_t* send_bytes,
                              int64_t nrecv, const int64_t* recv_peer, void* const* recv_buf, const int64_t* recv_bytes);
int hs_comm_unique_id(void* id128);
int hs_comm_create_rccl(const void* id128, int64_t rank, int64_t nranks, hs_comm** out);
int hs_comm_create_host(hs_transfer_fn fn, void* user, int64_t rank, int64_t nranks, hs_comm** out);
void hs_comm_free(hs_comm* c);
const char* hs_comm_kind(const hs_comm* c); /* "rccl" or "host" */
int hs_comm_selftest(hs_comm* c, int64_t bytes); /* ring shift of a byte pattern (to itself when nranks == 1), checked on the host */
int hs_comm_bandwidth(hs_comm* c, int64_t bytes, int64_t reps, double* gbps); /* ring shifts timed on the transfer stream: GB/s sent per rank (0 when nranks == 1) */
/* attach a communicator (borrowed: it must outlive the handle's factorizations); rank / nranks must equal the handle's */
int hs_set_comm(hs_handle* F, hs_comm* c);

/* gmres(A, b; Pr=F, reltol, abstol, restart, maxiter, log=true) -- the call of the reference's scenario (test/rungmres.jl:47-48; IterativeSolvers.jl
 * 0.9.0, not part of the reference tree): restarted GMRES, RIGHT-preconditioned by the factorization `Pr` (NULL: none), every vector resident on
 * the device, the preconditioner applied through hs_ldiv_dev_*.  A is n x n CSC with 1-based colptr / rowval as Julia holds them (host arrays);
 * b, x: host (where = 0) or device (where = 1) vectors; use_x0 != 0: x holds the initial guess, else zero.  Defaults as in the package for
 * restart <= 0 (min(20, n)), maxiter < 0 (n), reltol < 0 (sqrt(eps)).  Convergence: ||b - A x|| <= max(reltol * ||r0||, abstol).
 * resnorm (may be NULL) receives iters + 1 residual norms (the `log=true` history, resnorm[0] = ||r0||); it must hold maxiter + 1 doubles. */
int hs_gmres_d(hs_handle* Pr, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, double* x, int where, int use_x0,
               double reltol, double abstol, int64_t restart, int64_t maxiter, double* resnorm, int64_t* iters, int* converged, void* stream);
int hs_gmres_z(hs_handle* Pr, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const double* b, double* x, int where, int use_x0,
               double reltol, double abstol, int64_t restart, int64_t maxiter, double* resnorm, int64_t* iters, int* converged, void* stream);

int64_t hs_maxrank(const hs_handle* F); /* factornode.jl:49-57: largest of rank(L), rank(R) and the HSS ranks the factorization
                                           holds (hssrank of the interior blocks kept as HSS, hs_options.hss_d); 0 for the dense path */
/* ranks of one front's Gauss transforms (0 = dense); returns 1 if the front is compressed, 0 if not, <0 on error */
int hs_node_ranks(const hs_handle* F, int64_t node, int64_t* rank_L, int64_t* rank_R);
int hs_is_complex(const hs_handle* F);  /* eltype(F) == ComplexF64 */
int64_t hs_size(const hs_handle* F);    /* n */
void hs_free(hs_handle* F);
/* hs_free parks the three big device blocks of a factorization (factor arena, inverse blocks, Schur scratch; blocks >= 256 MiB, at most four)
 * in a process-wide cache, and the next hs_analyze / hs_factor_* of about the same size takes them over: the driver hands out 139 GiB in
 * 1.4 s in a fresh process but needs 4.9 s once memory of that size has been freed before (measured, MI355X / ROCm 7.2).  hs_trim gives the
 * parked blocks and the recycled blocks of the HSS / low-rank modules back to the driver and returns their bytes (call it with nothing in
 * flight; a failing allocation inside the library does it by itself).  HS_ARENA_CACHE=0 in the environment turns the parking off. */
int64_t hs_trim(void);
const char* hs_last_error(void);
int64_t hs_last_error_info(void); /* e.g. node id of a singular front */

/* ---- introspection (metrics the reference lacks) ---- */
typedef struct hs_stats {
  int64_t n, nnodes, nlevels;
  int64_t max_ni, max_nb;
  double flops_factor;   /* sum of F(ni,nb) = 2/3 ni^3 + 2 ni^2 nb + 2 ni nb^2 (x4 for complex), SURVEY.md 8(d) */
  double bytes_factors;  /* device bytes held by the factors */
  double bytes_solve;    /* algorithmic bytes one ldiv! with nrhs=1 must read */
  double t_symbolic, t_upload, t_assemble, t_panel, t_trsm, t_gemm, t_total; /* seconds, device time of last hs_factor */
  double t_solve;        /* seconds, device time of last hs_ldiv */
  double gemm_flops;     /* flops executed by the MFMA GEMM kernel in the last hs_factor */
  int64_t gemm_launches;
  /* every launch of the kernel `gemm_op_kernel` in the last hs_factor (the trailing/Schur updates; the 32-row TRSM base
   * cases run the same tile code as `trsm_inv_kernel`): what a rocprofv3 --stats line of that kernel is compared with */
  double t_mfma_kernel;
  int64_t mfma_kernel_launches;
  double gemm_bytes;     /* algorithmic bytes of those launches: A and B read once, C read and written once ((M*K + K*N + 2*M*N) * sizeof(T)
                            summed over the fronts of every launch) -- what measured HBM traffic is compared with */
} hs_stats;
int hs_get_stats(const hs_handle* F, hs_stats* out);

/* per-node sizes (post-order id): ni, nb, level (root = 1) */
int hs_node_info(const hs_handle* F, int64_t node, int64_t* ni, int64_t* nb, int64_t* level);

/* Export one node's stored blocks to host (column-major, tight leading dimension), for parity tests:
 *   HS_BLK_LU  : ni x ni   packed L\U of the pivoted interior block  (P*Aii = L*U)
 *   HS_BLK_LBI : nb x ni   Abi * U^{-1}
 *   HS_BLK_UIB : ni x nb   L^{-1} * P * Aib
 *   HS_BLK_S   : nb x nb   Schur complement in the node's own bnd order (needs opts.keep_schur)
 * out must hold rows*cols elements of T (2 doubles per element for complex).
 *   hs_node_export_piv: ni int64 values, 0-based row permutation p with (P*x)[i] = x[p[i]].
 * From these: D = P'LU, L = Lbi*L^{-1}*P, R = U^{-1}*Uib  (FactorNode fields, factornode.jl:7-12). */
enum { HS_BLK_LU = 0, HS_BLK_LBI = 1, HS_BLK_UIB = 2, HS_BLK_S = 3 };
int hs_node_export(const hs_handle* F, int64_t node, int which, double* out);
int hs_node_export_piv(const hs_handle* F, int64_t node, int64_t* out);

/* Library/device identification: fills name (e.g. "gfx950") and returns the CU count, or <0 if no usable device. */
int hs_device_info(char* arch_name, int64_t len, int64_t* cu_count, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* HS_SOLVER_H */
