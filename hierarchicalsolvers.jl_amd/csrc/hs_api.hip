// hs_api.hip -- host side of the C ABI (include/hs_solver.h): tree validation, HBM layout, the
// level-batched recursive-LU schedule and the triangular sweeps.  No arithmetic happens on the
// host; if there is no usable HIP device every entry point fails with HS_ERR_DEVICE.
//
// Reference control flow being replaced: factor/_factor recursion (src/factorization.jl:5-27),
// _factor_leaf / _factor_branch dense variants (:30-42, :62-75), ldiv! (src/factornode.jl:62-99).
// The reference walks the tree sequentially (left subtree, then right, :20-21); here every node of
// one tree level is processed by the same grouped kernel launches (blockIdx.y = node), and with
// nranks > 1 the subtrees below the cut level belong to different ranks (one process per GPU).
//
// Phases (hs_factor_* = analyze + numeric over all levels):
//   hs_analyze        plan, HBM allocation, upload of the sparsity pattern / index lists / descriptors
//   hs_numeric_begin  values of A (host or device pointer) -> device
//   hs_numeric_levels assemble + eliminate the owned fronts of a range of tree levels
//   hs_numeric_end    synchronise, singular-front check, timings
//   hs_solve_*_levels forward / backward sweeps over a range of levels on a device vector
#include <chrono>
#include <mutex>
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <utility>
#include <vector>

#include "../../include/hs_solver.h"
#include "../../include/hs_hss.h"
#include "hs_common.h"

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static thread_local long long g_err_info = 0;

void hs_set_error(int code, long long info, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  g_err_info = info;
  (void)code;
}
struct HsError {
  int code;
};
#define HS_FAIL(code, info, ...)           \
  do {                                     \
    hs_set_error(code, info, __VA_ARGS__); \
    throw HsError{code};                   \
  } while (0)

#define HS_GUARD(...)                                        \
  try {                                                      \
    __VA_ARGS__;                                             \
    return HS_OK;                                            \
  } catch (const HsError& e) {                               \
    return e.code;                                           \
  } catch (int code) {                                       \
    return code;                                             \
  } catch (const std::bad_alloc&) {                          \
    hs_set_error(HS_ERR_NOMEM, 0, "host allocation failed"); \
    return HS_ERR_NOMEM;                                     \
  }

extern "C" const char* hs_last_error(void) { return g_err.c_str(); }
extern "C" int64_t hs_last_error_info(void) { return g_err_info; }

extern "C" void hs_options_default(hs_options* o) {
  memset(o, 0, sizeof *o);
  o->swlevel = 5;
  o->swsize = 1;
  o->atol = 1e-6;
  o->rtol = 1e-6;
  o->c_tol = 0.5;
  o->leafsize = 32;
  o->kest = -1;
  o->stepsize = 10;
  o->verbose = 0;
  o->seed = 123;
}

static void chkopts(const hs_options& o) {  // HierarchicalSolvers.jl:73-79
  if (!(o.swsize >= 1)) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: swsize");
  if (!(o.atol >= 0.0)) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: atol");
  if (!(o.rtol >= 0.0)) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: rtol");
  if (!(o.c_tol > 0.0 && o.c_tol <= 1.0)) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: c_tol");
  if (!(o.leafsize >= 1)) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: leafsize");
  if (o.mf > 3) HS_FAIL(HS_ERR_ARGUMENT, o.mf, "ArgumentError: hs_options.mf must be in 0:3, got %d", (int)o.mf);
}

static void require_device() {
  int cnt = 0;
  hipError_t e = hipGetDeviceCount(&cnt);
  if (e != hipSuccess || cnt <= 0)
    HS_FAIL(HS_ERR_DEVICE, 0, "no HIP device available (%s): this library has no CPU fallback",
            e == hipSuccess ? "device count 0" : hipGetErrorString(e));
}

// ------------------------------------------------------------------------------------------------
// host-side plan
// ------------------------------------------------------------------------------------------------
struct NodeH {
  int left = -1, right = -1, parent = -1, level = 0;
  int ni = 0, nb = 0, m = 0;
  int ni1 = 0, nb1 = 0;
  bool leaf = true;
  int owner = 0;       // rank that eliminates this front
  bool mine = true;    // owner == my rank
  bool ghost = false;  // not mine, but one of my fronts absorbs its Schur complement (received from its owner)
  int glo = 0, gcnt = 1;  // the ranks [glo, glo+gcnt) hold the subtrees below this node
  bool dist = false;      // hs_options.dist_top: a front above the rank cut eliminated by its whole group (hs_dist.h); `mine` = member of the group
  int ldl = 0, ldu = 0, lds = 0;
  size_t off_LF = 0, off_UR = 0, off_SB = 0, off_inv = 0, off_inv256 = 0;        // element offsets
  // A compressed front (low-rank L / R, hs_compress.h) needs its dense front [Aii; Abi | Aib] only WHILE its level is eliminated: afterwards the
  // Gauss transforms live in the low-rank objects and only the LU of Aii is read again (ldiv!).  Such a front is assembled and eliminated in a
  // per-level SCRATCH arena (off_LF / off_UR index hs_handle::d_cfs) and its LU is copied, compactly (leading dimension ni), into the factor
  // arena at off_LFc: 2 ni nb elements per front less to keep (29 GiB of 153 at Poisson 128^3 with levels 2-4 compressed).
  bool cfront = false;
  size_t off_LFc = 0;
  int ldc = 2;
  size_t off_fidx = 0, off_ipiv = 0, off_rperm = 0, off_cmap = 0, off_cand = 0;  // int offsets
  int ncand = 0;
  int batch_pos = -1;      // index inside its level's batch of owned fronts
  void* ext_sb = nullptr;  // caller-provided device buffer for the Schur complement (exchange between ranks)
  long long woff = 0;
  bool compressed = false;  // level <= swlevel && |bnd| >= swsize (factorization.jl:15): low-rank Gauss transforms
  void* lrL = nullptr;      // LowRank<T>* of Lbi = Abi*U^-1  (nb x ni)
  void* lrR = nullptr;      // LowRank<T>* of Uib = L^-1*P*Aib (ni x nb)
  int last_rL = 0, last_rR = 0;  // ranks found by the previous factorization (initial sketch width of the next one)
  int kind = 0;                  // hs_split.h: 0 ordinary, 1 first slice of a split front, 2 later slice
  int user = -1;                 // user (post-order) node id this internal node belongs to
  int oni = 0, oni1 = 0, onb1 = 0;  // split points of the front the children address (= ni, ni1, nb1 unless kind 1)
  long long off_spos = -1;          // kind 1 with re-ordered interior: int offset of the slice-order -> original-position table
  // hs_hssfront.h: D = Aii kept as an HSS matrix (hs_options.hss_d)
  bool hssd = false;
  void* hss = nullptr;        // hs_hss*
  void* hW = nullptr;         // Aii^-1 * C_R (ni x rank(R), leading dimension hldw)
  int hldw = 0;
  void* ht = nullptr;         // ni entries: D^-1 rhs[int] between the two sweeps of ldiv!
  int last_k = 0;             // samples the previous compression of D ended with
  std::vector<int64_t> ilv;   // interleaved order of the interior positions (empty: as assembled)
  // hs_mffront.h: matrix-free compressed branch (hs_options.mf)
  bool s_hss = false;          // the Schur complement of this front leaves as an HSS matrix
  bool mf = false;             // both children hand over HSS Schur complements: the front is never assembled densely
  void* S_hss = nullptr;       // hs_hss*: S[perm, perm], perm = [int_loc; bnd_loc], first split at n1p
  int n1p = 0;                 // |int_loc|: boundary DOFs of this node that become interior at the parent
  bool mfd = false;            // matrix-free front whose interior block D = Aii is expanded and eliminated DENSELY (hs_options.mf == 1 and not an
                               // hss_d front): S, Aib, Abi still travel as generators; buffers sized by the generators' ranks, per factorization
  void *mfd_LF = nullptr, *mfd_UR = nullptr, *mfd_SB = nullptr;
  int mfd_nbp = 0, mfd_ldl = 0, mfd_ldu = 0;
  int last_ks = 0;             // samples the previous compression of S ended with
  std::vector<int64_t> sperm;  // [int_loc; bnd_loc] as 0-based positions in this node's boundary
  struct Coupling {            // entries of a sparse coupling block in block coordinates; e = position in nzval
    std::vector<int> row, col;
    std::vector<long long> e;
    size_t size() const { return row.size(); }
  } xr, xl,                    // A[int, bnd] / A[bnd, int] between the two children's parts (matrix-free fronts)
      x12, x21;                // A[int1, int2] / A[int2, int1] in block coordinates (the 2x2 block form of D, hs_options.mf == 3)
  // hs_options.mf == 3: D = blockfactor([A11 A12; A21 A22]) over HSS blocks (src/blockmatrix.jl:121-130) -- hss = A11 (a view of the left
  // child's Schur complement, kept alive in hss_keep), hss2 = S22 = A22 - A21*A11^-1*A12 recompressed, A12 = C12*Z12 and A21 = C21*Z21 exactly
  bool mfb = false;
  void *hss2 = nullptr, *hss_keep = nullptr;
  void *bW12 = nullptr, *bZ12 = nullptr, *bC21 = nullptr, *bZ21 = nullptr;  // A11^-1*C12 (n1 x k12), Z12 (k12 x n2), C21 (n2 x k21), Z21 (k21 x n1)
  int bk12 = 0, bk21 = 0, bldw = 0, bldz12 = 0, bldc21 = 0, bldz21 = 0;
};

struct LevelH {
  std::vector<int> nodes;  // every node of the level
  std::vector<int> mine;   // nodes this rank eliminates
  int maxni = 0, maxnb = 0, maxm = 0, maxnbc = 0;
  size_t lf_begin = 0, lf_end = 0;  // factor-arena range (elements) of this level's LF/UR
  size_t cfs_end = 0;               // scratch-arena extent (elements) of this level's compressed fronts (NodeH::cfront)
  size_t sb_begin = 0, sb_end = 0;  // SB range (elements)
  size_t desc_off = 0;              // first NodeDesc / SolveNode of this level (owned fronts only)
  size_t sc_off = 0, sc_cnt = 0;    // ScatterDesc range
  std::vector<int> h_ni, h_nb;
  int ndense = 0;                          // the first ndense entries of `mine` are dense fronts (one batch), the rest compressed
  int nplain = 0;                          // of those, the first nplain keep a dense LU of D, the others an HSS form (hs_hssfront.h)
  int nmf = 0;                             // the last nmf entries of `mine` are matrix-free fronts (hs_mffront.h): never assembled
  int dmaxni = 0, dmaxnb = 0, dmaxm = 0;   // extents of the dense batch
};

struct Exchange {
  int node, level, src, dst, nb;
  long long nelems;
  int hss = 0;  // 1: the node's Schur complement crosses as a packed HSS matrix (hs_schur_pack / hs_schur_unpack), not as a dense block
};

#include "hs_sched.h"
#include "hs_split.h"
#include "hs_comm.h"

struct hs_handle {
  size_t fac_bytes = 0, inv_bytes = 0, sb_bytes = 0;  // sizes d_fac / d_inv / d_sb were asked for (the arena cache parks them by size)
  void* d_cfs = nullptr;                              // scratch arena of the compressed fronts (NodeH::cfront), one level at a time
  size_t cfs_bytes = 0, cfs_elems = 0;
  bool is_complex = false;
  int64_t n = 0, nnz = 0;
  int nnodes = 0;  // tree nodes (+1 pseudo-node when the root keeps a boundary)
  int nreal = 0;   // internal nodes without the pseudo-root (= user nodes unless fronts are split, hs_split.h)
  int nuser = 0;   // nodes of the user's tree
  std::vector<int> last_of_user, first_of_user, u_ni, u_nb, u_level;  // only when fronts are split
  int rank = 0, nranks = 1, cut_level = 1;  // levels > cut_level are rank-local
  hs_options opts;
  std::vector<NodeH> nodes;
  std::vector<LevelH> levels;  // index = level (0 = pseudo-root)
  std::vector<int> fidx_host;
  std::vector<Exchange> exchanges;
  // device
  void* d_fac = nullptr;  // LF / UR
  void* d_inv = nullptr;  // invL / invU
  void* d_sb = nullptr;   // SB scratch (or permanent with keep_schur)
  int* d_int = nullptr;   // fidx, ipiv, rperm, cmap
  int* d_tmpi = nullptr;  // cand, pivlist, info[nnodes], own[n], pos[n]
  int* d_info = nullptr;
  int* d_growth = nullptr;
  bool optimistic = true;   // HS_OPTIMISTIC=0 turns it off; a level that had to be redone turns it off for the rest of the handle's life
  int* d_own = nullptr;
  int* d_pos = nullptr;
  int* d_owned = nullptr;   // 0-based ids of the DOFs this rank eliminates (hs_extract_owned: one launch)
  int64_t n_owned = 0;
  // hs_options.mf: CSR form of A next to the CSC one (both products of the block operators are gather-form), operator index map
  int64_t* d_rowptr = nullptr;
  int32_t* d_colind = nullptr;
  int64_t* d_tperm = nullptr;  // CSR entry -> CSC entry
  void* d_nzr = nullptr;
  int* d_lpos = nullptr;
  bool mf_on = false;
  double static_factor_bytes = 0.0;  // LF / UR / inverse blocks of the arena (the HSS and low-rank objects are counted per factorization)
  int64_t* d_colptr = nullptr;
  int32_t* d_rowval = nullptr;
  void* d_nz = nullptr;
  void* d_nodes = nullptr;  // NodeDesc<T>[] of the owned fronts, level by level
  void* d_sc = nullptr;     // ScatterDesc<T>[]
  void* d_solve = nullptr;  // SolveNode<T>[] (same order as d_nodes)
  void* d_w1 = nullptr;
  void* d_w2 = nullptr;
  int* d_flow = nullptr;    // dataflow sweeps (kernels_solve_wide.hip): 64 workgroup-id counters, one per launch, used as a ring
  void* d_e1 = nullptr;     // ... the exchange vectors (laid out like d_w1): y / x, and the finished w of the diagonal blocks
  void* d_e2 = nullptr;
  size_t flow_bytes = 0;
  int* h_flow_err = nullptr;  // pinned, device-visible: raised by a sweep workgroup whose bounded wait ran out
  int flow_seq = 0;
  void* d_part = nullptr;
  void* d_b = nullptr;
  size_t fac_elems = 0, inv_elems = 0, sb_elems = 0, int_elems = 0;
  bool sb_kept = false;
  bool numeric_open = false, factored = false;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;    // side stream for look-ahead panels (own compute units, or high priority)
  hipStream_t stream_la = nullptr;  // CU-masked pair for the look-ahead schedule of a lone front: stream_la = every CU
  hipStream_t stream2m = nullptr;   // but a reserved few, stream2m = the reserved ones
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hs_comm* comm = nullptr;           // borrowed (hs_set_comm): moves everything that crosses ranks when hs_options.dist_top is set
  hipStream_t stream_comm = nullptr; // every transfer is enqueued here
  int* d_gflags = nullptr;           // nranks ints: flags agreed inside a group (hs_dist.h)
  void* d_stage_u = nullptr;
  void *d_stage_s = nullptr, *d_stage_r = nullptr;  // packing buffers of the block-column messages (hs_dist.h)
  // ldiv! of a dist_top factorization inside the library: boundary values swapped at the joins, owned solution pieces gathered at the end
  void *d_xs = nullptr, *d_xr = nullptr, *d_xall = nullptr;
  int* d_owned_all = nullptr;            // the DOFs every rank owns, rank after rank
  std::vector<int64_t> owned_off;        // nranks + 1 offsets into it
  hipEvent_t ev_ca = nullptr, ev_cb = nullptr;
  Profiler prof;
  void* d_cdesc = nullptr;    // private descriptors (3 per front) of the compressed fronts being eliminated
  size_t cdesc_cap = 0;
  void* d_lr_t = nullptr;     // low-rank apply workspace
  void* d_lr_part = nullptr;
  size_t lr_t_elems = 0, lr_part_elems = 0;
  int64_t maxrank = 0;
  std::vector<std::pair<int, hipEvent_t>> level_events;
  double flops = 0.0;
  hs_stats stats;
};

static inline int rup(int x, int a) { return (x + a - 1) / a * a; }
static inline size_t rups(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------
// The three big blocks of a factorization -- factor arena (139 GiB at Poisson 128^3), inverse blocks, Schur scratch -- come back from the
// driver slowly once the process has freed memory of that size before: measured hipMalloc of the 139 GiB arena 1.4 s in a fresh process,
// 4.9 s after a hipFree of the previous one (tools/oneshot_probe.py; that was the unexplained 5 s between `factor_oneshot_s` and
// analyze + numeric in round 2).  hs_free therefore parks them in a small process-wide cache and the next hs_analyze / hs_factor_* of a
// similar size takes them over; hs_trim() (or HS_ARENA_CACHE=0) gives them back.  At most HS_ARENA_SLOTS blocks are parked; an allocation that
// fails anywhere in the library empties the cache and tries again.
// ------------------------------------------------------------------------------------------------
namespace {
struct ArenaCache {
  std::mutex mu;
  struct Blk { void* p; size_t bytes; };
  std::vector<Blk> v;
};
ArenaCache& arena_cache() {
  static ArenaCache* c = new ArenaCache();
  return *c;
}
constexpr size_t HS_ARENA_MIN = (size_t)256 << 20;  // smaller blocks are not worth parking
constexpr size_t HS_ARENA_SLOTS = 4;
bool arena_cache_on() {
  static const bool on = !(getenv("HS_ARENA_CACHE") && getenv("HS_ARENA_CACHE")[0] == '0');
  return on;
}
}  // namespace
extern "C" int64_t hs_arena_trim(void) {  // the parked arenas only (what the block pools of the HSS / low-rank modules call when THEY run out)
  int64_t freed = 0;
  ArenaCache& c = arena_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  for (auto& b : c.v) {
    (void)hipFree(b.p);
    freed += (int64_t)b.bytes;
  }
  c.v.clear();
  return freed;
}
extern "C" int64_t hs_trim(void) { return hs_arena_trim() + hs_hss_trim(); }  // device bytes given back to the driver
static void* arena_take(size_t bytes) {
  if (!arena_cache_on() || bytes < HS_ARENA_MIN) return nullptr;
  ArenaCache& c = arena_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  size_t best = c.v.size();
  for (size_t i = 0; i < c.v.size(); ++i)
    if (c.v[i].bytes >= bytes && c.v[i].bytes <= bytes + bytes / 8 && (best == c.v.size() || c.v[i].bytes < c.v[best].bytes)) best = i;
  if (best == c.v.size()) return nullptr;
  void* p = c.v[best].p;
  c.v.erase(c.v.begin() + (long)best);
  return p;
}
static void arena_give(void* p, size_t bytes) {
  if (!p) return;
  if (!arena_cache_on() || bytes < HS_ARENA_MIN) {
    (void)hipFree(p);
    return;
  }
  ArenaCache& c = arena_cache();
  std::lock_guard<std::mutex> lk(c.mu);
  c.v.push_back({p, bytes});
  while (c.v.size() > HS_ARENA_SLOTS) {  // the oldest goes back to the driver
    (void)hipFree(c.v.front().p);
    c.v.erase(c.v.begin());
  }
}

static void free_lowrank_any(hs_handle* h);
static void free_hss_any(hs_handle* h);
static void free_mfd_buffers(hs_handle* h);
static void free_handle(hs_handle* h) {
  if (!h) return;
  if (h->stream) (void)hipStreamSynchronize(h->stream);  // recycled blocks must be idle when they go back to the caches
  for (auto& x : h->nodes)
    if (x.S_hss) {
      hs_hss_free((hs_hss*)x.S_hss);
      x.S_hss = nullptr;
    }
  free_mfd_buffers(h);
  free_hss_any(h);
  free_lowrank_any(h);
  arena_give(h->d_fac, h->fac_bytes);
  arena_give(h->d_inv, h->inv_bytes);
  arena_give(h->d_sb, h->sb_bytes);
  arena_give(h->d_cfs, h->cfs_bytes);
  void* ptrs[] = {h->d_int, h->d_tmpi, h->d_colptr, h->d_rowval, h->d_nz,
                  h->d_nodes, h->d_sc,  h->d_solve, h->d_w1,  h->d_w2,   h->d_part,   h->d_b,   h->d_owned,
                  h->d_rowptr, h->d_colind, h->d_tperm, h->d_nzr, h->d_lpos};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->d_gflags) (void)hipFree(h->d_gflags);
  if (h->d_flow) (void)hipFree(h->d_flow);
  if (h->d_e1) (void)hipFree(h->d_e1);
  if (h->d_e2) (void)hipFree(h->d_e2);
  if (h->h_flow_err) (void)hipHostFree(h->h_flow_err);
  for (void* q : {h->d_xs, h->d_xr, h->d_xall, (void*)h->d_owned_all})
    if (q) (void)hipFree(q);
  if (h->ev_ca) (void)hipEventDestroy(h->ev_ca);
  if (h->ev_cb) (void)hipEventDestroy(h->ev_cb);
  if (h->d_stage_u) (void)hipFree(h->d_stage_u);
  if (h->d_stage_s) (void)hipFree(h->d_stage_s);
  if (h->d_stage_r) (void)hipFree(h->d_stage_r);
  if (h->stream_comm) (void)hipStreamDestroy(h->stream_comm);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream_la) (void)hipStreamDestroy(h->stream_la);
  if (h->stream2m) (void)hipStreamDestroy(h->stream2m);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

extern "C" void hs_free(hs_handle* h) { free_handle(h); }

// Build and validate the node table from the flat 1-based tree (symfact! output).
static void build_plan(hs_handle* h, int64_t n, const hs_tree* tr, const SplitTree* st = nullptr) {
  if (!tr || tr->nnodes <= 0) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: empty tree");
  const int nn = (int)tr->nnodes;
  h->nreal = nn;
  std::vector<NodeH>& N = h->nodes;
  N.assign(nn, NodeH());
  auto len = [](const int64_t* ptr, int i) { return (int)(ptr[i + 1] - ptr[i]); };
  for (int i = 0; i < nn; ++i) {
    NodeH& x = N[i];
    x.left = (int)tr->left[i];
    x.right = (int)tr->right[i];
    x.kind = st ? st->kind[i] : 0;
    x.user = st ? st->user[i] : i;
    if ((x.left < 0) != (x.right < 0) && x.kind != 2)  // a later slice of a split front has one child: the previous slice
      HS_FAIL(HS_ERR_TREE, i, "Expected nested dissection to be a binary tree. Found a node with only one child.");
    x.leaf = x.left < 0;
    if (!x.leaf) {
      if (x.left >= i || x.right >= i || x.left == x.right)
        HS_FAIL(HS_ERR_ARGUMENT, i, "ArgumentError: tree is not in post-order (children must precede node %d)", i);
      if (N[x.left].parent >= 0 || (x.right >= 0 && N[x.right].parent >= 0)) HS_FAIL(HS_ERR_ARGUMENT, i, "ArgumentError: node has two parents");
      N[x.left].parent = i;
      if (x.right >= 0) N[x.right].parent = i;
    }
    x.ni = len(tr->int_ptr, i);
    x.nb = len(tr->bnd_ptr, i);
    x.m = x.ni + x.nb;
  }
  for (int i = 0; i < nn - 1; ++i)
    if (N[i].parent < 0) HS_FAIL(HS_ERR_ARGUMENT, i, "ArgumentError: found either less than or more than one root.");
  N[nn - 1].level = 1;  // levels: root = 1
  int maxlevel = 1;
  for (int i = nn - 2; i >= 0; --i) {
    N[i].level = N[N[i].parent].level + 1;
    maxlevel = std::max(maxlevel, N[i].level);
  }
  // index sets + consistency with the local maps
  h->fidx_host.clear();
  std::vector<int>& F = h->fidx_host;
  for (int i = 0; i < nn; ++i) {
    NodeH& x = N[i];
    x.off_fidx = F.size();
    for (int64_t e = tr->int_ptr[i]; e < tr->int_ptr[i + 1]; ++e) {
      int64_t g = tr->int_idx[e];
      if (g < 1 || g > n) HS_FAIL(HS_ERR_DIMENSION, i, "BoundsError: int index %lld of node %d outside 1:%lld", (long long)g, i, (long long)n);
      F.push_back((int)(g - 1));
    }
    for (int64_t e = tr->bnd_ptr[i]; e < tr->bnd_ptr[i + 1]; ++e) {
      int64_t g = tr->bnd_idx[e];
      if (g < 1 || g > n) HS_FAIL(HS_ERR_DIMENSION, i, "BoundsError: bnd index %lld of node %d outside 1:%lld", (long long)g, i, (long long)n);
      F.push_back((int)(g - 1));
    }
  }
  for (int i = 0; i < nn; ++i) {
    NodeH& x = N[i];
    if (x.leaf) {
      x.ni1 = x.ni;
      x.nb1 = x.nb;
      x.oni = x.ni; x.oni1 = x.ni1; x.onb1 = x.nb1;
      continue;
    }
    if (x.kind == 1) {
      // first slice of a split front (hs_split.h): the children address the ORIGINAL front [int; bnd] of the user's node,
      // of which this node's [int; bnd] is the same sequence with the int/bnd border moved; the user's node was validated
      // by the caller of make_split_tree through its last slice
      x.oni = st->oni[i]; x.oni1 = st->oni1[i]; x.onb1 = st->onb1[i];
      x.ni1 = x.ni;
      x.nb1 = x.nb;
      continue;
    }
    // int = [left.bnd[iloc_left]; right.bnd[iloc_right]], bnd likewise (nesteddissection.jl:64-65, factorization.jl:63-64)
    int pi = 0, pb = 0;
    for (int side = 0; side < 2; ++side) {
      int c = side == 0 ? x.left : x.right;
      if (c < 0) continue;
      const NodeH& ch = N[c];
      const int* cb = F.data() + ch.off_fidx + ch.ni;
      for (int64_t e = tr->iloc_ptr[c]; e < tr->iloc_ptr[c + 1]; ++e, ++pi) {
        int64_t q = tr->iloc_idx[e];
        if (q < 1 || q > ch.nb) HS_FAIL(HS_ERR_DIMENSION, c, "BoundsError: nd_loc.int position %lld outside child bnd 1:%d", (long long)q, ch.nb);
        if (pi >= x.ni || F[x.off_fidx + pi] != cb[q - 1])
          HS_FAIL(HS_ERR_DIMENSION, i, "DimensionMismatch: int of node %d is not [left.bnd[loc.int]; right.bnd[loc.int]]", i);
      }
      for (int64_t e = tr->bloc_ptr[c]; e < tr->bloc_ptr[c + 1]; ++e, ++pb) {
        int64_t q = tr->bloc_idx[e];
        if (q < 1 || q > ch.nb) HS_FAIL(HS_ERR_DIMENSION, c, "BoundsError: nd_loc.bnd position %lld outside child bnd 1:%d", (long long)q, ch.nb);
        if (pb >= x.nb || F[x.off_fidx + x.ni + pb] != cb[q - 1])
          HS_FAIL(HS_ERR_DIMENSION, i, "DimensionMismatch: bnd of node %d is not [left.bnd[loc.bnd]; right.bnd[loc.bnd]]", i);
      }
      if (side == 0) {
        x.ni1 = pi;
        x.nb1 = pb;
      }
    }
    if (pi != x.ni || pb != x.nb)
      HS_FAIL(HS_ERR_DIMENSION, i, "DimensionMismatch: children contribute (%d,%d) DOFs, node %d has (%d,%d)", pi, pb, i, x.ni, x.nb);
    x.oni = x.ni; x.oni1 = x.ni1; x.onb1 = x.nb1;
  }
  // every DOF is eliminated at most once; the reference concatenates child Schur complements, it
  // never extend-adds (factorization.jl:118-121), so fronts of unrelated nodes must be disjoint: the fronts of one
  // level are assembled by the same grouped launches (mark / gather write own[g], pos[g] per DOF of [int; bnd]), so a
  // DOF shared by two fronts of a level (a vertex-separator tree) or listed twice in one front is rejected here
  {
    std::vector<char> seen(n, 0);
    for (int i = 0; i < nn; ++i)
      for (int e = 0; e < N[i].ni; ++e) {
        int g = F[N[i].off_fidx + e];
        if (seen[g]) HS_FAIL(HS_ERR_DIMENSION, i, "DimensionMismatch: DOF %d is interior to two nodes", g + 1);
        seen[g] = 1;
      }
    std::vector<int> lastlv(n, -1), lastnode(n, -1);
    for (int i = 0; i < nn; ++i)
      for (int e = 0; e < N[i].m; ++e) {
        int g = F[N[i].off_fidx + e];
        if (lastnode[g] == i) HS_FAIL(HS_ERR_DIMENSION, i, "DimensionMismatch: DOF %d is listed twice in the front [int; bnd] of node %d", g + 1, i);
        if (lastlv[g] == N[i].level)
          HS_FAIL(HS_ERR_DIMENSION, i, "DimensionMismatch: DOF %d belongs to the fronts of nodes %d and %d of the same tree level (fronts of a level must be disjoint)",
                  g + 1, lastnode[g], i);
        lastlv[g] = N[i].level;
        lastnode[g] = i;
      }
  }
  // pseudo-node: the root's own boundary (normally empty) is eliminated last, `F.S \ C[F.bnd,:]` (factornode.jl:72)
  const NodeH root = N[nn - 1];
  if (root.nb > 0) {
    NodeH r;
    r.leaf = false;
    r.left = nn - 1;
    r.right = -1;
    r.ni = root.nb;
    r.nb = 0;
    r.m = r.ni;
    r.ni1 = r.ni;
    r.nb1 = 0;
    r.oni = r.ni; r.oni1 = r.ni1; r.onb1 = 0;
    r.level = 0;
    r.off_fidx = F.size();
    for (int e = 0; e < root.nb; ++e) F.push_back(F[root.off_fidx + root.ni + e]);
    N.push_back(r);
    N[nn - 1].parent = nn;
  }
  h->nnodes = (int)N.size();
  h->levels.assign(maxlevel + 1, LevelH());

  // ---- ownership: the 2^p subtrees rooted at level p+1 go one per rank; a node above the cut is
  // eliminated by the first rank of the group that owns the subtrees below it.
  int p = 0;
  while ((1 << (p + 1)) <= h->nranks) ++p;
  if ((1 << p) != h->nranks) HS_FAIL(HS_ERR_ARGUMENT, h->nranks, "ArgumentError: nranks = %d must be a power of two", h->nranks);
  h->cut_level = p + 1;
  {
    std::vector<int> lo(h->nnodes, 0), cnt(h->nnodes, h->nranks);  // rank range [lo, lo+cnt) below each node
    for (int i = h->nnodes - 1; i >= 0; --i) {
      NodeH& x = N[i];
      if (x.level == 0) {  // pseudo-root: same owner as the root
        lo[x.left] = 0;
        cnt[x.left] = h->nranks;
      }
      x.owner = lo[i];
      x.glo = lo[i];
      x.gcnt = x.level == 0 ? 1 : cnt[i];
      x.dist = h->opts.dist_top && h->nranks > 1 && x.level >= 1 && cnt[i] > 1;
      x.mine = x.dist ? (h->rank >= x.glo && h->rank < x.glo + x.gcnt) : (x.owner == h->rank);
      if (x.level == 0 && h->opts.dist_top && h->nranks > 1)
        HS_FAIL(HS_ERR_UNSUPPORTED, i, "hs_options.dist_top with a root that keeps a boundary (|root.bnd| = %d)", x.ni);
      if (x.dist && x.leaf)  // a branch of the tree ends above the rank cut: its group would have nothing to join
        HS_FAIL(HS_ERR_UNSUPPORTED, i, "hs_options.dist_top: node %d is a leaf at level %d, above the rank cut of %d ranks (level %d); use fewer ranks or dist_top = 0", i,
                x.level, h->nranks, h->cut_level);
      if (!x.leaf && x.level >= 1) {
        if (x.right < 0) {  // later slice of a split front (single-rank plans only): same ranks as its only child
          lo[x.left] = lo[i];
          cnt[x.left] = cnt[i];
        } else if (cnt[i] > 1) {
          lo[x.left] = lo[i];
          cnt[x.left] = cnt[i] / 2;
          lo[x.right] = lo[i] + cnt[i] / 2;
          cnt[x.right] = cnt[i] / 2;
        } else {
          lo[x.left] = lo[x.right] = lo[i];
          cnt[x.left] = cnt[x.right] = 1;
        }
      }
    }
  }
  h->exchanges.clear();
  for (int i = 0; i < h->nnodes; ++i) {
    NodeH& x = N[i];
    if (x.parent >= 0 && N[x.parent].dist) {
      // the parent is eliminated by the union of the two children's groups: every rank of this node's group swaps with its partner in the
      // sibling's group (rank +- gcnt), so that each member of the parent's group holds both Schur complements
      const NodeH& p = N[x.parent];
      const bool is_left = p.left == i;
      for (int r = x.glo; r < x.glo + x.gcnt; ++r) h->exchanges.push_back({i, x.level, r, is_left ? r + x.gcnt : r - x.gcnt, x.nb, 0});
      if (p.mine && !x.mine) x.ghost = true;
    } else if (x.parent >= 0 && N[x.parent].owner != x.owner) {
      h->exchanges.push_back({i, x.level, x.owner, N[x.parent].owner, x.nb, 0});
      if (N[x.parent].mine) x.ghost = true;
    }
  }
  for (int i = 0; i < h->nnodes; ++i) {
    LevelH& L = h->levels[N[i].level];
    L.nodes.push_back(i);
    if (!N[i].mine) continue;
    N[i].batch_pos = (int)L.mine.size();
    L.mine.push_back(i);
    L.h_ni.push_back(N[i].ni);
    L.h_nb.push_back(N[i].nb);
    L.maxni = std::max(L.maxni, N[i].ni);
    L.maxnb = std::max(L.maxnb, N[i].nb);
    L.maxm = std::max(L.maxm, N[i].m);
  }
}

static void dmalloc(void** p, size_t bytes, const char* what);
template <class T>
static void mf_compress_schur_dense(hs_handle* h, int id, const T* SB, int lds, const T* C_, int ldc, const T* M, int ldm, const T* Z, int ldz, int r1, int r2,
                                    hipStream_t stream);
template <class T>
struct MfSchurArgs;
template <class T>
static void mf_compress_schur_dense_batch(hs_handle* h, const std::vector<MfSchurArgs<T>>& a, hipStream_t stream);
template <class F>
static void mf_parallel(hs_handle* h, int count, F&& body);
#include "hs_compress.h"
#include "hs_hssfront.h"
#define MfCoupling NodeH::Coupling
#include "hs_mffront.h"
#include "hs_dist.h"

static double front_flops(double ni, double nb) { return (2.0 / 3.0) * ni * ni * ni + 2.0 * ni * ni * nb + 2.0 * ni * nb * nb; }

static void dmalloc(void** p, size_t bytes, const char* what) {
  if (bytes == 0) bytes = 256;
  if (hipMalloc(p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    (void)hs_trim();  // parked arenas and the recycled blocks of the HSS / low-rank modules go back to the driver first
    if (hipMalloc(p, bytes) != hipSuccess) {
      *p = nullptr;
      (void)hipGetLastError();
      HS_FAIL(HS_ERR_NOMEM, 0, "hipMalloc of %.3f GiB for %s failed", bytes / 1073741824.0, what);
    }
  }
}
// one of the three big blocks: a parked one of about this size if there is one
static void dmalloc_arena(void** p, size_t* held, size_t bytes, const char* what) {
  bytes = std::max<size_t>(bytes, 256);
  *p = arena_take(bytes);
  if (!*p) dmalloc(p, bytes, what);
  *held = bytes;
}

// ------------------------------------------------------------------------------------------------
// analyze
// ------------------------------------------------------------------------------------------------
template <class T>
static hs_handle* analyze_impl(int64_t n, const int64_t* colptr, const int64_t* rowval, const hs_tree* tree, const hs_options* opts_in,
                               int rank, int nranks, bool plan_only = false) {
  hs_options opts;
  if (opts_in)
    opts = *opts_in;
  else
    hs_options_default(&opts);
  chkopts(opts);
  if (n <= 0 || (!plan_only && (!colptr || !rowval))) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: empty matrix");
  if (!plan_only && colptr[0] != 1) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: colptr must be 1-based (SparseMatrixCSC)");
  if (nranks < 1 || rank < 0 || rank >= nranks) HS_FAIL(HS_ERR_ARGUMENT, rank, "ArgumentError: rank %d of %d", rank, nranks);
  if (!plan_only) require_device();

  hs_handle* h = new hs_handle();
  const bool vta = getenv("HS_VERBOSE_ONESHOT") != nullptr && !plan_only;  // diagnostics: wall time of the stages of the analysis
  auto twa = std::chrono::steady_clock::now();
  auto lapa = [&](const char* what) {
    if (!vta) return;
    (void)hipDeviceSynchronize();
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[hs analyze] %-40s %8.3f s\n", what, std::chrono::duration<double>(now - twa).count());
    twa = now;
  };
  try {
    h->is_complex = sizeof(T) == 16;
    h->n = n;
    h->opts = opts;
    h->rank = rank;
    h->nranks = nranks;
    memset(&h->stats, 0, sizeof h->stats);
    h->nuser = tree ? (int)tree->nnodes : 0;
    SplitTree split;
    make_split_tree(tree, opts, nranks, split, n, plan_only ? nullptr : colptr, plan_only ? nullptr : rowval);  // hs_split.h: large compressed fronts are eliminated in slices
    if (split.active) {
      build_plan(h, n, tree);  // validates the user's tree (same errors as without slices) ...
      free_lowrank_any(h);
      h->nodes.clear();
      h->levels.clear();
      h->exchanges.clear();
      tree = &split.view;      // ... which is then replaced by the rewritten one
      build_plan(h, n, tree, &split);
      h->last_of_user = split.last_of_user;
      h->first_of_user = split.first_of_user;
      h->u_ni = split.u_ni;
      h->u_nb = split.u_nb;
      h->u_level = split.u_level;
    } else {
      build_plan(h, n, tree);
    }
    std::vector<NodeH>& N = h->nodes;
    const int nlev = (int)h->levels.size() - 1;
    // depth(nd) in the reference counts levels; negative swlevel counts from the leaves (factorization.jl:8)
    int64_t swlevel = opts.swlevel < 0 ? std::max<int64_t>(nlev + opts.swlevel, 0) : opts.swlevel;
    for (int i = 0; i < h->nreal; ++i)  // compression_flag of factorization.jl:15 (hs_compress.h); levels of the USER's tree when fronts are split
      N[i].compressed = split.active ? (split.cflag[i] != 0) : hs_compression_flag(N[i].level, N[i].ni, N[i].nb, N[i].leaf, swlevel, opts.swsize);
    for (int i = 0; i < h->nreal; ++i)
      if (N[i].dist) N[i].compressed = false;  // fronts eliminated by a group of ranks are eliminated exactly (hs_dist.h)
    // options this plan cannot honour are refused, never dropped (round 2 fell back to the dense-S flow with HS_OK)
    if (opts.mf && split.active) HS_FAIL(HS_ERR_UNSUPPORTED, 0, "hs_options.mf together with hs_options.split: sliced fronts are eliminated on dense blocks");
    if (opts.mf && nranks > 1 && opts.dist_top)
      HS_FAIL(HS_ERR_UNSUPPORTED, 0, "hs_options.mf together with hs_options.dist_top: group fronts are eliminated exactly; use dist_top = 0 (the joins then ship HSS generators)");
    if (opts.hss_d > 0 && nranks > 1) HS_FAIL(HS_ERR_UNSUPPORTED, 0, "hs_options.hss_d with %d ranks: single-rank factorizations only (hs_options.mf runs over ranks)", nranks);
    if (opts.hss_d > 0 && split.active) HS_FAIL(HS_ERR_UNSUPPORTED, 0, "hs_options.hss_d together with hs_options.split");
    if (opts.dist_top && nranks > 1)
      for (int i = 0; i < h->nreal; ++i)
        if (N[i].dist && (split.active ? (split.cflag[i] != 0) : hs_compression_flag(N[i].level, N[i].ni, N[i].nb, N[i].leaf, swlevel, opts.swsize)))
          HS_FAIL(HS_ERR_UNSUPPORTED, i, "hs_options.dist_top: node %d (level %d) is above the rank cut AND flagged for compression (swlevel = %lld); group fronts are eliminated "
                                         "exactly -- use swlevel = 0 or dist_top = 0", i, N[i].level, (long long)opts.swlevel);
    if (opts.hss_d > 0 && nranks == 1 && !split.active && swlevel > 0 && !plan_only) {
      // fronts of the compressed levels with a large interior block keep D as an HSS matrix (hs_hssfront.h); the root too
      std::vector<int> where((size_t)n, -1);
      for (int i = 0; i < h->nreal; ++i) {
        NodeH& x = N[i];
        if (x.leaf || x.level > swlevel || x.ni < (int)opts.hss_d * 1024 || !x.mine) continue;
        if (x.nb > 0 && !x.compressed) continue;  // |bnd| < swsize: stays dense like in the reference
        x.hssd = true;
        static const bool ilv_only = getenv("HS_HSS_ORDER") && getenv("HS_HSS_ORDER")[0] == 'i';  // diagnostics: interleave only
        x.ilv = ilv_only ? hss_interleave_perm(h->fidx_host.data() + x.off_fidx, x.ni, x.ni1, n, colptr, rowval, where)
                         : hss_bisect_perm(h->fidx_host.data() + x.off_fidx, x.ni, n, colptr, rowval, where);
      }
    }
    if (opts.mf && swlevel > 0) {
      // matrix-free compressed branch (hs_mffront.h): flagged fronts hand their Schur complement on as an HSS matrix, a parent of two
      // such fronts is assembled from their generators and the sparse couplings of A.  The flags are properties of the TREE: with several
      // ranks (dist_top = 0) a flagged child of another rank's front sends its HSS matrix packed into one buffer (hs_schur_pack / _unpack)
      mf_plan(h, swlevel);
      std::vector<int> where(plan_only ? (size_t)0 : (size_t)n, -1);
      for (int i = 0; i < h->nreal; ++i) {
        NodeH& x = N[i];
        if (x.s_hss && (x.mine || x.ghost)) h->mf_on = true;
        if (!x.mf) continue;
        x.mfd = opts.mf == 1 && !x.hssd;
        x.mfb = opts.mf == 3;
        if (!x.mine || plan_only) continue;
        h->mf_on = true;
        x.mfd = opts.mf == 1 && !x.hssd;  // mf == 2: D of every matrix-free front is ONE HSS matrix; mf == 3: the reference's 2x2 block form
        x.mfb = opts.mf == 3;
        x.hssd = false;  // the matrix-free form supersedes hs_options.hss_d
        x.ilv = hss_bisect_perm(h->fidx_host.data() + x.off_fidx, x.ni, n, colptr, rowval, where);
      }
    }
    for (auto& L : h->levels) {  // dense fronts first: they are eliminated as one batch, compressed fronts one by one
      std::vector<int> ord;
      for (int pass = 0; pass < 4; ++pass)
        for (int id : L.mine)
          if ((N[id].mf ? 3 : (N[id].hssd ? 2 : (int)N[id].compressed)) == pass) ord.push_back(id);
      L.mine = ord;
      L.nplain = 0;
      L.nmf = 0;
      for (int id : L.mine) {
        if (N[id].mf) L.nmf++;
        else if (N[id].compressed && !N[id].hssd) L.nplain++;
      }
      L.h_ni.clear();
      L.h_nb.clear();
      L.ndense = 0;
      L.dmaxni = L.dmaxnb = L.dmaxm = 0;
      for (size_t k = 0; k < L.mine.size(); ++k) {
        NodeH& x = N[L.mine[k]];
        x.batch_pos = (int)k;
        L.h_ni.push_back(x.ni);
        L.h_nb.push_back(x.nb);
        if (!x.compressed && !x.hssd && !x.mf) {
          L.ndense = (int)k + 1;
          L.dmaxni = std::max(L.dmaxni, x.ni);
          L.dmaxnb = std::max(L.dmaxnb, x.nb);
          L.dmaxm = std::max(L.dmaxm, x.m);
        }
      }
    }

    // ---- HBM layout (owned fronts: LF/UR/inv; owned + ghost fronts: SB) ------------------------------
    size_t fac = 0, inv = 0, ints = h->fidx_host.size(), tmpi = 0;
    size_t sbpar[2] = {0, 0};
    size_t cfs_max = 0;
    std::vector<size_t> sb_level_size(h->levels.size(), 0);
    long long woff = 0, poff = 0;
    size_t ndesc = 0;
    for (int lv = (int)h->levels.size() - 1; lv >= 0; --lv) {
      LevelH& L = h->levels[lv];
      L.lf_begin = fac;
      size_t cfs_lv = 0;
      L.desc_off = ndesc;
      ndesc += L.mine.size();
      size_t sb = 0;
      for (int id : L.nodes) {
        NodeH& x = N[id];
        x.ldl = rup(std::max(x.m, 1), 2);
        x.ldu = rup(std::max(x.ni, 1), 2);
        x.lds = rup(std::max(x.nb, 1), 2);
        if (x.mine || x.ghost) {
          x.off_SB = sb;
          // a matrix-free front whose S leaves as HSS never holds it densely; neither does a front of another rank whose S ARRIVES as HSS
          if (!(x.mf && x.s_hss) && !(x.ghost && !x.mine && x.s_hss)) sb += rups((size_t)x.lds * x.nb, 32);
          x.off_cmap = ints;
          ints += x.nb;
        }
        if (!x.mine) continue;
        if (x.mf) {  // no dense front, no dense factors: D, L, R, S live in the HSS / low-rank objects of hs_mffront.h
          x.off_LF = x.off_UR = fac;
          x.off_inv = x.off_inv256 = inv;
          x.off_ipiv = x.off_rperm = ints;
          x.off_cand = tmpi;
          x.woff = woff;
          if (x.mfd) {  // dense D: what its LU and the sweeps of ldiv! need besides the matrix itself (allocated per factorization)
            const int nblk = (x.ni + HS_PB - 1) / HS_PB;
            inv += (size_t)2 * nblk * HS_PB * HS_PB;
            x.off_inv256 = inv;
            inv += (size_t)2 * ((x.ni + 255) / 256) * 256 * 256;
            ints += x.ni;
            x.off_rperm = ints;
            ints += x.ni;
            x.ncand = ((x.ni + HS_CHUNK - 1) / HS_CHUNK + 1) * HS_PB;
            tmpi += (size_t)2 * x.ncand + HS_PB;
            woff += x.ni;
          }
          continue;
        }
        static const bool cfront_on = !(getenv("HS_COMPACT_D") && getenv("HS_COMPACT_D")[0] == '0');
        x.cfront = cfront_on && x.compressed && !x.hssd && !x.leaf && !x.dist && !split.active && x.nb > 0;
        if (x.cfront) {
          x.ldc = rup(std::max(x.ni, 1), 2);
          x.off_LFc = fac;
          fac += rups((size_t)x.ldc * x.ni, 32);
          x.off_LF = cfs_lv;
          cfs_lv += rups((size_t)x.ldl * x.ni, 32);
          x.off_UR = cfs_lv;
          cfs_lv += rups((size_t)x.ldu * x.nb, 32);
        } else {
          x.off_LF = fac;
          fac += rups((size_t)x.ldl * x.ni, 32);
          x.off_UR = fac;
          fac += rups((size_t)x.ldu * x.nb, 32);
        }
        int nblk = (x.ni + HS_PB - 1) / HS_PB;
        x.off_inv = inv;
        inv += (size_t)2 * nblk * HS_PB * HS_PB;
        x.off_inv256 = inv;  // inverses of the 256 x 256 diagonal blocks of L and U (wide sweeps of ldiv!)
        inv += (size_t)2 * ((x.ni + 255) / 256) * 256 * 256;
        x.off_ipiv = ints;
        ints += x.ni;
        x.off_rperm = ints;
        ints += x.ni;
        if (split.active && x.kind == 1 && !split.newpos[id].empty()) {  // slice order -> original position of the interior DOFs
          x.off_spos = (long long)ints;
          ints += x.oni;
        }
        x.ncand = ((x.ni + HS_CHUNK - 1) / HS_CHUNK + 1) * HS_PB;
        x.off_cand = tmpi;
        tmpi += (size_t)2 * x.ncand + HS_PB;
        x.woff = woff;
        woff += x.ni;
      }
      L.lf_end = fac;
      L.cfs_end = cfs_lv;
      cfs_max = std::max(cfs_max, cfs_lv);
      sb_level_size[lv] = sb;
      if (!opts.keep_schur) sbpar[lv & 1] = std::max(sbpar[lv & 1], sb);
    }
    size_t sb_total = 0;
    if (opts.keep_schur) {
      for (int lv = 0; lv < (int)h->levels.size(); ++lv) {
        h->levels[lv].sb_begin = sb_total;
        sb_total += sb_level_size[lv];
        h->levels[lv].sb_end = sb_total;
      }
      h->sb_kept = true;
    } else {
      sb_total = sbpar[0] + sbpar[1];
      for (int lv = 0; lv < (int)h->levels.size(); ++lv) {
        h->levels[lv].sb_begin = (lv & 1) ? sbpar[0] : 0;
        h->levels[lv].sb_end = h->levels[lv].sb_begin + sb_level_size[lv];
      }
    }
    for (int lv = 0; lv < (int)h->levels.size(); ++lv)
      for (int id : h->levels[lv].nodes)
        if (N[id].mine || N[id].ghost) N[id].off_SB += h->levels[lv].sb_begin;
    for (auto& ex : h->exchanges) {
      ex.hss = N[ex.node].s_hss ? 1 : 0;  // the Schur complement of this node crosses ranks as a packed HSS matrix (size known after its compression)
      ex.nelems = ex.hss ? 0 : (long long)N[ex.node].lds * N[ex.node].nb;
    }
    h->fac_elems = fac;
    h->cfs_elems = cfs_max;
    h->inv_elems = inv;
    h->sb_elems = sb_total;
    h->int_elems = ints;
    if (plan_only) {  // host-side plan only (ownership, exchanges, sizes): no device is touched
      h->stats.n = n;
      h->stats.nnodes = h->nuser > 0 ? h->nuser : h->nreal;
      h->stats.nlevels = nlev;
      h->stats.bytes_factors = (double)(fac + inv) * sizeof(T);
      return h;
    }

    lapa("plan + layout (host)");
    HS_HIP(hipStreamCreate(&h->stream));
    hs_create_lookahead_streams(&h->stream_la, &h->stream2m, &h->stream2);
    HS_HIP(hipEventCreate(&h->ev0));
    HS_HIP(hipEventCreate(&h->ev1));
    if (opts.dist_top && nranks > 1) {
      HS_HIP(hipStreamCreate(&h->stream_comm));
      dmalloc((void**)&h->d_gflags, (size_t)nranks * sizeof(int), "group flags");
      size_t stage = 0;
      const int dist_nb = Sched<T>::env_int("HS_DIST_NB", 512);
      for (int i = 0; i < h->nnodes; ++i)
        if (N[i].dist && N[i].mine) stage = std::max(stage, dist_stage_elems(N[i].ldl, N[i].ni, std::max(dist_nb, 256), sizeof(T)));
      if (stage > 0) {
        dmalloc(&h->d_stage_s, stage * sizeof(T), "block-column send buffer");
        dmalloc(&h->d_stage_r, stage * sizeof(T), "block-column receive buffer");
        dmalloc(&h->d_stage_u, stage * sizeof(T), "block-column send buffer (U parts)");
      }
      // ldiv! inside the library: swap buffers for the boundary values of the joins, every rank's owned DOFs
      int maxnbx = 1;
      for (auto& ex : h->exchanges) maxnbx = std::max(maxnbx, ex.nb);
      dmalloc(&h->d_xs, (size_t)maxnbx * sizeof(T), "boundary swap buffer");
      dmalloc(&h->d_xr, (size_t)maxnbx * sizeof(T), "boundary swap buffer");
      dmalloc(&h->d_xall, (size_t)n * sizeof(T), "solution gather buffer");
      std::vector<int> all;
      h->owned_off.assign(nranks + 1, 0);
      for (int r = 0; r < nranks; ++r) {
        for (int i = 0; i < h->nnodes; ++i)
          if (N[i].owner == r) all.insert(all.end(), h->fidx_host.begin() + N[i].off_fidx, h->fidx_host.begin() + N[i].off_fidx + N[i].ni);
        h->owned_off[r + 1] = (int64_t)all.size();
      }
      dmalloc((void**)&h->d_owned_all, std::max<size_t>(all.size(), 1) * sizeof(int), "owned index lists");
      if (!all.empty()) HS_HIP(hipMemcpy(h->d_owned_all, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
      HS_HIP(hipEventCreateWithFlags(&h->ev_ca, hipEventDisableTiming));
      HS_HIP(hipEventCreateWithFlags(&h->ev_cb, hipEventDisableTiming));
    }
    lapa("streams, events");
    {  // refuse at plan time what cannot fit, and say which flow does: dist_top keeps every group front WHOLE on every rank of its group
      size_t free_b = 0, total_b = 0;
      const double need = ((double)fac + (double)inv + (double)sb_total + (double)h->cfs_elems) * sizeof(T);
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0 && need > (double)total_b) {
        double grp = 0.0;
        int ngrp = 0;
        for (const NodeH& x : N)
          if (x.mine && x.dist) {
            grp += ((double)x.ldl * x.ni + (double)x.ldu * x.nb + (double)x.nb * x.nb) * sizeof(T);
            ++ngrp;
          }
        if (ngrp > 0)
          HS_FAIL(HS_ERR_NOMEM, 0,
                  "OutOfMemoryError: this rank needs %.1f GiB of factors and scratch, the device has %.1f GiB; %.1f GiB of it are %d group fronts that hs_options.dist_top keeps whole on every "
                  "rank of their group (memory per rank does not shrink with the group).  Use the compressed flow over ranks (hs_options.mf with swlevel != 0: a join ships HSS generators) or "
                  "dist_top = 0 with more ranks; tools/size_model.py prints the per-rank bytes of every flow",
                  need / 1073741824.0, (double)total_b / 1073741824.0, grp / 1073741824.0, ngrp);
        HS_FAIL(HS_ERR_NOMEM, 0, "OutOfMemoryError: this rank needs %.1f GiB of factors and scratch, the device has %.1f GiB (tools/size_model.py prints the per-rank bytes of every flow)",
                need / 1073741824.0, (double)total_b / 1073741824.0);
      }
    }
    dmalloc_arena(&h->d_fac, &h->fac_bytes, fac * sizeof(T), "the factors (LF/UR)");
    lapa("hipMalloc of the factor arena");
    dmalloc_arena(&h->d_inv, &h->inv_bytes, inv * sizeof(T), "the inverse diagonal blocks");
    lapa("hipMalloc of the inverse blocks");
    HS_HIP(hipMemset(h->d_inv, 0, inv * sizeof(T)));  // identity padding / unwritten corners must read as zero
    lapa("memset of the inverse blocks");
    dmalloc_arena(&h->d_sb, &h->sb_bytes, sb_total * sizeof(T), "the Schur-complement scratch");
    if (h->cfs_elems > 0) dmalloc_arena(&h->d_cfs, &h->cfs_bytes, h->cfs_elems * sizeof(T), "the scratch fronts of the compressed levels");
    lapa("hipMalloc of the Schur scratch");
    dmalloc((void**)&h->d_int, ints * sizeof(int), "index lists");
    const size_t tmpi_total = tmpi + 2 * (size_t)h->nnodes + 2 * (size_t)n;
    dmalloc((void**)&h->d_tmpi, tmpi_total * sizeof(int), "pivoting scratch");
    HS_HIP(hipMemset(h->d_tmpi, 0, tmpi_total * sizeof(int)));
    h->d_info = h->d_tmpi + tmpi;
    h->d_growth = h->d_info + h->nnodes;  // optimistic-pivoting flags, one per front
    h->d_own = h->d_growth + h->nnodes;
    h->d_pos = h->d_own + n;
    T* dfac = (T*)h->d_fac;
    T* dinv = (T*)h->d_inv;
    T* dsb = (T*)h->d_sb;
    int* dint = h->d_int;

    // ---- sparsity pattern of A (0-based), index lists, cmaps -------------------------------------------
    const int64_t nnz = colptr[n] - 1;
    h->nnz = nnz;
    dmalloc((void**)&h->d_colptr, (n + 1) * sizeof(int64_t), "colptr");
    dmalloc((void**)&h->d_rowval, nnz * sizeof(int32_t), "rowval");
    dmalloc(&h->d_nz, nnz * sizeof(T), "nzval");
    {
      std::vector<int64_t> cp(n + 1);
      for (int64_t j = 0; j <= n; ++j) {
        cp[j] = colptr[j] - 1;
        if (j > 0 && cp[j] < cp[j - 1]) HS_FAIL(HS_ERR_ARGUMENT, j, "ArgumentError: colptr not monotone");
      }
      std::vector<int32_t> rv(nnz);
      for (int64_t e = 0; e < nnz; ++e) {
        int64_t r = rowval[e];
        if (r < 1 || r > n) HS_FAIL(HS_ERR_DIMENSION, e, "BoundsError: rowval[%lld] = %lld outside 1:%lld", (long long)e + 1, (long long)r, (long long)n);
        rv[e] = (int32_t)(r - 1);
      }
      HS_HIP(hipMemcpy(h->d_colptr, cp.data(), (n + 1) * sizeof(int64_t), hipMemcpyHostToDevice));
      HS_HIP(hipMemcpy(h->d_rowval, rv.data(), nnz * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    {
      std::vector<int> hint(ints, 0);
      std::copy(h->fidx_host.begin(), h->fidx_host.end(), hint.begin());
      // cmap of every non-root node: position of each of its bnd DOFs in the parent's front
      for (int i = 0; i < h->nnodes; ++i) {
        NodeH& x = N[i];
        if (x.parent < 0 || !(x.mine || x.ghost)) continue;
        int* cm = hint.data() + x.off_cmap;
        for (int e = 0; e < x.nb; ++e) cm[e] = -1;
        const NodeH& p = N[x.parent];
        if (p.level == 0) {  // pseudo-root: identity
          for (int e = 0; e < x.nb; ++e) cm[e] = e;
          continue;
        }
        bool is_left = (p.left == i);
        // positions in the front the children address: [int; bnd] of the node, or of the user's node for a first slice
        int offi = is_left ? 0 : p.oni1, offb = is_left ? 0 : p.onb1;
        // a first slice with re-ordered interior: original int position -> position in the slice order (hs_split.h)
        const int* np = (split.active && p.kind == 1 && !split.newpos[x.parent].empty()) ? split.newpos[x.parent].data() : nullptr;
        int q = 0;
        for (int64_t e = tree->iloc_ptr[i]; e < tree->iloc_ptr[i + 1]; ++e, ++q) cm[tree->iloc_idx[e] - 1] = np ? np[offi + q] : offi + q;
        q = 0;
        for (int64_t e = tree->bloc_ptr[i]; e < tree->bloc_ptr[i + 1]; ++e, ++q) cm[tree->bloc_idx[e] - 1] = p.oni + offb + q;
      }
      for (int i = 0; i < h->nnodes; ++i) {  // slice order -> original position tables (which child a front position came from)
        NodeH& x = N[i];
        if (x.off_spos < 0) continue;
        const std::vector<int>& np = split.newpos[i];
        int* sp = hint.data() + x.off_spos;
        for (int e = 0; e < x.oni; ++e) sp[np[e]] = e;
      }
      HS_HIP(hipMemcpy(dint, hint.data(), ints * sizeof(int), hipMemcpyHostToDevice));
      if (h->mf_on) mf_sperm(h, hint);
    }
    if (h->mf_on) {
      mf_couplings(h, n, colptr, rowval);
      mf_build_csr(h, n, colptr, rowval);
    }
    lapa("pattern, index lists, cmaps (host + upload)");
    if (nranks > 1) {  // the DOFs this rank eliminates, as one index list
      std::vector<int> owned;
      for (int i = 0; i < h->nnodes; ++i)
        if (N[i].owner == h->rank) owned.insert(owned.end(), h->fidx_host.begin() + N[i].off_fidx, h->fidx_host.begin() + N[i].off_fidx + N[i].ni);
      h->n_owned = (int64_t)owned.size();
      dmalloc((void**)&h->d_owned, owned.size() * sizeof(int), "owned index list");
      if (!owned.empty()) HS_HIP(hipMemcpy(h->d_owned, owned.data(), owned.size() * sizeof(int), hipMemcpyHostToDevice));
    }

    // ---- device descriptors (built once; pointers are fixed from here on) ------------------------------------
    {
      std::vector<NodeDesc<T>> hn;
      std::vector<ScatterDesc<T>> hsc;
      std::vector<SolveNode<T>> sn;
      for (int lv = (int)h->levels.size() - 1; lv >= 0; --lv) {
        LevelH& L = h->levels[lv];
        L.sc_off = hsc.size();
        for (int id : L.mine) {
          NodeH& x = N[id];
          NodeDesc<T> d;
          memset(&d, 0, sizeof d);
          d.LF = (x.cfront ? (T*)h->d_cfs : dfac) + x.off_LF;  // (the solve descriptor below points there too until the level is done: factor_compressed_level)
          d.UR = (x.cfront ? (T*)h->d_cfs : dfac) + x.off_UR;
          d.SB = dsb + x.off_SB;
          int nblk = (x.ni + HS_PB - 1) / HS_PB;
          d.invL = dinv + x.off_inv;
          d.invU = dinv + x.off_inv + (size_t)nblk * HS_PB * HS_PB;
          d.inv256L = dinv + x.off_inv256;
          d.inv256U = d.inv256L + (size_t)((x.ni + 255) / 256) * 256 * 256;
          d.ipiv = dint + x.off_ipiv;
          d.rperm = dint + x.off_rperm;
          d.cand0 = h->d_tmpi + x.off_cand;
          d.cand1 = d.cand0 + x.ncand;
          d.pivlist = d.cand1 + x.ncand;
          d.info = h->d_info + id;
          d.growth = h->d_growth + id;
          d.fidx = dint + x.off_fidx;
          d.ni = x.ni; d.nb = x.nb; d.m = x.m;
          d.ldl = x.ldl; d.ldu = x.ldu; d.lds = x.lds;
          d.ni1 = x.ni1; d.nb1 = x.nb1;
          d.s_ni = x.oni; d.s_ni1 = x.oni1; d.s_nb1 = x.onb1;
          d.spos = x.off_spos >= 0 ? dint + x.off_spos : nullptr;
          d.isleaf = x.leaf ? 1 : 0;
          d.node = id;
          d.finalize();
          hn.push_back(d);
          SolveNode<T> q;
          memset(&q, 0, sizeof q);
          q.LF = d.LF; q.UR = d.UR; q.invL = d.invL; q.invU = d.invU;
          q.inv256L = dinv + x.off_inv256;
          q.inv256U = q.inv256L + (size_t)((x.ni + 255) / 256) * 256 * 256;
          q.rperm = d.rperm; q.fidx = d.fidx;
          q.ni = x.ni; q.nb = x.nb; q.m = x.m; q.ldl = x.ldl; q.ldu = x.ldu;
          q.compressed = x.compressed ? 1 : 0;
          q.mrows = x.compressed ? x.ni : x.m;
          if (x.hssd || x.mf) q.ni = q.nb = q.m = q.mrows = 0;  // the dense sweeps skip it: solve_hss_fwd / solve_hss_bwd (hs_hssfront.h)
          q.woff = x.woff;
          q.poff = poff;
          poff += (long long)((x.nb + 511) / 512) * x.ni;
          sn.push_back(q);
          h->flops += front_flops(x.ni, x.nb);
          for (int side = 0; side < 2; ++side) {
            int c = side == 0 ? x.left : x.right;
            if (x.leaf || c < 0) continue;
            const NodeH& ch = N[c];
            if (ch.nb == 0 || x.mf) continue;  // a matrix-free front reads its children's generators, not their dense S
            ScatterDesc<T> sc;
            sc.S = dsb + ch.off_SB;
            sc.cmap = dint + ch.off_cmap;
            sc.nbc = ch.nb;
            sc.lds = ch.lds;
            sc.parent = x.batch_pos;
            hsc.push_back(sc);
            L.maxnbc = std::max(L.maxnbc, ch.nb);
          }
        }
        L.sc_cnt = hsc.size() - L.sc_off;
      }
      dmalloc(&h->d_nodes, hn.size() * sizeof(NodeDesc<T>), "front descriptors");
      dmalloc(&h->d_sc, hsc.size() * sizeof(ScatterDesc<T>), "scatter descriptors");
      dmalloc(&h->d_solve, sn.size() * sizeof(SolveNode<T>), "solve descriptors");
      if (!hn.empty()) HS_HIP(hipMemcpy(h->d_nodes, hn.data(), hn.size() * sizeof(NodeDesc<T>), hipMemcpyHostToDevice));
      if (!hsc.empty()) HS_HIP(hipMemcpy(h->d_sc, hsc.data(), hsc.size() * sizeof(ScatterDesc<T>), hipMemcpyHostToDevice));
      if (!sn.empty()) HS_HIP(hipMemcpy(h->d_solve, sn.data(), sn.size() * sizeof(SolveNode<T>), hipMemcpyHostToDevice));
    }
    lapa("descriptors");
    dmalloc(&h->d_w1, (size_t)(woff + 1) * sizeof(T), "solve workspace");
    dmalloc(&h->d_w2, (size_t)(woff + 1) * sizeof(T), "solve workspace");
    dmalloc((void**)&h->d_flow, 64 * sizeof(int), "sweep counters");
    HS_HIP(hipMemset(h->d_flow, 0, 64 * sizeof(int)));
    h->flow_bytes = (size_t)(woff + 1) * sizeof(T);
    dmalloc(&h->d_e1, h->flow_bytes, "sweep exchange vector");
    dmalloc(&h->d_e2, h->flow_bytes, "sweep exchange vector");
    HS_HIP(hipHostMalloc((void**)&h->h_flow_err, sizeof(int), hipHostMallocMapped));
    *h->h_flow_err = 0;
    dmalloc(&h->d_part, (size_t)(poff + 1) * sizeof(T), "solve partial sums");
    dmalloc(&h->d_b, (size_t)n * sizeof(T), "right-hand side");

    hs_stats& st = h->stats;
    st.n = n;
    st.nnodes = h->nuser > 0 ? h->nuser : h->nreal;
    st.nlevels = nlev;
    for (int i = 0; i < h->nnodes; ++i) {
      if (!N[i].mine) continue;
      st.max_ni = std::max<int64_t>(st.max_ni, N[i].ni);
      st.max_nb = std::max<int64_t>(st.max_nb, N[i].nb);
      st.bytes_solve += ((double)N[i].ni * N[i].ni + 2.0 * N[i].ni * N[i].nb) * sizeof(T);
    }
    st.bytes_solve += 3.0 * n * sizeof(T);
    st.flops_factor = h->flops * (h->is_complex ? 4.0 : 1.0);
    st.bytes_factors = (double)(fac + inv) * sizeof(T);
    h->static_factor_bytes = st.bytes_factors;
    return h;
  } catch (...) {
    free_handle(h);
    throw;
  }
}

// ------------------------------------------------------------------------------------------------
// numeric factorization
// ------------------------------------------------------------------------------------------------
static void check_handle(const hs_handle* h) {
  if (!h) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: null factorization handle");
}
static void check_device_handle(const hs_handle* h) {
  check_handle(h);
  if (!h->d_nodes) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: handle holds a host-side plan only (hs_plan); use hs_analyze");
}

template <class T>
static void numeric_begin(hs_handle* h, const void* nzval, int on_device) {
  if (!nzval) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: nzval == NULL");
  hipStream_t s = h->stream;
  // the blocks released below are recycled, not returned to the driver (a hipFree would have waited for the device): nothing of the previous
  // factorization -- a solve still in flight -- may be reading them when another stream takes them over
  HS_HIP(hipStreamSynchronize(s));
  free_mfd_buffers(h);
  free_hss_nodes<T>(h);
  free_mf_nodes<T>(h);
  free_lowrank_nodes<T>(h);
  HS_HIP(hipMemcpyAsync(h->d_nz, nzval, h->nnz * sizeof(T), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
  if (h->mf_on) launch_perm_gather<T>((const T*)h->d_nz, h->d_tperm, (T*)h->d_nzr, h->nnz, s);
  HS_HIP(hipMemsetAsync(h->d_own, 0xff, h->n * sizeof(int), s));
  h->prof = Profiler();
  h->prof.on = h->opts.profile != 0;
  HS_HIP(hipEventRecord(h->ev0, s));
  h->numeric_open = true;
  h->factored = false;
}

template <class T>
static void numeric_levels(hs_handle* h, int lv_from, int lv_to) {
  if (!h->numeric_open) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_numeric_levels before hs_numeric_begin");
  const int nl = (int)h->levels.size();
  lv_from = std::min(lv_from, nl - 1);
  lv_to = std::max(lv_to, 0);
  hipStream_t s = h->stream;
  T* dfac = (T*)h->d_fac;
  T* dsb = (T*)h->d_sb;
  const NodeDesc<T>* dn_all = (const NodeDesc<T>*)h->d_nodes;
  const ScatterDesc<T>* dsc_all = (const ScatterDesc<T>*)h->d_sc;
  for (int lv = lv_from; lv >= lv_to; --lv) {
    LevelH& L = h->levels[lv];
    if (L.mine.empty()) continue;
    const NodeDesc<T>* dn = dn_all + L.desc_off;
    const int nb_ = (int)L.mine.size() - L.nmf;  // the fronts that are assembled densely (matrix-free fronts come last in `mine`)
    static const bool progress = getenv("HS_PROGRESS") != nullptr;  // one line per level as it is enqueued (long profiler runs)
    if (progress) fprintf(stderr, "[hs] enqueue level %d (%d fronts)\n", lv, nb_);
    // Optimistic pivoting (Sched::optimistic): the dense fronts of the level are first eliminated with every panel
    // pivoting among its own 32 rows; if any front raised its growth flag the level is assembled again and eliminated with
    // tournament pivoting, and the handle stops trying (a matrix that needs real pivoting needs it everywhere).
    static const bool opt_env = !(getenv("HS_OPTIMISTIC") && getenv("HS_OPTIMISTIC")[0] == '0');
    bool try_opt = opt_env && h->optimistic && (L.ndense > 0 || L.nplain > 0);
    const NodeH* dx = (L.mine.size() == 1 && h->nodes[L.mine[0]].dist) ? &h->nodes[L.mine[0]] : nullptr;  // a front eliminated by its group
    DistFront<T> DF;
    if (dx) {
      if (!h->comm) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_options.dist_top needs a communicator (hs_set_comm) before hs_numeric_levels");
      hipStream_t sc = h->stream_comm;
      static const int dist_nb = Sched<T>::env_int("HS_DIST_NB", 512);
      if (dist_nb < 256 || (dist_nb & (dist_nb - 1))) HS_FAIL(HS_ERR_ARGUMENT, dist_nb, "ArgumentError: HS_DIST_NB = %d must be a power of two >= 256", dist_nb);
      const size_t nblk32 = (dx->ni + HS_PB - 1) / HS_PB;
      T* dinv = (T*)h->d_inv;
      static const int dist_period = std::max(1, Sched<T>::env_int("HS_DIST_PERIOD", 2));
      DF = DistFront<T>{h->comm, dx->glo, dx->gcnt, h->rank, dist_nb, dist_period, sc, dx->ni, dx->nb, dx->m, dx->ldl, dx->ldu, dx->lds,
                        dfac + dx->off_LF, dfac + dx->off_UR, dsb + dx->off_SB, dinv + dx->off_inv, dinv + dx->off_inv + nblk32 * HS_PB * HS_PB,
                        dinv + dx->off_inv256, dinv + dx->off_inv256 + (size_t)((dx->ni + 255) / 256) * 65536, h->d_int + dx->off_ipiv,
                        (T*)h->d_stage_s, (T*)h->d_stage_r, (T*)h->d_stage_u};
      // the join: this rank holds the Schur complement of the child its group eliminated, its partner in the sibling's group the other one
      const NodeH& cl = h->nodes[dx->left];
      const NodeH& cr = h->nodes[dx->right];
      const NodeH& cm = cl.mine ? cl : cr;   // my child
      const NodeH& co = cl.mine ? cr : cl;   // the sibling's (ghost: its buffer is filled here)
      const int partner = cl.mine ? h->rank + cm.gcnt : h->rank - cm.gcnt;
      hipEvent_t ej;
      HS_HIP(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
      HS_HIP(hipEventRecord(ej, s));
      HS_HIP(hipStreamWaitEvent(sc, ej, 0));
      std::vector<HsPiece> sends, recvs;
      if (cm.nb > 0) sends.push_back({partner, cm.ext_sb ? (T*)cm.ext_sb : dsb + cm.off_SB, (size_t)cm.lds * cm.nb * sizeof(T)});
      if (co.nb > 0) recvs.push_back({partner, co.ext_sb ? (T*)co.ext_sb : dsb + co.off_SB, (size_t)co.lds * co.nb * sizeof(T)});
      h->comm->transfer(sends, recvs, sc);
      HS_HIP(hipEventRecord(ej, sc));
      HS_HIP(hipStreamWaitEvent(s, ej, 0));
      HS_HIP(hipStreamSynchronize(sc));
      (void)hipEventDestroy(ej);
      // every member of the group pivots the same way
      try_opt = dist_group_or(h->comm, h->d_gflags, dx->glo, dx->gcnt, h->rank, try_opt ? 0 : 1, sc) == 0 && L.ndense > 0;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
    h->prof.tag = lv;
    hipEvent_t ea = h->prof.begin(s);
    // zero-fill this level's fronts: LF/UR are contiguous per level; SB of the owned fronts only
    // (a ghost child's SB holds received data and must survive)
    if (L.lf_end > L.lf_begin) HS_HIP(hipMemsetAsync(dfac + L.lf_begin, 0, (L.lf_end - L.lf_begin) * sizeof(T), s));
    if (L.cfs_end > 0) HS_HIP(hipMemsetAsync(h->d_cfs, 0, L.cfs_end * sizeof(T), s));  // the scratch fronts of this level's compressed fronts
    if (h->nranks == 1) {
      if (L.sb_end > L.sb_begin) HS_HIP(hipMemsetAsync(dsb + L.sb_begin, 0, (L.sb_end - L.sb_begin) * sizeof(T), s));
    } else {
      for (int id : L.mine) {
        const NodeH& x = h->nodes[id];
        if (x.nb == 0) continue;
        T* sb = x.ext_sb ? (T*)x.ext_sb : dsb + x.off_SB;
        HS_HIP(hipMemsetAsync(sb, 0, (size_t)x.lds * x.nb * sizeof(T), s));
      }
    }
    launch_init_fronts<T>(dn, nb_, L.maxni, s);
    launch_mark<T>(dn, nb_, L.maxm, h->d_own, h->d_pos, s);
    launch_gather<T>(dn, nb_, L.maxm, h->d_colptr, h->d_rowval, (const T*)h->d_nz, h->d_own, h->d_pos, s);
    launch_scatter<T>(dn, dsc_all + L.sc_off, (int)L.sc_cnt, L.maxnbc, s);
    h->prof.end(ea, HS_CAT_ASSEMBLE, s);
    if (L.ndense > 0) {
      Sched<T> sch{dn, L.ndense, L.dmaxni, L.dmaxnb, L.dmaxm, s, &h->prof, L.h_ni.data(), L.h_nb.data(), h->stream2, 0, h->stream_la, h->stream2m};
      sch.sn = (const SolveNode<T>*)h->d_solve + L.desc_off;  // lu_rec leaves the 256x256 inverse diagonal blocks behind
      sch.optimistic = try_opt && attempt == 0;
      if (dx)
        factor_front_dist<T>(sch, DF);
      else
        sch.factor_fronts();
    }
    // the interior blocks of the compressed fronts: the same optimistic attempt, the same redo (hs_compress.h, phase 1)
    if (L.nplain > 0)
      factor_compressed_level<T>(h, L.mine.data() + L.ndense, L.nplain, dn + L.ndense, (const SolveNode<T>*)h->d_solve + L.desc_off + L.ndense, 1, try_opt && attempt == 0);
      if (!(try_opt && attempt == 0)) break;
      std::vector<int> gr(h->nnodes);
      HS_HIP(hipMemcpyAsync(gr.data(), h->d_growth, sizeof(int) * h->nnodes, hipMemcpyDeviceToHost, s));
      HS_HIP(hipStreamSynchronize(s));
      static const bool force_redo = getenv("HS_OPTIMISTIC_FORCE_REDO") != nullptr;  // tests: exercise the redo machinery
      bool redo = force_redo;
      for (int k = 0; k < L.ndense + L.nplain; ++k) redo = redo || gr[L.mine[k]] != 0;
      if (dx) redo = dist_group_or(h->comm, h->d_gflags, dx->glo, dx->gcnt, h->rank, redo ? 1 : 0, h->stream_comm) != 0;
      if (!redo) break;
      h->optimistic = false;
      if (h->opts.verbose) fprintf(stderr, "[hs] level %d: a pivot outside the diagonal block was needed; redoing the level with tournament pivoting\n", lv);
    }
    if (L.nplain > 0) factor_compressed_level<T>(h, L.mine.data() + L.ndense, L.nplain, dn + L.ndense, (const SolveNode<T>*)h->d_solve + L.desc_off + L.ndense, 2);  // hs_compress.h, steps B-F
    if (nb_ > L.ndense + L.nplain) factor_hss_fronts<T>(h, L.mine.data() + L.ndense + L.nplain, nb_ - L.ndense - L.nplain, dn + L.ndense + L.nplain);  // hs_hssfront.h
    {  // F2: a flagged front eliminated densely (a leaf) still hands its S on as an HSS matrix (factorization.jl:45-59)
      std::vector<int> todo;
      for (int k = 0; k < L.ndense; ++k)
        if (h->nodes[L.mine[k]].s_hss) todo.push_back(L.mine[k]);
      if (!todo.empty()) {
        std::vector<MfSchurArgs<T>> args;
        for (int id : todo) {
          const NodeH& x = h->nodes[id];
          args.push_back(MfSchurArgs<T>{id, x.ext_sb ? (const T*)x.ext_sb : dsb + x.off_SB, x.lds, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0});
        }
        mf_compress_schur_dense_batch<T>(h, args, s);
      }
    }
    if (L.nmf > 0) factor_mf_fronts<T>(h, L.mine.data() + nb_, L.nmf);  // hs_mffront.h
    static const bool lvl_env = getenv("HS_VERBOSE_LEVELS") != nullptr;
    if (h->opts.profile || lvl_env) {  // per-level wall time (HS_VERBOSE_LEVELS=1 prints it at hs_numeric_end)
      hipEvent_t e = nullptr;
      (void)hipEventCreate(&e);
      (void)hipEventRecord(e, s);
      h->level_events.push_back({lv, e});
    }
  }
}

static void numeric_end(hs_handle* h) {
  if (!h->numeric_open) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_numeric_end before hs_numeric_begin");
  hipStream_t s = h->stream;
  HS_HIP(hipEventRecord(h->ev1, s));
  HS_HIP(hipStreamSynchronize(s));
  h->numeric_open = false;
  float ms = 0.f;
  HS_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  h->stats.t_total = ms * 1e-3;
  Profiler& prof = h->prof;
  prof.collect();
  if (!h->level_events.empty()) {
    const char* ev = getenv("HS_VERBOSE_LEVELS");
    hipEvent_t prev = h->ev0;
    for (auto& le : h->level_events) {
      float lms = 0.f;
      (void)hipEventElapsedTime(&lms, prev, le.second);
      if (ev && ev[0] == '1') {
        const LevelH& L = h->levels[le.first];
        double fl = 0.0;
        for (int id : L.mine) fl += front_flops(h->nodes[id].ni, h->nodes[id].nb);
        if (h->is_complex) fl *= 4.0;
        fprintf(stderr, "[hs] level %2d: %4zu fronts, max (ni,nb)=(%d,%d)  %9.3f ms  %8.2f TFLOP/s (minimal count)\n", le.first, L.mine.size(),
                L.maxni, L.maxnb, lms, lms > 0 ? fl / (lms * 1e-3) / 1e12 : 0.0);
        if (prof.on && le.first >= 0 && le.first < 64) {
          const double* m = prof.ms_tag[le.first];
          fprintf(stderr, "[hs]          per-launch events: gemm %.1f  panel %.1f  laswp %.1f  trsm %.1f  assemble %.1f ms\n", m[HS_CAT_GEMM], m[HS_CAT_PANEL],
                  m[HS_CAT_LASWP], m[HS_CAT_TRSM], m[HS_CAT_ASSEMBLE]);
        }
      }
      if (prev != h->ev0) (void)hipEventDestroy(prev);
      prev = le.second;
    }
    if (prev != h->ev0) (void)hipEventDestroy(prev);
    h->level_events.clear();
  }
  h->stats.t_mfma_kernel = prof.ms[HS_CAT_GEMM] * 1e-3;  // gemm_op_kernel only: the TRSM base cases run as trsm_inv_kernel
  h->stats.mfma_kernel_launches = prof.launches[HS_CAT_GEMM];
  h->stats.t_gemm = prof.ms[HS_CAT_GEMM] * 1e-3;
  h->stats.t_panel = prof.ms[HS_CAT_PANEL] * 1e-3;
  h->stats.t_trsm = (prof.ms[HS_CAT_TRSM] + prof.ms[HS_CAT_LASWP]) * 1e-3;
  h->stats.t_assemble = prof.ms[HS_CAT_ASSEMBLE] * 1e-3;
  h->stats.gemm_flops = prof.flops[HS_CAT_GEMM];
  h->stats.gemm_launches = prof.launches[HS_CAT_GEMM];
  h->stats.gemm_bytes = prof.gemm_bytes;
  {  // bytes the factors hold: the dense arena + the HSS / low-rank objects of the compressed fronts of this factorization
    double dyn = 0.0;
    const double esz = h->is_complex ? 16.0 : 8.0;
    for (const NodeH& x : h->nodes) {
      if (x.hss) dyn += (double)hs_hss_bytes((const hs_hss*)x.hss);
      if (x.S_hss) dyn += (double)hs_hss_bytes((const hs_hss*)x.S_hss);
      if (x.hW) dyn += (double)x.hldw * std::max(x.last_rR, 0) * esz;
      if (x.mfd_LF) dyn += ((double)x.mfd_ldl * x.ni + (double)x.mfd_ldu * x.mfd_nbp) * esz;  // dense D with Z_L*U^-1 below it, L^-1*P*C_R
      dyn += ((double)x.nb + x.ni) * (std::max(x.last_rL, 0) + std::max(x.last_rR, 0)) * esz * ((x.lrL || x.lrR) ? 1.0 : 0.0);
    }
    h->stats.bytes_factors = h->static_factor_bytes + dyn;
  }
  std::vector<int> info(h->nnodes);
  HS_HIP(hipMemcpy(info.data(), h->d_info, h->nnodes * sizeof(int), hipMemcpyDeviceToHost));
  for (int i = 0; i < h->nnodes; ++i)
    if (h->nodes[i].mine && info[i] != 0)
      HS_FAIL(HS_ERR_SINGULAR, i, "SingularException(%d): exactly zero pivot in the interior block of node %d (ni=%d, nb=%d)", info[i], i,
              h->nodes[i].ni, h->nodes[i].nb);
  h->factored = true;
}

// ------------------------------------------------------------------------------------------------
// ldiv!
// ------------------------------------------------------------------------------------------------
// dataflow sweeps (kernels_solve_wide.hip): HS_SOLVE_FLOW=0 goes back to one launch per 256 columns
static bool solve_flow_on() {
  static const bool on = !(getenv("HS_SOLVE_FLOW") && getenv("HS_SOLVE_FLOW")[0] == '0');
  return on;
}
// the workgroup-id counter of the next sweep launch: a ring of 64, cleared (in stream order) every time it wraps
static int* flow_counter(hs_handle* h, hipStream_t s) {
  const int slot = h->flow_seq++ % 64;
  if (slot == 0) HS_HIP(hipMemsetAsync(h->d_flow, 0, 64 * sizeof(int), s));
  return h->d_flow + slot;
}
// the exchange vectors hold the sentinel (all bits set) wherever nothing has been published yet: once per sweep direction, every level of
// the sweep has its own range of them
static void flow_arm(hs_handle* h, hipStream_t s) {
  if (!solve_flow_on() || !h->d_e1) return;
  HS_HIP(hipMemsetAsync(h->d_e1, 0xFF, h->flow_bytes, s));
  HS_HIP(hipMemsetAsync(h->d_e2, 0xFF, h->flow_bytes, s));
}
static void flow_check(hs_handle* h) {
  if (h->h_flow_err && *h->h_flow_err) {
    *h->h_flow_err = 0;
    HS_FAIL(HS_ERR_DEVICE, 0, "ldiv!: a sweep workgroup waited for a block that never arrived (dataflow sweeps; HS_SOLVE_FLOW=0 selects the launch-per-step sweeps)");
  }
}
template <class T>
static void solve_fwd(hs_handle* h, T* db, int lv_from, int lv_to, hipStream_t s) {
  flow_arm(h, s);
  const SolveNode<T>* sn = (const SolveNode<T>*)h->d_solve;
  T* w1 = (T*)h->d_w1;
  T* w2 = (T*)h->d_w2;
  const int nl = (int)h->levels.size();
  lv_from = std::min(lv_from, nl - 1);
  lv_to = std::max(lv_to, 0);
  for (int lv = lv_from; lv >= lv_to; --lv) {  // leaves -> root (-> pseudo-root)
    const LevelH& L = h->levels[lv];
    if (L.mine.empty() || L.maxni == 0) continue;
    const SolveNode<T>* dn = sn + L.desc_off;
    const int nb_ = (int)L.mine.size();
    launch_fwd_gather<T>(dn, nb_, L.maxni, db, w1, s);
    static const bool wide = !(getenv("HS_SOLVE_WIDE") && getenv("HS_SOLVE_WIDE")[0] == '0');  // 256 columns per launch (kernels_solve_wide.hip)
    if (solve_flow_on()) {  // the whole level in one launch
      launch_fwd_flow<T>(dn, nb_, L.maxni, L.maxnb, w1, w2, db, (T*)h->d_e1, (T*)h->d_e2, flow_counter(h, s), h->h_flow_err, s);
    } else if (wide) {
      const int nblk = (L.maxni + hs_solve_wide_cols() - 1) / hs_solve_wide_cols();
      for (int blk = 0; blk < nblk; ++blk) launch_fwd_wide<T>(dn, nb_, blk, L.maxm, w1, w2, db, s);
    } else {
      const int nblk = (L.maxni + HS_PB - 1) / HS_PB;
      for (int blk = 0; blk < nblk; ++blk) launch_fwd_step<T>(dn, nb_, blk, L.maxm, w1, w2, db, s);
    }
    solve_lr_fwd<T>(h, lv, db, s);
    solve_hss_fwd<T>(h, lv, db, s);
  }
}
template <class T>
static void solve_bwd(hs_handle* h, T* db, int lv_from, int lv_to, hipStream_t s) {
  flow_arm(h, s);
  const SolveNode<T>* sn = (const SolveNode<T>*)h->d_solve;
  T* w1 = (T*)h->d_w1;
  T* w2 = (T*)h->d_w2;
  T* part = (T*)h->d_part;
  const int nl = (int)h->levels.size();
  lv_from = std::max(lv_from, 0);
  lv_to = std::min(lv_to, nl - 1);
  for (int lv = lv_from; lv <= lv_to; ++lv) {  // root -> leaves
    const LevelH& L = h->levels[lv];
    if (L.mine.empty() || L.maxni == 0) continue;
    const SolveNode<T>* dn = sn + L.desc_off;
    const int nb_ = (int)L.mine.size();
    launch_int_update<T>(dn, nb_, L.maxni, L.maxnb, db, part, w2, w1, s);
    solve_lr_bwd<T>(h, lv, db, s);
    const int nblk = (L.maxni + HS_PB - 1) / HS_PB;
    static const bool wide = !(getenv("HS_SOLVE_WIDE") && getenv("HS_SOLVE_WIDE")[0] == '0');
    if (solve_flow_on()) {
      launch_bwd_flow<T>(dn, nb_, L.maxni, w1, w2, (T*)h->d_e1, (T*)h->d_e2, flow_counter(h, s), h->h_flow_err, s);
    } else if (wide) {
      const int nw = (L.maxni + hs_solve_wide_cols() - 1) / hs_solve_wide_cols();
      for (int blk = nw - 1; blk >= 0; --blk) launch_bwd_wide<T>(dn, nb_, blk, w1, w2, s, blk == nw - 1);
    } else {
      for (int blk = nblk - 1; blk >= 0; --blk) launch_bwd_step<T>(dn, nb_, blk, w1, w2, s);
    }
    launch_bwd_scatter<T>(dn, nb_, L.maxni, db, w2, s);
    solve_hss_bwd<T>(h, lv, db, s);
  }
}

// ldiv! of a dist_top factorization, communication included (every rank passes the same right-hand side and receives the whole
// solution).  Fronts above the cut are held by every rank of their group, so the forward sweep needs the sibling's boundary values
// before each join (one pairwise swap), the backward sweep nothing; at the end every rank sends the solution entries it owns to all.
static void comm_transfer_on(hs_handle* h, const std::vector<HsPiece>& sends, const std::vector<HsPiece>& recvs, hipStream_t s) {
  HS_HIP(hipEventRecord(h->ev_ca, s));
  HS_HIP(hipStreamWaitEvent(h->stream_comm, h->ev_ca, 0));
  h->comm->transfer(sends, recvs, h->stream_comm);
  HS_HIP(hipEventRecord(h->ev_cb, h->stream_comm));
  HS_HIP(hipStreamWaitEvent(s, h->ev_cb, 0));
}
template <class T>
static void solve_dist(hs_handle* h, T* db, hipStream_t s) {
  if (!h->comm) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: ldiv! on a dist_top factorization needs the communicator (hs_set_comm)");
  const int nl = (int)h->levels.size(), cut = h->cut_level, esz = (int)sizeof(T);
  solve_fwd<T>(h, db, nl - 1, cut, s);
  for (int lv = cut - 1; lv >= 1; --lv) {
    const LevelH& L = h->levels[lv];
    if (L.mine.size() != 1 || !h->nodes[L.mine[0]].dist) HS_FAIL(HS_ERR_UNSUPPORTED, lv, "internal: level %d above the cut holds no group front", lv);
    const NodeH& x = h->nodes[L.mine[0]];
    const NodeH& cl = h->nodes[x.left];
    const NodeH& cr = h->nodes[x.right];
    const NodeH& cm = cl.mine ? cl : cr;
    const NodeH& co = cl.mine ? cr : cl;
    const int partner = cl.mine ? h->rank + cm.gcnt : h->rank - cm.gcnt;
    if (cm.nb > 0) launch_pack_idx(h->d_int + cm.off_fidx + cm.ni, cm.nb, db, h->d_xs, esz, s);
    std::vector<HsPiece> sends, recvs;
    if (cm.nb > 0) sends.push_back({partner, h->d_xs, (size_t)cm.nb * sizeof(T)});
    if (co.nb > 0) recvs.push_back({partner, h->d_xr, (size_t)co.nb * sizeof(T)});
    comm_transfer_on(h, sends, recvs, s);
    if (co.nb > 0) launch_unpack_idx(h->d_int + co.off_fidx + co.ni, co.nb, db, h->d_xr, esz, s);
    solve_fwd<T>(h, db, lv, lv, s);
  }
  solve_bwd<T>(h, db, 1, nl - 1, s);
  // gather: all[off[r] ..] = b[owned(r)] on rank r, exchanged all-to-all, scattered back
  T* all = (T*)h->d_xall;
  const int64_t* off = h->owned_off.data();
  const int me = h->rank;
  if (off[me + 1] > off[me]) launch_pack_idx(h->d_owned_all + off[me], (int)(off[me + 1] - off[me]), db, all + off[me], esz, s);
  std::vector<HsPiece> sends, recvs;
  for (int r = 0; r < h->nranks; ++r) {
    if (r == me) continue;
    if (off[me + 1] > off[me]) sends.push_back({r, all + off[me], (size_t)(off[me + 1] - off[me]) * sizeof(T)});
    if (off[r + 1] > off[r]) recvs.push_back({r, all + off[r], (size_t)(off[r + 1] - off[r]) * sizeof(T)});
  }
  comm_transfer_on(h, sends, recvs, s);
  launch_unpack_idx(h->d_owned_all, (int)off[h->nranks], db, all, esz, s);
}

static void check_solve_args(hs_handle* h, bool cplx, int64_t ldc, int64_t ldb, int64_t n, int64_t nrhs) {
  check_handle(h);
  flow_check(h);
  if (!h->factored) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: factorization is not complete");
  if (h->is_complex != cplx) HS_FAIL(HS_ERR_ARGUMENT, 0, "MethodError: eltype of F and B differ");
  if (n != h->n || ldc < n || ldb < n || nrhs < 0)
    HS_FAIL(HS_ERR_DIMENSION, 0, "DimensionMismatch: B has %lld rows, F is %lld x %lld", (long long)n, (long long)h->n, (long long)h->n);
}

template <class T>
static void ldiv_host(hs_handle* h, T* C, int64_t ldc, const T* B, int64_t ldb, int64_t n, int64_t nrhs) {
  check_solve_args(h, sizeof(T) == 16, ldc, ldb, n, nrhs);
  const bool dsolve = h->nranks > 1 && h->opts.dist_top;
  if (h->nranks > 1 && !dsolve) HS_FAIL(HS_ERR_UNSUPPORTED, 0, "hs_ldiv_* on a distributed factorization without hs_options.dist_top: drive hs_solve_*_levels from the host layer");
  hipStream_t s = h->stream;
  double tsum = 0.0;
  const int nl = (int)h->levels.size();
  for (int64_t r = 0; r < nrhs; ++r) {
    T* db = (T*)h->d_b;
    HS_HIP(hipMemcpyAsync(db, B + r * ldb, n * sizeof(T), hipMemcpyHostToDevice, s));
    HS_HIP(hipEventRecord(h->ev0, s));
    if (dsolve) {
      solve_dist<T>(h, db, s);
    } else {
      solve_fwd<T>(h, db, nl - 1, 0, s);
      solve_bwd<T>(h, db, 0, nl - 1, s);
    }
    HS_HIP(hipEventRecord(h->ev1, s));
    HS_HIP(hipMemcpyAsync(C + r * ldc, db, n * sizeof(T), hipMemcpyDeviceToHost, s));
    HS_HIP(hipStreamSynchronize(s));
    flow_check(h);
    float ms = 0.f;
    HS_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    tsum += ms * 1e-3;
  }
  h->stats.t_solve = tsum;
}

template <class T>
static void ldiv_dev(hs_handle* h, T* dC, int64_t ldc, const T* dB, int64_t ldb, int64_t n, int64_t nrhs, void* stream) {
  check_solve_args(h, sizeof(T) == 16, ldc, ldb, n, nrhs);
  const bool dsolve = h->nranks > 1 && h->opts.dist_top;
  if (h->nranks > 1 && !dsolve) HS_FAIL(HS_ERR_UNSUPPORTED, 0, "hs_ldiv_dev_* on a distributed factorization without hs_options.dist_top: drive hs_solve_*_levels from the host layer");
  hipStream_t s = (hipStream_t)stream;
  const int nl = (int)h->levels.size();
  for (int64_t r = 0; r < nrhs; ++r) {
    T* c = dC + r * ldc;
    if (c != dB + r * ldb) HS_HIP(hipMemcpyAsync(c, dB + r * ldb, n * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (dsolve) {
      solve_dist<T>(h, c, s);
      continue;
    }
    solve_fwd<T>(h, c, nl - 1, 0, s);
    solve_bwd<T>(h, c, 0, nl - 1, s);
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int hs_analyze(int is_complex, int64_t n, const int64_t* colptr, const int64_t* rowval, const hs_tree* tree, const hs_options* opts,
                          int64_t rank, int64_t nranks, hs_handle** out) {
  if (!out) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: out == NULL");
    return HS_ERR_ARGUMENT;
  }
  *out = nullptr;
  HS_GUARD(*out = is_complex ? analyze_impl<cplx>(n, colptr, rowval, tree, opts, (int)rank, (int)nranks)
                             : analyze_impl<double>(n, colptr, rowval, tree, opts, (int)rank, (int)nranks));
}

extern "C" int hs_plan(int is_complex, int64_t n, const hs_tree* tree, const hs_options* opts, int64_t rank, int64_t nranks, hs_handle** out) {
  if (!out) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: out == NULL");
    return HS_ERR_ARGUMENT;
  }
  *out = nullptr;
  HS_GUARD(*out = is_complex ? analyze_impl<cplx>(n, nullptr, nullptr, tree, opts, (int)rank, (int)nranks, true)
                             : analyze_impl<double>(n, nullptr, nullptr, tree, opts, (int)rank, (int)nranks, true));
}

extern "C" int hs_numeric_begin(hs_handle* h, const void* nzval, int on_device) {
  HS_GUARD(check_device_handle(h); if (h->is_complex) numeric_begin<cplx>(h, nzval, on_device); else numeric_begin<double>(h, nzval, on_device));
}
extern "C" int hs_numeric_levels(hs_handle* h, int64_t lv_from, int64_t lv_to) {
  HS_GUARD(check_handle(h); if (h->is_complex) numeric_levels<cplx>(h, (int)lv_from, (int)lv_to);
           else numeric_levels<double>(h, (int)lv_from, (int)lv_to));
}
extern "C" int hs_set_comm(hs_handle* h, hs_comm* c) {
  HS_GUARD(check_handle(h); if (c && (c->rank != h->rank || c->nranks != h->nranks))
               HS_FAIL(HS_ERR_ARGUMENT, c->rank, "ArgumentError: communicator is rank %d of %d, the factorization rank %d of %d", c->rank, c->nranks, h->rank, h->nranks);
           h->comm = c);
}
extern "C" int hs_numeric_end(hs_handle* h) { HS_GUARD(check_handle(h); numeric_end(h)); }

template <class T>
static int factor_entry(int64_t n, const int64_t* colptr, const int64_t* rowval, const T* nzval, const hs_tree* tree, const hs_options* opts,
                        hs_handle** out) {
  if (!out) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: out == NULL");
    return HS_ERR_ARGUMENT;
  }
  *out = nullptr;
  hs_handle* h = nullptr;
  try {
    if (!nzval) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: nzval == NULL");
    const bool vt = getenv("HS_VERBOSE_ONESHOT") != nullptr;  // diagnostics: wall time of the stages of the one-shot entry point
    auto tw = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!vt) return;
      auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[hs oneshot] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - tw).count());
      tw = now;
    };
    h = analyze_impl<T>(n, colptr, rowval, tree, opts, 0, 1);
    lap("analyze (plan, uploads, arena)");
    numeric_begin<T>(h, nzval, 0);
    lap("numeric_begin (values)");
    numeric_levels<T>(h, (int)h->levels.size() - 1, 0);
    lap("numeric_levels (enqueue)");
    numeric_end(h);
    lap("numeric_end (device done)");
    if (!h->sb_kept) {  // one-shot path: the Schur scratch is not needed again
      arena_give(h->d_sb, h->sb_bytes);
      h->d_sb = nullptr;
      arena_give(h->d_cfs, h->cfs_bytes);  // (the compact LUs of the compressed fronts live in the factor arena)
      h->d_cfs = nullptr;
    }
    lap("free of the Schur scratch");
    *out = h;
    return HS_OK;
  } catch (const HsError& e) {
    free_handle(h);
    return e.code;
  } catch (int code) {
    free_handle(h);
    return code;
  } catch (const std::bad_alloc&) {
    free_handle(h);
    hs_set_error(HS_ERR_NOMEM, 0, "host allocation failed");
    return HS_ERR_NOMEM;
  }
}

extern "C" int hs_factor_d(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const hs_tree* tree,
                           const hs_options* opts, hs_handle** out) {
  return factor_entry<double>(n, colptr, rowval, nzval, tree, opts, out);
}
extern "C" int hs_factor_z(int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval, const hs_tree* tree,
                           const hs_options* opts, hs_handle** out) {
  return factor_entry<cplx>(n, colptr, rowval, reinterpret_cast<const cplx*>(nzval), tree, opts, out);
}

extern "C" int hs_ldiv_d(hs_handle* F, double* C, int64_t ldc, const double* B, int64_t ldb, int64_t n, int64_t nrhs) {
  HS_GUARD(ldiv_host<double>(F, C, ldc, B, ldb, n, nrhs));
}
extern "C" int hs_ldiv_z(hs_handle* F, double* C, int64_t ldc, const double* B, int64_t ldb, int64_t n, int64_t nrhs) {
  HS_GUARD(ldiv_host<cplx>(F, (cplx*)C, ldc, (const cplx*)B, ldb, n, nrhs));
}
extern "C" int hs_ldiv_dev_d(hs_handle* F, double* dC, int64_t ldc, const double* dB, int64_t ldb, int64_t n, int64_t nrhs, void* stream) {
  HS_GUARD(ldiv_dev<double>(F, dC, ldc, dB, ldb, n, nrhs, stream));
}
extern "C" int hs_ldiv_dev_z(hs_handle* F, double* dC, int64_t ldc, const double* dB, int64_t ldb, int64_t n, int64_t nrhs, void* stream) {
  HS_GUARD(ldiv_dev<cplx>(F, (cplx*)dC, ldc, (const cplx*)dB, ldb, n, nrhs, stream));
}

extern "C" int hs_solve_fwd_levels(hs_handle* h, void* d_b, int64_t lv_from, int64_t lv_to, void* stream) {
  HS_GUARD(check_handle(h); if (h->is_complex) solve_fwd<cplx>(h, (cplx*)d_b, (int)lv_from, (int)lv_to, (hipStream_t)stream);
           else solve_fwd<double>(h, (double*)d_b, (int)lv_from, (int)lv_to, (hipStream_t)stream));
}
extern "C" int hs_solve_bwd_levels(hs_handle* h, void* d_b, int64_t lv_from, int64_t lv_to, void* stream) {
  HS_GUARD(check_handle(h); if (h->is_complex) solve_bwd<cplx>(h, (cplx*)d_b, (int)lv_from, (int)lv_to, (hipStream_t)stream);
           else solve_bwd<double>(h, (double*)d_b, (int)lv_from, (int)lv_to, (hipStream_t)stream));
}

// ---- multi-rank plumbing: who owns what, which Schur complements / boundary vectors cross ranks ----------------
extern "C" int64_t hs_nlevels(const hs_handle* h) { return h ? (int64_t)h->levels.size() - 1 : 0; }
extern "C" int64_t hs_cut_level(const hs_handle* h) { return h ? h->cut_level : 0; }
extern "C" int64_t hs_num_exchanges(const hs_handle* h) { return h ? (int64_t)h->exchanges.size() : 0; }
extern "C" int hs_exchange_info(const hs_handle* h, int64_t k, int64_t* out6) {
  HS_GUARD(check_handle(h);
           if (k < 0 || k >= (int64_t)h->exchanges.size() || !out6) HS_FAIL(HS_ERR_ARGUMENT, k, "BoundsError: exchange %lld", (long long)k);
           const Exchange& e = h->exchanges[k]; out6[0] = e.node; out6[1] = e.level; out6[2] = e.src; out6[3] = e.dst; out6[4] = e.nb;
           out6[5] = e.nelems);
}
// Stream ordering between the library's stream and the host layer's (the communicator's) WITHOUT blocking the host: direction 0 = `other`
// waits for everything the library has enqueued so far (before a send reads a Schur buffer), 1 = the library's stream waits for everything
// enqueued on `other` so far (after a receive wrote one).  Round 2's host layer bracketed every transfer with device-wide synchronisations.
extern "C" int hs_stream_order(hs_handle* h, void* other, int direction) {
  HS_GUARD(check_device_handle(h); hipStream_t o = (hipStream_t)other; hipEvent_t e = nullptr;
           HS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
           hipError_t st = hipEventRecord(e, direction == 0 ? h->stream : o);
           if (st == hipSuccess) st = hipStreamWaitEvent(direction == 0 ? o : h->stream, e, 0);
           (void)hipEventDestroy(e);  // (the runtime keeps the event alive until the wait has been satisfied)
           if (st != hipSuccess) HS_FAIL(HS_ERR_DEVICE, 0, "hs_stream_order: %s", hipGetErrorString(st)));
}
extern "C" int64_t hs_exchange_kind(const hs_handle* h, int64_t k) {
  if (!h || k < 0 || k >= (int64_t)h->exchanges.size()) return -1;
  return h->exchanges[(size_t)k].hss;
}
// ---- a Schur complement that crosses ranks as an HSS matrix (hs_options.mf with nranks > 1, dist_top = 0) ---------------------------------
static NodeH& schur_hss_node(hs_handle* h, int64_t node, bool need_matrix) {
  check_device_handle(h);
  if (node < 0 || node >= h->nnodes) HS_FAIL(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
  NodeH& x = h->nodes[(size_t)node];
  if (!x.s_hss) HS_FAIL(HS_ERR_ARGUMENT, node, "ArgumentError: the Schur complement of node %lld does not travel as an HSS matrix (hs_exchange_kind)", (long long)node);
  if (need_matrix && !x.S_hss)
    HS_FAIL(HS_ERR_ARGUMENT, node, "ArgumentError: node %lld holds no HSS Schur complement yet (factor its level first; a parent on this rank may have absorbed it)", (long long)node);
  return x;
}
extern "C" int hs_schur_pack_size(hs_handle* h, int64_t node, int64_t* bytes) {
  HS_GUARD(if (!bytes) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: bytes == NULL"); NodeH& x = schur_hss_node(h, node, true);
           HS_HIP(hipStreamSynchronize(h->stream));  // the compression that produced it ran on the handle's (or a worker's, already joined) stream
           const int st = hs_hss_pack_size((hs_hss*)x.S_hss, bytes); if (st != 0) throw HsError{st});
}
extern "C" int hs_schur_pack(hs_handle* h, int64_t node, void* dev_buf, int64_t bytes, void* stream) {
  HS_GUARD(NodeH& x = schur_hss_node(h, node, true);
           const int st = hs_hss_pack((hs_hss*)x.S_hss, dev_buf, bytes, stream ? stream : (void*)h->stream); if (st != 0) throw HsError{st};
           if (!h->opts.keep_schur && !(x.parent >= 0 && h->nodes[(size_t)x.parent].mine)) {  // sent away: nothing on this rank reads it again
             hs_hss_free((hs_hss*)x.S_hss);
             x.S_hss = nullptr;
           });
}
extern "C" int hs_schur_unpack(hs_handle* h, int64_t node, const void* dev_buf, int64_t bytes, void* stream) {
  HS_GUARD(NodeH& x = schur_hss_node(h, node, false);
           if (!x.ghost && !x.mine) HS_FAIL(HS_ERR_ARGUMENT, node, "ArgumentError: node %lld is neither owned nor received by rank %d", (long long)node, h->rank);
           if (x.S_hss) { hs_hss_free((hs_hss*)x.S_hss); x.S_hss = nullptr; }
           hs_hss* H = nullptr;
           const int st = hs_hss_unpack(dev_buf, bytes, h->is_complex ? 1 : 0, stream ? stream : (void*)h->stream, &H); if (st != 0) throw HsError{st};
           if (hs_hss_size(H) != x.nb) { hs_hss_free(H); HS_FAIL(HS_ERR_DIMENSION, node, "DimensionMismatch: received an HSS matrix of order %lld for node %lld with |bnd| = %d",
                                                                  (long long)hs_hss_size(H), (long long)node, x.nb); }
           x.S_hss = H;
           { std::lock_guard<std::mutex> lk(g_mf_mu); h->maxrank = std::max<int64_t>(h->maxrank, hs_hss_rank(H)); });
}
// out8 = {hs_options.mf in effect (0: the dense-S flow), matrix-free fronts of this rank, fronts of this rank whose S leaves as HSS, fronts with low-rank L / R,
//         fronts whose D is an HSS matrix (hss_d), group fronts (dist_top), ranks, fronts eliminated in slices}: what a run actually did with its options
extern "C" int hs_flow_info(const hs_handle* h, int64_t* out8) {
  HS_GUARD(check_handle(h); if (!out8) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: out8 == NULL");
           for (int k = 0; k < 8; ++k) out8[k] = 0;
           out8[0] = h->mf_on ? h->opts.mf : 0; out8[6] = h->nranks;
           for (int i = 0; i < h->nreal; ++i) {
             const NodeH& x = h->nodes[(size_t)i];
             if (!x.mine) continue;
             out8[1] += x.mf ? 1 : 0; out8[2] += x.s_hss ? 1 : 0; out8[3] += (x.compressed && !x.leaf) ? 1 : 0; out8[4] += x.hssd ? 1 : 0; out8[5] += x.dist ? 1 : 0;
             out8[7] += x.kind != 0 ? 1 : 0;
           });
}
extern "C" int64_t hs_node_owner(const hs_handle* h, int64_t node) {
  if (!h || node < 0 || node >= h->nnodes) return -1;
  return h->nodes[node].owner;
}

template <class T>
static void set_schur_buffer(hs_handle* h, int node, void* dptr) {
  NodeH& x = h->nodes[node];
  if (!(x.mine || x.ghost)) HS_FAIL(HS_ERR_ARGUMENT, node, "ArgumentError: node %d is neither owned nor received by rank %d", node, h->rank);
  x.ext_sb = dptr;
  T* p = (T*)dptr;
  if (x.mine) {  // the front's own descriptor
    NodeDesc<T>* d = (NodeDesc<T>*)h->d_nodes + h->levels[x.level].desc_off + x.batch_pos;
    HS_HIP(hipMemcpy(&d->SB, &p, sizeof p, hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(&d->mp[2], &p, sizeof p, hipMemcpyHostToDevice));
  }
  if (x.parent >= 0 && h->nodes[x.parent].mine) {  // the parent's scatter descriptor (stored left child first)
    const LevelH& L = h->levels[h->nodes[x.parent].level];
    size_t k = L.sc_off;
    for (int id : L.mine) {
      const NodeH& y = h->nodes[id];
      for (int side = 0; side < 2; ++side) {
        int c = side == 0 ? y.left : y.right;
        if (y.leaf || c < 0 || h->nodes[c].nb == 0) continue;
        if (c == node) {
          ScatterDesc<T>* sc = (ScatterDesc<T>*)h->d_sc + k;
          const T* cp = p;
          HS_HIP(hipMemcpy(&sc->S, &cp, sizeof cp, hipMemcpyHostToDevice));
        }
        ++k;
      }
    }
  }
}
extern "C" int hs_set_schur_buffer(hs_handle* h, int64_t node, void* dptr) {
  HS_GUARD(check_device_handle(h); if (node < 0 || node >= h->nnodes || !dptr) HS_FAIL(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
           if (h->is_complex) set_schur_buffer<cplx>(h, (int)node, dptr); else set_schur_buffer<double>(h, (int)node, dptr));
}
// buf[j] = b[bnd_j(node)] and back: the boundary segment of a front as a contiguous vector
extern "C" int hs_pack_bnd(const hs_handle* h, int64_t node, const void* d_b, void* d_buf, void* stream) {
  HS_GUARD(check_device_handle(h); if (node < 0 || node >= h->nnodes) HS_FAIL(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
           const NodeH& x = h->nodes[node];
           launch_pack_idx(h->d_int + x.off_fidx + x.ni, x.nb, d_b, d_buf, h->is_complex ? 16 : 8, (hipStream_t)stream));
}
extern "C" int hs_unpack_bnd(const hs_handle* h, int64_t node, void* d_b, const void* d_buf, void* stream) {
  HS_GUARD(check_device_handle(h); if (node < 0 || node >= h->nnodes) HS_FAIL(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
           const NodeH& x = h->nodes[node];
           launch_unpack_idx(h->d_int + x.off_fidx + x.ni, x.nb, d_b, d_buf, h->is_complex ? 16 : 8, (hipStream_t)stream));
}
// d_out (zero-filled by the caller) receives the solution entries this rank owns: out[int(node)] = b[int(node)]
extern "C" int hs_extract_owned(const hs_handle* h, const void* d_b, void* d_out, void* stream) {
  HS_GUARD(check_device_handle(h); const int esz = h->is_complex ? 16 : 8;
           if (h->d_owned) { launch_copy_idx(h->d_owned, (int)h->n_owned, d_b, d_out, esz, (hipStream_t)stream); return HS_OK; }
           for (int i = 0; i < h->nnodes; ++i) {  // single rank: every DOF is owned
             const NodeH& x = h->nodes[i];
             if (!x.mine || x.ni == 0) continue;
             launch_copy_idx(h->d_int + x.off_fidx, x.ni, d_b, d_out, esz, (hipStream_t)stream);
           });
}

// ------------------------------------------------------------------------------------------------
// introspection
// ------------------------------------------------------------------------------------------------
extern "C" int64_t hs_maxrank(const hs_handle* F) {
  return F ? F->maxrank : 0;  // max over fronts of rank(L), rank(R) (factornode.jl:49-57); 0 on the dense path
}
// node ids of the C ABI are post-order positions in the USER's tree (the pseudo-root follows them); with split fronts
// (hs_split.h) a user node is a chain of internal nodes
static inline bool hs_is_split(const hs_handle* F, int64_t node) {
  return !F->last_of_user.empty() && node < F->nuser && F->first_of_user[node] != F->last_of_user[node];
}
static inline int hs_internal_id(const hs_handle* F, int64_t node) {
  if (F->last_of_user.empty()) return (int)node;
  return node < F->nuser ? F->last_of_user[node] : (int)(F->nreal + (node - F->nuser));
}
static inline int64_t hs_num_user_nodes(const hs_handle* F) { return F->last_of_user.empty() ? F->nnodes : F->nuser + (F->nnodes - F->nreal); }

extern "C" int hs_node_ranks(const hs_handle* F, int64_t node, int64_t* rank_L, int64_t* rank_R) {
  if (!F || node < 0 || node >= hs_num_user_nodes(F)) {
    hs_set_error(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
    return HS_ERR_ARGUMENT;
  }
  if (hs_is_split(F, node)) {  // a front eliminated in slices: the largest ranks over its slices
    int64_t rl = 0, rr = 0;
    int any = 0;
    for (int id = F->first_of_user[node]; id <= F->last_of_user[node]; ++id) {
      const NodeH& y = F->nodes[id];
      if (!(y.compressed && y.lrL && y.lrR)) continue;
      any = 1;
      rl = std::max<int64_t>(rl, F->is_complex ? ((const LowRank<cplx>*)y.lrL)->r : ((const LowRank<double>*)y.lrL)->r);
      rr = std::max<int64_t>(rr, F->is_complex ? ((const LowRank<cplx>*)y.lrR)->r : ((const LowRank<double>*)y.lrR)->r);
    }
    if (rank_L) *rank_L = rl;
    if (rank_R) *rank_R = rr;
    return any;
  }
  const NodeH& x = F->nodes[hs_internal_id(F, node)];
  int64_t rl = 0, rr = 0;  // 0 = dense Gauss transform, like rank terms of maxrank (factornode.jl:54-55)
  if (x.compressed && x.lrL && x.lrR) {
    if (F->is_complex) {
      rl = ((const LowRank<cplx>*)x.lrL)->r;
      rr = ((const LowRank<cplx>*)x.lrR)->r;
    } else {
      rl = ((const LowRank<double>*)x.lrL)->r;
      rr = ((const LowRank<double>*)x.lrR)->r;
    }
  }
  if (rank_L) *rank_L = rl;
  if (rank_R) *rank_R = rr;
  return x.compressed ? 1 : 0;
}
extern "C" int hs_is_complex(const hs_handle* F) { return F && F->is_complex ? 1 : 0; }
extern "C" int64_t hs_size(const hs_handle* F) { return F ? F->n : 0; }

extern "C" int hs_get_stats(const hs_handle* F, hs_stats* out) {
  if (!F || !out) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: null argument");
    return HS_ERR_ARGUMENT;
  }
  *out = F->stats;
  return HS_OK;
}

extern "C" int hs_node_info(const hs_handle* F, int64_t node, int64_t* ni, int64_t* nb, int64_t* level) {
  if (!F || node < 0 || node >= hs_num_user_nodes(F)) {
    hs_set_error(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
    return HS_ERR_ARGUMENT;
  }
  if (!F->last_of_user.empty() && node < F->nuser) {  // sizes and level of the user's node, however it is eliminated
    if (ni) *ni = F->u_ni[node];
    if (nb) *nb = F->u_nb[node];
    if (level) *level = F->u_level[node];
    return HS_OK;
  }
  const NodeH& x = F->nodes[hs_internal_id(F, node)];
  if (ni) *ni = x.ni;
  if (nb) *nb = x.nb;
  if (level) *level = x.level;
  return HS_OK;
}

template <class T>
static void export_block(const hs_handle* F, const NodeH& x, int which, T* out) {
  if (!x.mine) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: node is owned by rank %d", x.owner);
  if (x.hssd || x.mf) HS_FAIL(HS_ERR_UNSUPPORTED, 0, "the interior block of this node is an HSS matrix (hs_options.hss_d / mf): it has no dense D, L, R blocks");
  if (x.compressed && (which == HS_BLK_LBI || which == HS_BLK_UIB)) {  // dense reconstruction C*Z of the low-rank transform
    const void* lr = which == HS_BLK_LBI ? x.lrL : x.lrR;
    if (!lr) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: node has no compressed Gauss transforms yet");
    lowrank_to_dense<T>(*(const LowRank<T>*)lr, out);
    return;
  }
  const T* base;
  int rows, cols, ld;
  switch (which) {
    case HS_BLK_LU:
      if (x.cfront) { base = (const T*)F->d_fac + x.off_LFc; rows = x.ni; cols = x.ni; ld = x.ldc; break; }
      base = (const T*)F->d_fac + x.off_LF; rows = x.ni; cols = x.ni; ld = x.ldl; break;
    case HS_BLK_LBI: base = (const T*)F->d_fac + x.off_LF + x.ni; rows = x.nb; cols = x.ni; ld = x.ldl; break;
    case HS_BLK_UIB: base = (const T*)F->d_fac + x.off_UR; rows = x.ni; cols = x.nb; ld = x.ldu; break;
    case HS_BLK_S:
      if (!F->sb_kept || !F->d_sb) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: Schur complements were not kept (opts.keep_schur)");
      base = x.ext_sb ? (const T*)x.ext_sb : (const T*)F->d_sb + x.off_SB; rows = x.nb; cols = x.nb; ld = x.lds; break;
    default: HS_FAIL(HS_ERR_ARGUMENT, which, "ArgumentError: unknown block id %d", which);
  }
  if (rows == 0 || cols == 0) return;
  HS_HIP(hipMemcpy2D(out, (size_t)rows * sizeof(T), base, (size_t)ld * sizeof(T), (size_t)rows * sizeof(T), cols, hipMemcpyDeviceToHost));
}

// Host-only test hook (include/hs_kernels.h): the order in which an HSS interior block lists its DOFs (hs_hssfront.h).
extern "C" int hsk_bisect_perm(int64_t n, const int64_t* colptr, const int64_t* rowval, int64_t ni, const int64_t* ids, int64_t* perm_out) {
  if (n <= 0 || ni <= 0 || ni > n || !colptr || !rowval || !ids || !perm_out) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: hsk_bisect_perm needs a pattern and an index set");
    return HS_ERR_ARGUMENT;
  }
  HS_GUARD(std::vector<int> I((size_t)ni); for (int64_t e = 0; e < ni; ++e) {
             if (ids[e] < 1 || ids[e] > n) HS_FAIL(HS_ERR_DIMENSION, e, "BoundsError: index %lld outside 1:%lld", (long long)ids[e], (long long)n);
             I[(size_t)e] = (int)(ids[e] - 1);
           } std::vector<int> where((size_t)n, -1);
           std::vector<int64_t> p = hss_bisect_perm(I.data(), (int)ni, n, colptr, rowval, where);
           for (int64_t e = 0; e < ni; ++e) perm_out[e] = p.empty() ? e : p[(size_t)e]);
}

// F.S of one node as an HssMatrix: `compress(S[perm, perm], cl, cl; atol, rtol)` with perm = [nd_loc.int; nd_loc.bnd] and
// cl = bisection_cluster((length(nd_loc.int), length(nd.bnd)); leafsize) (src/factorization.jl:56-57; for a branch the same
// object comes out of randcompress_adaptive, :109-110).  Needs opts.keep_schur; the HSS matrix keeps S's stored index order.
extern "C" int hs_node_schur_hss(const hs_handle* F, int64_t node, const hs_hss_options* o, hs_hss** out) {
  if (!out) return HS_ERR_ARGUMENT;
  *out = nullptr;
  HS_GUARD(check_device_handle(F); if (node < 0 || node >= hs_num_user_nodes(F)) HS_FAIL(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
           if (hs_is_split(F, node)) HS_FAIL(HS_ERR_UNSUPPORTED, node, "node %lld is eliminated in slices (hs_options.split)", (long long)node);
           const NodeH& x = F->nodes[hs_internal_id(F, node)];
           if (!x.mine) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: node is owned by rank %d", x.owner);
           if (!F->sb_kept || !F->d_sb) HS_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: Schur complements were not kept (opts.keep_schur)");
           if (x.nb <= 0 || x.parent < 0) HS_FAIL(HS_ERR_ARGUMENT, node, "ArgumentError: node %lld has no Schur complement", (long long)node);
           std::vector<int> cm(x.nb);
           HS_HIP(hipMemcpy(cm.data(), F->d_int + x.off_cmap, sizeof(int) * x.nb, hipMemcpyDeviceToHost));
           std::vector<int64_t> perm(x.nb);
           for (int e = 0; e < x.nb; ++e) perm[e] = e;
           std::stable_sort(perm.begin(), perm.end(), [&](int64_t a, int64_t b) { return cm[a] < cm[b]; });  // [int_loc; bnd_loc] in the parent's order
           const NodeH& par = F->nodes[x.parent];
           int n1 = 0;
           if (par.level != 0)
             for (int e = 0; e < x.nb; ++e) n1 += cm[e] >= 0 && cm[e] < par.oni;
           hs_hss_options oo;
           hs_hss_options_default(&oo);
           if (o) {
             oo = *o;
           } else {
             oo.leafsize = F->opts.leafsize;
             oo.atol = F->opts.atol;
             oo.rtol = F->opts.rtol;
             if (F->opts.kest > 0) oo.kest = F->opts.kest;
             oo.seed = F->opts.seed;
           }
           oo.first_split = (n1 > 0 && n1 < x.nb) ? n1 : 0;
           const void* base = x.ext_sb ? x.ext_sb : (F->is_complex ? (const void*)((const cplx*)F->d_sb + x.off_SB) : (const void*)((const double*)F->d_sb + x.off_SB));
           const int st = F->is_complex ? hs_hss_compress_ex_z(x.nb, (const double*)base, x.lds, 1, perm.data(), &oo, nullptr, out)
                                        : hs_hss_compress_ex_d(x.nb, (const double*)base, x.lds, 1, perm.data(), &oo, nullptr, out);
           if (st != 0) throw HsError{st});
}

extern "C" int hs_node_export(const hs_handle* F, int64_t node, int which, double* out) {
  HS_GUARD(check_handle(F); if (node < 0 || node >= hs_num_user_nodes(F) || !out) HS_FAIL(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
           if (hs_is_split(F, node)) HS_FAIL(HS_ERR_UNSUPPORTED, node, "node %lld is eliminated in slices (hs_options.split): it has no single D, L, R", (long long)node);
           const NodeH& x = F->nodes[hs_internal_id(F, node)];
           if (F->is_complex) export_block<cplx>(F, x, which, (cplx*)out); else export_block<double>(F, x, which, out));
}

extern "C" int hs_node_export_piv(const hs_handle* F, int64_t node, int64_t* out) {
  HS_GUARD(check_handle(F); if (node < 0 || node >= hs_num_user_nodes(F) || !out) HS_FAIL(HS_ERR_ARGUMENT, node, "BoundsError: node %lld", (long long)node);
           if (hs_is_split(F, node)) HS_FAIL(HS_ERR_UNSUPPORTED, node, "node %lld is eliminated in slices (hs_options.split): it has no single D, L, R", (long long)node);
           const NodeH& x = F->nodes[hs_internal_id(F, node)]; if (!x.mine) HS_FAIL(HS_ERR_ARGUMENT, node, "ArgumentError: node is owned by rank %d", x.owner);
           std::vector<int> tmp(x.ni);
           if (x.ni) HS_HIP(hipMemcpy(tmp.data(), F->d_int + x.off_rperm, x.ni * sizeof(int), hipMemcpyDeviceToHost));
           for (int i = 0; i < x.ni; ++i) out[i] = tmp[i]);
}

extern "C" int hs_device_info(char* arch_name, int64_t len, int64_t* cu_count, int64_t* hbm_bytes) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    hs_set_error(HS_ERR_DEVICE, 0, "no HIP device available");
    return HS_ERR_DEVICE;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    hs_set_error(HS_ERR_DEVICE, 0, "hipGetDeviceProperties failed");
    return HS_ERR_DEVICE;
  }
  if (arch_name && len > 0) {
    strncpy(arch_name, prop.gcnArchName, (size_t)len - 1);
    arch_name[len - 1] = 0;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return prop.multiProcessorCount;
}
