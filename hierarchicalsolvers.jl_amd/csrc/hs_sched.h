// hs_sched.h -- the level-batched recursive LU schedule (host side; launches only).
//
// Recursive (binary-splitting) right-looking LU over a batch of fronts: all fronts of one tree level
// walk the same split tree over global column indices (multiples of 32), each clipping the ranges
// to its own ni / nb inside the kernels.  Every trailing update is one MFMA GEMM whose inner
// dimension is the width of the finished left half, so almost all flops run at large K.
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/hs_solver.h"
#include "hs_common.h"

// ------------------------------------------------------------------------------------------------
// Optional per-launch timing (opts.profile): a HIP event pair around every launch, on the stream the
// kernels run on, summed per category after the factorization.  Off by default (events add gaps).
enum { HS_CAT_GEMM = 0, HS_CAT_PANEL = 1, HS_CAT_LASWP = 2, HS_CAT_TRSM = 3, HS_CAT_ASSEMBLE = 4, HS_NCAT = 5 };
struct Profiler {
  bool on = false;
  struct Rec {
    hipEvent_t e0, e1;
    int cat;
    double fl;
    int M, N, K, nbatch, tag;
  };
  int tag = 0;                          // set by the caller (tree level): per-level sums for HS_VERBOSE_LEVELS
  double ms_tag[64][HS_NCAT] = {};
  std::vector<Rec> recs;
  double flops[HS_NCAT] = {0, 0, 0, 0, 0};
  double gemm_bytes = 0.0;              // algorithmic bytes of the GEMM launches: A + B read once, C read and written once
  double ms[HS_NCAT] = {0, 0, 0, 0, 0};
  long long launches[HS_NCAT] = {0, 0, 0, 0, 0};
  hipEvent_t begin(hipStream_t s) {
    hipEvent_t e = nullptr;
    if (on) {
      (void)hipEventCreate(&e);
      (void)hipEventRecord(e, s);
    }
    return e;
  }
  void end(hipEvent_t e0, int cat, hipStream_t s, double fl = 0.0, int M = 0, int N = 0, int K = 0, int nbatch = 0) {
    launches[cat]++;
    flops[cat] += fl;
    if (!on) return;
    hipEvent_t e1;
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e1, s);
    recs.push_back({e0, e1, cat, fl, M, N, K, nbatch, tag});
  }
  void collect() {  // call after the stream has been synchronised
    const char* logf = getenv("HS_GEMM_LOG");  // diagnostics: one line per GEMM launch (max M, N, K of the batch, flops, ms)
    FILE* lf = (logf && on) ? fopen(logf, "a") : nullptr;
    for (auto& r : recs) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
        ms[r.cat] += t;
        if (r.tag >= 0 && r.tag < 64) ms_tag[r.tag][r.cat] += t;
      }
      if (lf && r.cat == HS_CAT_GEMM) fprintf(lf, "%d %d %d %d %.6g %.6f\n", r.M, r.N, r.K, r.nbatch, r.fl, t);
      (void)hipEventDestroy(r.e0);
      (void)hipEventDestroy(r.e1);
    }
    if (lf) fclose(lf);
    recs.clear();
  }
};

template <class T>
struct Sched {
  const NodeDesc<T>* dn;
  int nbatch, maxni, maxnb, maxm;
  hipStream_t s;
  Profiler* pf;
  const int* h_ni;  // per-front sizes on the host (exact flop accounting); may be null
  const int* h_nb;
  hipStream_t s2 = nullptr;  // optional high-priority side stream: look-ahead panels for large fronts
  int maxpiv = 0;            // largest pivot-candidate row limit in the batch (0: = maxni)
  hipStream_t s_la = nullptr;  // optional CU-masked pair for the look-ahead schedule: s_la = every CU but a reserved
  hipStream_t s2m = nullptr;   // few, s2m = the reserved ones
  int hiprio = 0;              // 1: this schedule is the look-ahead side stream (its GEMMs raise their wave priority)
  const SolveNode<T>* sn = nullptr;  // solve descriptors of the same batch: lu_rec leaves the inverses of the 256x256 diagonal
                                     // blocks of L and U behind (TRSM base case of 256 rows; ldiv! sweeps); null = 32-row base only
  bool wide = false;                 // the descriptors carry valid inv256L / inv256U for the rows being solved
  int rlim = HS_BIG;                 // rows the panel kernels and the in-group updates of lu_rec may touch (diagonal-block-first groups)
  bool optimistic = false;           // no tournament: every panel pivots among its own 32 rows, panel_l21 checks the multipliers
                                     // (NodeDesc::growth); the caller redoes the batch with the tournament if the flag went up

  // HS_DEBUG_SYNC=1: synchronise after every launch and report the first failing one (diagnostics only)
  void dbg(const char* what, int a = 0, int b = 0, int c = 0, int d = 0) {
    static int on = -1;
    if (on < 0) {
      const char* e = getenv("HS_DEBUG_SYNC");
      on = (e && (e[0] == '1' || e[0] == '2')) ? 1 : 0;
    }
    if (!on) return;
    hipError_t e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) fprintf(stderr, "[hs debug] %s(%d,%d,%d,%d) nbatch=%d maxni=%d maxnb=%d maxm=%d -> %s\n", what, a, b, c, d, nbatch, maxni, maxnb, maxm, hipGetErrorString(e));
  }

  static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
  }
  int rows_of(int mat) const { return mat == HS_MAT_LF ? maxm : (mat == HS_MAT_UR ? maxni : maxnb); }
  int cols_of(int mat) const { return mat == HS_MAT_LF ? maxni : maxnb; }

  void gemm(int cmat, int bmat, int r0, int r1, int c0, int c1, int k0, int k1, int cap = 0) {
    GemmOp op{cmat, bmat, r0, r1, c0, c1, k0, k1, 0, cap, hiprio};
    int M = std::min(r1, rows_of(cmat)) - r0, N = std::min(c1, cols_of(cmat)) - c0, K = std::min(k1, maxni) - k0;
    if (M <= 0 || N <= 0 || K <= 0) return;
    double fl = 0.0;
    if (h_ni) {
      for (int i = 0; i < nbatch; ++i) {
        int ni = h_ni[i], nb = h_nb[i], m = ni + nb;
        int rows = cmat == HS_MAT_LF ? m : (cmat == HS_MAT_UR ? ni : nb), cols = cmat == HS_MAT_LF ? ni : nb;
        double Mi = std::min(r1, rows) - r0, Ni = std::min(c1, cols) - c0, Ki = std::min(k1, ni) - k0;
        if (Mi > 0 && Ni > 0 && Ki > 0) {
          fl += 2.0 * Mi * Ni * Ki;
          pf->gemm_bytes += (Mi * Ki + Ki * Ni + 2.0 * Mi * Ni) * sizeof(T);
        }
      }
      if (sizeof(T) == 16) fl *= 4.0;
    }
    hipEvent_t e0 = pf->begin(s);
    launch_gemm_op<T>(dn, nbatch, M, N, op, s);
    pf->end(e0, HS_CAT_GEMM, s, fl, M, N, K, nbatch);
    dbg("gemm", cmat, r0, c0, k0);
  }
  void panel(int pb, int fuse = 0) {
    int c0 = pb * HS_PB;
    if (c0 >= maxni) return;
    // pivot search: 256-row chunks (4 waves, modest registers: finds a slot next to a running GEMM) by default;
    // HS_TOUR_BIG=1 selects the 1024-row-chunk kernel (fewer stages, but a workgroup needs a whole idle CU)
    static const int big = env_int("HS_TOUR_BIG", 0);
    hipEvent_t e0 = pf->begin(s);
    if (optimistic) {
      fuse |= 4;
    } else if (big) {
      const int BR = hs_tour_block_rows(sizeof(T) == 16);
      int cnt = (maxpiv > 0 ? maxpiv : maxni) - c0, nblk = (cnt + BR - 1) / BR;
      for (int stage = 0;; ++stage) {
        launch_tournament_stage<T>(dn, nbatch, pb, stage, nblk, s);
        dbg("tournament", pb, stage, nblk);
        if (nblk == 1) break;
        cnt = nblk * HS_PB;
        nblk = (cnt + BR - 1) / BR;
      }
    } else {
      int cnt = (maxpiv > 0 ? maxpiv : maxni) - c0, nch = (cnt + HS_CHUNK - 1) / HS_CHUNK;
      for (int round = 0;; ++round) {
        launch_tournament_round<T>(dn, nbatch, pb, round, nch, s);
        dbg("tournament", pb, round, nch);
        if (nch == 1) break;
        cnt = nch * HS_PB;
        nch = (cnt + HS_CHUNK - 1) / HS_CHUNK;
      }
    }
    launch_panel_pivot<T>(dn, nbatch, pb, fuse, s);
    dbg("panel_pivot", pb);
    launch_panel_l21<T>(dn, nbatch, pb, std::min(maxm, rlim) - c0, fuse, s, rlim);
    dbg("panel_l21", pb);
    pf->end(e0, HS_CAT_PANEL, s);
  }
  void laswp(int mat, int c0, int c1, int k0, int k1) {
    int nc = std::min(c1, cols_of(mat)) - c0;
    if (nc <= 0 || k0 >= maxni) return;
    hipEvent_t e0 = pf->begin(s);
    launch_laswp<T>(dn, nbatch, mat, c0, c1, k0, k1, nc, s);
    pf->end(e0, HS_CAT_LASWP, s);
    dbg("laswp", mat, c0, k0, k1);
  }
  // X[r0:r1, c0:c1) <- L[r0:r1, r0:r1]^-1 X
  // 256-row base case: multiply by the stored inverse of the 256x256 diagonal block, two in-place half products
  void trsm256(int mat, int r0, int c0, int c1, int nc, bool upper) {
    const int first = upper ? 5 : 3, second = upper ? 6 : 4;
    for (int code : {first, second}) {
      GemmOp op{mat, mat, r0, r0 + 256, c0, c1, 0, 0, code};
      hipEvent_t e0 = pf->begin(s);
      launch_gemm_op<T>(dn, nbatch, 128, nc, op, s);
      pf->end(e0, HS_CAT_TRSM, s);
      dbg("trsm256", mat, r0, c0, code);
    }
  }
  void trsm_rec(int mat, int r0, int r1, int c0, int c1) {
    if (r0 >= maxni) return;
    int nc = std::min(c1, cols_of(mat)) - c0;
    if (nc <= 0) return;
    if (r1 - r0 == 256 && (sn || wide)) {
      trsm256(mat, r0, c0, c1, nc, false);
      return;
    }
    if (r1 - r0 == 2 * HS_PB || r1 - r0 == 4 * HS_PB) {
      // 64 / 128 rows (the solves inside a 256-column group): one launch instead of the 3 / 7 of the recursion below
      hipEvent_t e0 = pf->begin(s);
      if (launch_trsm_small<T>(dn, nbatch, mat, r0, r1 - r0, c0, c1, nc, s)) {
        pf->end(e0, HS_CAT_TRSM, s);
        dbg("trsm_small", mat, r0, c0, c1);
        return;
      }
      if (e0) (void)hipEventDestroy(e0);
    }
    if (r1 - r0 == HS_PB) {
      // base case: multiply by the stored inverse of the 32x32 unit-lower diagonal block (MFMA GEMM, in place)
      GemmOp op{mat, mat, r0, r0 + HS_PB, c0, c1, 0, 0, 1};
      hipEvent_t e0 = pf->begin(s);
      launch_gemm_op<T>(dn, nbatch, HS_PB, nc, op, s);
      pf->end(e0, HS_CAT_TRSM, s);
      dbg("trsm_blk", mat, r0, c0, c1);
      return;
    }
    int mid = (r0 + r1) / 2;
    trsm_rec(mat, r0, mid, c0, c1);
    if (mid < maxni) {
      gemm(mat, mat, mid, r1, c0, c1, r0, mid);
      trsm_rec(mat, mid, r1, c0, c1);
    }
  }
  // X[r0:r1, c0:c1) <- U[r0:r1, r0:r1]^-1 X  (upper factor of the interior block; backward block substitution)
  void utrsm_rec(int mat, int r0, int r1, int c0, int c1) {
    if (r0 >= maxni) return;
    int nc = std::min(c1, cols_of(mat)) - c0;
    if (nc <= 0) return;
    if (r1 - r0 == 256 && (sn || wide)) {
      trsm256(mat, r0, c0, c1, nc, true);
      return;
    }
    if (r1 - r0 == HS_PB) {
      GemmOp op{mat, mat, r0, r0 + HS_PB, c0, c1, 0, 0, 2};
      hipEvent_t e0 = pf->begin(s);
      launch_gemm_op<T>(dn, nbatch, HS_PB, nc, op, s);
      pf->end(e0, HS_CAT_TRSM, s);
      dbg("utrsm_blk", mat, r0, c0, c1);
      return;
    }
    int mid = (r0 + r1) / 2;
    if (mid < maxni) {
      utrsm_rec(mat, mid, r1, c0, c1);
      gemm(mat, mat, r0, mid, c0, c1, mid, r1);
    }
    utrsm_rec(mat, r0, mid, c0, c1);
  }
  // Optimistic pivoting never looks below the 32 rows of a diagonal block, so a 256-column group does not need the rows below ITS
  // diagonal block until the group is done: the panel chain (pivot, L21, swaps, the 64 / 128-row solves and the K <= 128 updates) runs
  // on the 256 x 256 block alone -- every kernel of it one workgroup, which finds a slot next to a running GEMM at once, where the
  // 128-workgroup L21 step of a 32,768-row front waited ~80 us for its slots -- and the multipliers of all rows below follow as
  // L_below = A_below * inv(U_group): two in-place MFMA products with the stored inverse of the group's U (GemmOp::ainv 7, 8), which
  // also check the growth bound of the rows partial pivoting could have picked.  ComplexF64 (round 3; its tile is 64 columns wide): four
  // in-place products, one per 64-column block from the right (GemmOp::ainv 19, 18, 17, 16) -- at Helmholtz 112^3 the full-height
  // `panel_l21_kernel<cplx>` was 14 % of the device time (0.78 s per factorization: one row per thread, 256 VGPRs, 2,048 FP64 FMAs per row
  // and panel on the vector pipe).  The tournament path keeps the full-height panels.
  bool diag_first() const {
    static const int on = env_int("HS_DIAG_FIRST", 1), on_z = env_int("HS_DIAG_FIRST_Z", 1);
    return on && (sizeof(T) == 8 || on_z) && optimistic && sn && rlim == HS_BIG;
  }
  void lu_rec(int c0, int c1) {
    if (c0 >= maxni) return;
    if (c1 - c0 == 256 && diag_first()) {
      hipEvent_t eg = pf->begin(s);
      if (launch_group256<T>(dn, nbatch, c0 / 256, s)) {  // the whole chain of the diagonal block in one launch (kernels_panel.hip)
        pf->end(eg, HS_CAT_PANEL, s);
        dbg("group256", c0);
      } else {
        if (eg) (void)hipEventDestroy(eg);
        rlim = c0 + 256;
        lu_rec_inner(c0, c1);
        rlim = HS_BIG;
      }
      launch_inv256<T>(sn, nbatch, maxni, s, c0 / 256);
      const int r0 = c0 + 256;
      if (r0 < maxm) {
        if (sizeof(T) == 8) {
          for (int code : {7, 8}) {
            GemmOp op{HS_MAT_LF, HS_MAT_LF, r0, HS_BIG, c0, c1, c0, c1, code, 0, hiprio};
            hipEvent_t e0 = pf->begin(s);
            launch_gemm_op<T>(dn, nbatch, maxm - r0, 128, op, s);
            pf->end(e0, HS_CAT_TRSM, s);
            dbg("l_below", c0, r0, code);
          }
        } else {
          const int wl = std::min(256, maxni - c0);
          for (int q = (wl - 1) / 64; q >= 0; --q) {
            GemmOp op{HS_MAT_LF, HS_MAT_LF, r0, HS_BIG, c0, c1, c0, c1, 16 + q, 0, hiprio};
            hipEvent_t e0 = pf->begin(s);
            launch_gemm_op<T>(dn, nbatch, maxm - r0, 64, op, s);
            pf->end(e0, HS_CAT_TRSM, s);
            dbg("l_below", c0, r0, 16 + q);
          }
        }
      }
      return;
    }
    lu_rec_inner(c0, c1);
    if (sn && c1 - c0 == 256) launch_inv256<T>(sn, nbatch, maxni, s, c0 / 256);  // the block is final: leave its inverses behind
  }
  void lu_rec_inner(int c0, int c1) {
    if (c1 - c0 == HS_PB) {
      panel(c0 / HS_PB);
      return;
    }
    static const int pair = env_int("HS_PANEL_PAIR", 1);
    if (pair && c1 - c0 == 2 * HS_PB) {
      // 64-column pair: the first panel's kernels also swap, solve and update the second panel's columns, the second
      // panel's pivot kernel also swaps the first panel's columns -- 6 launches instead of 10, no sub-tile GEMM
      panel(c0 / HS_PB, 1);
      if (c0 + HS_PB < maxni) panel(c0 / HS_PB + 1, 2);
      return;
    }
    int mid = (c0 + c1) / 2;
    lu_rec(c0, mid);
    if (mid < maxni) {
      laswp(HS_MAT_LF, mid, c1, c0, mid);
      trsm_rec(HS_MAT_LF, c0, mid, mid, c1);
      gemm(HS_MAT_LF, HS_MAT_LF, mid, rlim, mid, c1, c0, mid);
      lu_rec(mid, c1);
      laswp(HS_MAT_LF, c0, mid, mid, c1);
    }
  }
  // Blocked right-looking LU with look-ahead for batches of LARGE fronts (few fronts per level: the
  // 32-column panel chain -- tournament rounds, pivot, L21, swaps -- would otherwise run alone on the
  // chip; at Poisson 128^3 the root front spent more time in it than in its GEMMs).  Block column
  // j+1 is brought up to date first and factored on the side stream while the main stream applies
  // block j to the rest of the trailing matrix and to Aib.  Left swaps are applied once at the end.
  void factor_fronts_lookahead(int NB) {
    // `mn` drives the trailing updates, `side` the panel of the next block column.  When the handle owns
    // CU-masked streams (s_la: all CUs but a reserved few; s2: the reserved ones) the two never compete
    // for a compute unit: without the reservation every one of the ~10 dependent tiny kernels of a
    // 32-column panel step waited for a GEMM workgroup to retire, which doubled the panel chain.
    Sched<T> mn = *this;
    Sched<T> side = *this;
    hipStream_t s_la = (nbatch == 1 && s2m) ? this->s_la : nullptr;  // reservation pays for a lone front (the root):
    hipStream_t s2 = s_la ? s2m : this->s2;                          // batched fronts keep every CU on the GEMMs
    hipEvent_t ev_main, ev_side;
    (void)hipEventCreateWithFlags(&ev_main, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&ev_side, hipEventDisableTiming);
    if (s_la) {
      mn.s = s_la;
      (void)hipEventRecord(ev_main, s);
      (void)hipStreamWaitEvent(s_la, ev_main, 0);
    }
    mn.s2 = nullptr;
    mn.s_la = nullptr;
    side.s = s2;
    side.s2 = nullptr;
    side.s_la = nullptr;
    side.hiprio = 1;
    if (s_la) {
      (void)hipStreamWaitEvent(s2, ev_main, 0);
      side.lu_rec(0, NB);
      (void)hipEventRecord(ev_side, s2);
      (void)hipStreamWaitEvent(mn.s, ev_side, 0);
    } else {
      mn.lu_rec(0, NB);
    }
    // the host may run at most 3 block columns (a few thousand launches) ahead of the device: an unbounded run-ahead
    // filled the queues of a 32,768 front and crashed rocprofv3's queue interception
    // the updates that run while the side stream factors the next panel leave workgroup slots free (every CU can hold
    // two GEMM workgroups; HS_LA_GEMM_CAP of them in total, default 448 of the 512): the ~10 tiny dependent kernels of
    // each 32-column panel step then start at once instead of waiting for a GEMM workgroup to retire
    static const int cap_total = env_int("HS_LA_GEMM_CAP", 448);
    // measured: helps a lone front whose panels run the tournament, hurts batches (uneven fronts) and is not needed once
    // the panel chain is short (optimistic pivoting: 520 -> 500 ms on the 32,768 root without the cap)
    static const int cap_opt = env_int("HS_LA_CAP_OPT", 0);  // 1: cap the concurrent update of a lone front under optimistic pivoting too
    const int gcap = (cap_total > 0 && nbatch == 1 && (!optimistic || cap_opt)) ? cap_total : 0;
    hipEvent_t ev_iter[3];
    for (auto& e : ev_iter) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    int iter = 0;
    for (int c0 = 0; c0 < maxni; c0 += NB, ++iter) {
      const int c1 = c0 + NB, c2 = c1 + NB;
      const bool has_next = c1 < maxni;
      if (iter >= 3) (void)hipEventSynchronize(ev_iter[iter % 3]);
      if (has_next) {
        mn.laswp(HS_MAT_LF, c1, c2, c0, c1);
        mn.trsm_rec(HS_MAT_LF, c0, c1, c1, c2);
        mn.gemm(HS_MAT_LF, HS_MAT_LF, c1, HS_BIG, c1, c2, c0, c1);
        (void)hipEventRecord(ev_main, mn.s);
        (void)hipStreamWaitEvent(s2, ev_main, 0);
        side.lu_rec(c1, c2);
        (void)hipEventRecord(ev_side, s2);
        mn.laswp(HS_MAT_LF, c2, HS_BIG, c0, c1);
        mn.trsm_rec(HS_MAT_LF, c0, c1, c2, HS_BIG);
        mn.gemm(HS_MAT_LF, HS_MAT_LF, c1, HS_BIG, c2, HS_BIG, c0, c1, has_next ? gcap : 0);
      }
      if (maxnb > 0) {
        mn.laswp(HS_MAT_UR, 0, HS_BIG, c0, c1);
        mn.trsm_rec(HS_MAT_UR, c0, c1, 0, HS_BIG);
        mn.gemm(HS_MAT_UR, HS_MAT_UR, c1, HS_BIG, 0, HS_BIG, c0, c1, has_next ? gcap : 0);
      }
      if (has_next) (void)hipStreamWaitEvent(mn.s, ev_side, 0);
      (void)hipEventRecord(ev_iter[iter % 3], mn.s);
    }
    for (auto& e : ev_iter) (void)hipEventDestroy(e);
    for (int c0 = 0; c0 + NB < maxni; c0 += NB) mn.laswp(HS_MAT_LF, c0, c0 + NB, c0 + NB, HS_BIG);
    if (s_la) {  // hand back to the handle's stream: the Schur update may use every CU again
      (void)hipEventRecord(ev_main, s_la);
      (void)hipStreamWaitEvent(s, ev_main, 0);
    }
    if (maxnb > 0) gemm(HS_MAT_SB, HS_MAT_UR, 0, HS_BIG, 0, HS_BIG, 0, HS_BIG);
    (void)hipEventDestroy(ev_main);
    (void)hipEventDestroy(ev_side);
  }
  void factor_fronts() {
    if (maxni <= 0) return;
    // ComplexF64: block columns of 512 (a complex product has 4x the flops per byte, K = 512 costs it nothing) and look-ahead from 1,200 interior
    // columns on -- the 1,458-column fronts of levels 6-7 of Helmholtz 112^3 ran without any: metric workload 3.87 -> 3.82 s.  Float64 loses 2 % with 512
    static const int la_min = env_int("HS_LA_MIN", sizeof(T) == 16 ? 1200 : 1536), la_nb = env_int("HS_LA_NB", sizeof(T) == 16 ? 512 : 1024);
    if (s2 && la_nb >= HS_PB && (la_nb & (la_nb - 1)) == 0 && maxni >= la_min && maxni > la_nb) {
      factor_fronts_lookahead(la_nb);
      return;
    }
    int P2 = HS_PB;
    while (P2 < maxni) P2 *= 2;
    lu_rec(0, P2);
    if (sn && P2 < 256) launch_inv256<T>(sn, nbatch, maxni, s, 0);  // fronts narrower than one 256-block
    if (maxnb > 0) {
      laswp(HS_MAT_UR, 0, HS_BIG, 0, P2);
      trsm_rec(HS_MAT_UR, 0, P2, 0, HS_BIG);
      gemm(HS_MAT_SB, HS_MAT_UR, 0, HS_BIG, 0, HS_BIG, 0, HS_BIG);
    }
  }
};

