// hs_compress.h -- compressed fronts: elimination with LOW-RANK off-diagonal blocks.
//
// Reference: `_factor_branch(..., Val(true))` (src/factorization.jl:78-112) for a front at a level <= swlevel
// with |bnd| >= swsize (:15): L and R are LowRankMatrix objects -- `pqrfact(Abi)` / `pqrfact(Aib)` at
// 0.5*atol, 0.5*rtol (:99-100,171-182), then `Aii^-1` applied to the skinny factor (:174,180) -- and the
// Schur complement is `Abb - Abi*R` with that low-rank R (:228-242, blockmatrix.jl:100).
//
// What is built here, per flagged front (one at a time; they are the few large fronts at the top of the tree):
//   A. P*Aii = L*U            dense, the same recursive / look-ahead LU as every front (rows of Abi stay out of it)
//   B. Aib ~= C_R*Z_R,  Abi ~= C_L*Z_L      randomized row ID on the device (hs_lowrank.hip)
//   C. G = L^-1*P*C_R         (ni x rR)      so that  L^-1*P*Aib ~= G*Z_R        -- the stored "R" transform
//   D. W = U^-1*G             (ni x rR)      = Aii^-1*C_R
//   E. S = Abb - C_L*((Z_L*W)*Z_R)           rank-rL update: 2*nb^2*rL flops instead of 2*ni*nb^2 + 2*ni^2*nb
//   F. Z_L' = Z_L*U^-1        (rL x ni)      so that  Abi*U^-1 ~= C_L*Z_L'       -- the stored "L" transform
// `ldiv!` applies  rhs[bnd] -= C_L*(Z_L'*y)  and  y -= G*(Z_R*rhs[bnd])  (y = L^-1*P*rhs[int]), exactly where the
// dense path applies Abi*U^-1 and L^-1*P*Aib.  `maxrank(F)` = largest rank (src/factornode.jl:49-57).
// NOT built: HSS form of D = Aii (`blockfactor` on HssMatrix blocks, blockmatrix.jl:121-130)
// and of S (`randcompress_adaptive`, factorization.jl:110): D stays a dense LU, S a dense matrix.
#pragma once
#include <chrono>
#include "hs_lowrank.h"

template <class T>
static void free_lowrank_nodes(hs_handle* h) {
  for (auto& x : h->nodes) {
    if (x.lrL) {
      lowrank_free(*(LowRank<T>*)x.lrL);
      delete (LowRank<T>*)x.lrL;
      x.lrL = nullptr;
    }
    if (x.lrR) {
      lowrank_free(*(LowRank<T>*)x.lrR);
      delete (LowRank<T>*)x.lrR;
      x.lrR = nullptr;
    }
  }
  h->maxrank = 0;
}

static void free_lowrank_any(hs_handle* h) {
  if (h->is_complex)
    free_lowrank_nodes<cplx>(h);
  else
    free_lowrank_nodes<double>(h);
  if (h->d_lr_t) (void)hipFree(h->d_lr_t);
  if (h->d_lr_part) (void)hipFree(h->d_lr_part);
  if (h->d_cdesc) (void)hipFree(h->d_cdesc);
  h->d_lr_t = h->d_lr_part = h->d_cdesc = nullptr;
  h->lr_t_elems = h->lr_part_elems = 0;
  h->cdesc_cap = 0;
}

// eliminate the compressed fronts of one level: `dn` = their descriptors on the device (fronts already assembled),
// `ids` their node ids.  Steps A, C, D, E run as ONE batch over the fronts (the panel chains of single fronts would
// otherwise run back to back); B and F are per front.
// phase 0: everything; 1: step A only (the LU of every Aii), with OPTIMISTIC pivoting when `optimistic` is set -- the caller checks the fronts'
// growth flags afterwards and, if one went up, assembles the level again and repeats the step with the tournament (hs_api.hip numeric_levels: the same
// redo as for the dense fronts.  Until round 3 the interior blocks of the compressed fronts always ran the tournament path: full-height panels, 2-4
// dependent tournament rounds and the general pivot kernel per 32 columns -- at Helmholtz 112^3 their LUs took 570 ms for 280 ms of GEMM);
// 2: steps B-F after a phase-1 call.
template <class T>
static void factor_compressed_level(hs_handle* h, const int* ids, int count, const NodeDesc<T>* dn, const SolveNode<T>* sn, int phase = 0, bool optimistic = false) {
  if (count <= 0) return;
  hipStream_t s = h->stream;
  static const bool vt = getenv("HS_VERBOSE_COMPRESS") != nullptr;  // per-step wall times (diagnostics; adds syncs)
  auto tlast = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!vt) return;
    (void)hipStreamSynchronize(s);
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[hs compress] level %d, %d fronts: %-28s %9.3f ms\n", h->nodes[ids[0]].level, count, what,
            std::chrono::duration<double, std::milli>(now - tlast).count());
    tlast = now;
  };
  if (h->cdesc_cap < (size_t)3 * count) {
    if (h->d_cdesc) (void)hipFree(h->d_cdesc);
    h->d_cdesc = nullptr;
    dmalloc(&h->d_cdesc, (size_t)3 * count * sizeof(NodeDesc<T>), "descriptors of the compressed fronts");
    h->cdesc_cap = (size_t)3 * count;
  }
  NodeDesc<T>* dd = (NodeDesc<T>*)h->d_cdesc;
  std::vector<NodeDesc<T>> hd(count), tmp(count);
  HS_HIP(hipMemcpy(hd.data(), dn, sizeof(NodeDesc<T>) * count, hipMemcpyDeviceToHost));
  // fronts that live in the scratch arena (NodeH::cfront): while the level is eliminated their solve descriptors point at the scratch front
  // too (lu_rec builds the 256 x 256 inverse blocks from it); at the end the LU moves to its compact place and the descriptors follow
  std::vector<SolveNode<T>> hsn(count);
  bool any_cfront = false;
  for (int i = 0; i < count; ++i) any_cfront = any_cfront || h->nodes[ids[i]].cfront;
  if (any_cfront) {
    if (!h->d_cfs) HS_FAIL(HS_ERR_ARGUMENT, ids[0], "internal: the scratch arena of the compressed fronts was released (one-shot factorizations cannot be repeated)");
    HS_HIP(hipMemcpy(hsn.data(), sn, sizeof(SolveNode<T>) * count, hipMemcpyDeviceToHost));
    for (int i = 0; i < count; ++i) {
      if (!h->nodes[ids[i]].cfront) continue;
      hsn[i].LF = hd[i].LF;
      hsn[i].ldl = hd[i].ldl;
    }
    HS_HIP(hipMemcpy((void*)sn, hsn.data(), sizeof(SolveNode<T>) * count, hipMemcpyHostToDevice));
  }
  int maxni = 0, maxnb = 0;
  for (int i = 0; i < count; ++i) {
    maxni = std::max(maxni, hd[i].ni);
    maxnb = std::max(maxnb, hd[i].nb);
  }
  int P2 = HS_PB;
  while (P2 < maxni) P2 *= 2;

  // A. LU of every Aii: the same fronts with their boundary rows and columns hidden
  for (int i = 0; i < count; ++i) {
    tmp[i] = hd[i];
    tmp[i].nb = 0;
    tmp[i].nb1 = 0;
    tmp[i].m = hd[i].ni;
    tmp[i].pivrows = hd[i].ni;
    tmp[i].finalize();
  }
  HS_HIP(hipMemcpy(dd, tmp.data(), sizeof(NodeDesc<T>) * count, hipMemcpyHostToDevice));
  if (phase != 2) {
    std::vector<int> hni(count), hnb(count, 0);
    for (int i = 0; i < count; ++i) hni[i] = hd[i].ni;
    Sched<T> sA{dd, count, maxni, 0, maxni, s, &h->prof, hni.data(), hnb.data(), h->stream2, 0, h->stream_la, h->stream2m};
    sA.sn = sn;  // solve descriptors of the same fronts: the 256x256 inverse diagonal blocks are built inside lu_rec
    sA.optimistic = optimistic;
    sA.factor_fronts();
    lap(optimistic ? "A: LU(Aii), optimistic pivoting" : "A: LU(Aii)");
  }
  if (phase == 1) return;

  // B. low-rank forms of the two off-diagonal blocks (tolerances of factorization.jl:99-100).  The sketch width
  // starts from the rank the same front had in the previous factorization of this handle (kest, else 128, the first time).
  std::vector<LowRank<T>*> LL(count), RR(count);
  int maxrL = 0, maxrR = 0;
  {
    std::vector<LowRankJob<T>> jobs(2 * count);
    const int k0 = h->opts.kest > 0 ? (int)h->opts.kest : 128;
    for (int i = 0; i < count; ++i) {
      NodeH& x = h->nodes[ids[i]];
      x.lrL = LL[i] = new LowRank<T>();
      x.lrR = RR[i] = new LowRank<T>();
      const int kL = x.last_rL > 0 ? (x.last_rL + 8 + 31) / 32 * 32 + 32 : k0, kR = x.last_rR > 0 ? (x.last_rR + 8 + 31) / 32 * 32 + 32 : k0;
      const uint64_t seed = (uint64_t)h->opts.seed * 2654435761ull + (uint64_t)ids[i] * 2;
      jobs[2 * i] = LowRankJob<T>{hd[i].UR, hd[i].ldu, hd[i].ni, hd[i].nb, kR, seed, RR[i], 0};
      jobs[2 * i + 1] = LowRankJob<T>{hd[i].LF + hd[i].ni, hd[i].ldl, hd[i].nb, hd[i].ni, kL, seed + 1, LL[i], 0};
    }
    // `pqrfact` at 0.5*atol, 0.5*rtol (factorization.jl:99-100): interpolative form, rank and interpolation from the orthogonalisation of the
    // sketch rows in tournament-pivot order (lowrank_id_batch); HS_LR_QR=0 keeps the LU-based rank rule and factors (diagnostics)
    static const bool lr_qr = !(getenv("HS_LR_QR") && getenv("HS_LR_QR")[0] == '0');
    int st = lr_qr ? lowrank_id_batch<T>(jobs.data(), 2 * count, 0.5 * h->opts.atol, 0.5 * h->opts.rtol, s)
                   : lowrank_compress_batch<T>(jobs.data(), 2 * count, 0.5 * h->opts.atol, 0.5 * h->opts.rtol, s);
    if (st != 0) throw HsError{st};
  }
  for (int i = 0; i < count; ++i) {
    NodeH& x = h->nodes[ids[i]];
    LowRank<T>*lrL = LL[i], *lrR = RR[i];
    x.last_rL = lrL->r;
    x.last_rR = lrR->r;
    maxrL = std::max(maxrL, lrL->r);
    maxrR = std::max(maxrR, lrR->r);
    h->maxrank = std::max<int64_t>(h->maxrank, std::max(lrL->r, lrR->r));
    if (h->opts.verbose)
      fprintf(stderr, "[hs] node %d (level %d, ni=%d, nb=%d): rank(L)=%d rank(R)=%d\n", ids[i], x.level, x.ni, x.nb, lrL->r, lrR->r);
    lowrank_expand<T>(*lrR, s);
    lowrank_expand<T>(*lrL, s);
    if ((lrR->r > 0 && !lrR->Cd) || (lrL->r > 0 && !lrL->Cd)) HS_FAIL(HS_ERR_NOMEM, ids[i], "hipMalloc of a low-rank factor of node %d failed", ids[i]);
  }
  lap("B: compress Aib, Abi");

  std::vector<T*> W(count, nullptr), W2(count, nullptr), W3(count, nullptr), ZLo(count, nullptr);
  std::vector<void*> tofree;
  GemmProb<T>* dgp = nullptr;
  auto cleanup = [&]() {
    for (int i = 0; i < count; ++i) {
      if (W[i]) (void)hipFree(W[i]);
      if (W2[i]) (void)hipFree(W2[i]);
      if (W3[i]) (void)hipFree(W3[i]);
      hs_lr_free(ZLo[i]);
    }
    for (void* p : tofree) hs_lr_free(p);
    if (dgp) (void)hipFree(dgp);
  };
  try {
    if (maxrR > 0) {
      // C. G = L^-1 * P * C_R, in place on the dense factors (fronts with rank 0 take part with zero columns)
      for (int i = 0; i < count; ++i) {
        tmp[i] = hd[i];
        tmp[i].UR = RR[i]->Cd ? RR[i]->Cd : hd[i].UR;
        tmp[i].ldu = RR[i]->Cd ? RR[i]->ldc : hd[i].ldu;
        tmp[i].nb = RR[i]->r;
        tmp[i].nb1 = RR[i]->r;
        tmp[i].m = hd[i].ni;
        tmp[i].finalize();
      }
      HS_HIP(hipMemcpy(dd + count, tmp.data(), sizeof(NodeDesc<T>) * count, hipMemcpyHostToDevice));
      Sched<T> sG{dd + count, count, maxni, maxrR, maxni, s, &h->prof, nullptr, nullptr};
      sG.wide = true;  // step A left inv256L / inv256U of every front behind
      sG.laswp(HS_MAT_UR, 0, HS_BIG, 0, P2);
      sG.trsm_rec(HS_MAT_UR, 0, P2, 0, HS_BIG);
      lap("C: G = L^-1 P C_R");
      if (maxrL > 0) {
        // D. W = U^-1 * G
        for (int i = 0; i < count; ++i) {
          if (RR[i]->r == 0) continue;
          const size_t wel = (size_t)RR[i]->ldc * RR[i]->r;
          dmalloc((void**)&W[i], (wel + 32) * sizeof(T), "Aii^-1*C_R");
          HS_HIP(hipMemcpyAsync(W[i], RR[i]->Cd, wel * sizeof(T), hipMemcpyDeviceToDevice, s));
          tmp[i].UR = W[i];
          tmp[i].finalize();
        }
        HS_HIP(hipMemcpy(dd + 2 * count, tmp.data(), sizeof(NodeDesc<T>) * count, hipMemcpyHostToDevice));
        Sched<T> sW{dd + 2 * count, count, maxni, maxrR, maxni, s, &h->prof, nullptr, nullptr};
        sW.wide = true;
        sW.utrsm_rec(HS_MAT_UR, 0, P2, 0, HS_BIG);
        lap("D: W = U^-1 G");
        // E. S -= C_L * ((Z_L * W) * Z_R): three grouped GEMMs over the fronts
        std::vector<GemmProb<T>> gp(3 * count);
        int mrL = 0, mrR = 0;
        for (int i = 0; i < count; ++i) {
          const int rL = LL[i]->r, rR = RR[i]->r, ni = hd[i].ni, nb = hd[i].nb;
          if (rL == 0 || rR == 0) {  // nothing to subtract: empty problems
            gp[i] = gp[count + i] = gp[2 * count + i] = GemmProb<T>{nullptr, nullptr, nullptr, 0, 0, 0, 2, 2, 2};
            continue;
          }
          const int ldw2 = (rL + 1) / 2 * 2;
          dmalloc((void**)&W2[i], ((size_t)ldw2 * rR + 32) * sizeof(T), "Z_L*W");
          dmalloc((void**)&W3[i], ((size_t)ldw2 * nb + 32) * sizeof(T), "(Z_L*W)*Z_R");
          gp[i] = GemmProb<T>{LL[i]->Z, W[i], W2[i], rL, rR, ni, LL[i]->ldz, RR[i]->ldc, ldw2};
          gp[count + i] = GemmProb<T>{W2[i], RR[i]->Z, W3[i], rL, nb, rR, ldw2, RR[i]->ldz, ldw2};
          gp[2 * count + i] = GemmProb<T>{LL[i]->Cd, W3[i], hd[i].SB, nb, nb, rL, LL[i]->ldc, ldw2, hd[i].lds};
          if (h->nodes[ids[i]].s_hss)  // hs_options.mf: S is compressed from the operator Abb - C_L*(Z_L*W)*Z_R below, never formed
            gp[count + i] = gp[2 * count + i] = GemmProb<T>{nullptr, nullptr, nullptr, 0, 0, 0, 2, 2, 2};
          mrL = std::max(mrL, rL);
          mrR = std::max(mrR, rR);
        }
        if (mrL > 0) {
          dmalloc((void**)&dgp, sizeof(GemmProb<T>) * gp.size(), "GEMM descriptors");
          HS_HIP(hipMemcpy(dgp, gp.data(), sizeof(GemmProb<T>) * gp.size(), hipMemcpyHostToDevice));
          launch_gemm_probs<T>(dgp, count, mrL, mrR, 0, s);
          launch_gemm_probs<T>(dgp + count, count, mrL, maxnb, 0, s);
          launch_gemm_probs<T>(dgp + 2 * count, count, maxnb, maxnb, 1, s);
        }
        lap("E: Schur update");
        {  // the transition of the matrix-free branch: `randcompress_adaptive` on the Schur operator (factorization.jl:108-110)
          std::vector<int> todo;
          for (int i = 0; i < count; ++i)
            if (h->nodes[ids[i]].s_hss && W2[i]) todo.push_back(i);
          std::vector<MfSchurArgs<T>> args;
          for (int i : todo) {
            const int ldw2 = (LL[i]->r + 1) / 2 * 2;
            args.push_back(MfSchurArgs<T>{ids[i], hd[i].SB, hd[i].lds, LL[i]->Cd, LL[i]->ldc, W2[i], ldw2, RR[i]->Z, RR[i]->ldz, LL[i]->r, RR[i]->r});
          }
          mf_compress_schur_dense_batch<T>(h, args, s);  // one batched compression for the fronts of the level (hs_mffront.h)
        }
        lap("S: HSS compression of the Schur operator");
      }
    }
    for (int i = 0; i < count; ++i)  // fronts without a low-rank update (a rank came out 0): S = Abb as assembled
      if (h->nodes[ids[i]].s_hss && !h->nodes[ids[i]].S_hss) mf_compress_schur_dense<T>(h, ids[i], hd[i].SB, hd[i].lds, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, s);
    // F. Z_L' = Z_L * U^-1 (Z_L is consumed)
    {
      std::vector<RtrsmJob<T>> rj(count);
      for (int i = 0; i < count; ++i) {
        LowRank<T>* lrL = LL[i];
        if (lrL->r > 0) dmalloc((void**)&ZLo[i], ((size_t)lrL->ldz * hd[i].ni + 32) * sizeof(T), "Z_L*U^-1");
        rj[i] = RtrsmJob<T>{lrL->Z, lrL->ldz, ZLo[i], lrL->ldz, hd[i].LF, hd[i].ldl, hd[i].invU, hd[i].ni, lrL->r, hd[i].inv256U};
      }
      void* dp = nullptr;
      int e = rtrsm_upper_batch<T>(rj.data(), count, s, &dp);
      if (dp) tofree.push_back(dp);
      if (e != 0) HS_FAIL(HS_ERR_DEVICE, ids[0], "right triangular solves of level %d failed (HIP error %d)", h->nodes[ids[0]].level, e);
    }
    if (any_cfront) {  // the LU of Aii to its compact place in the factor arena; ldiv! reads it there
      for (int i = 0; i < count; ++i) {
        const NodeH& x = h->nodes[ids[i]];
        if (!x.cfront || x.ni == 0) continue;
        T* dst = (T*)h->d_fac + x.off_LFc;
        HS_HIP(hipMemcpy2DAsync(dst, (size_t)x.ldc * sizeof(T), hd[i].LF, (size_t)hd[i].ldl * sizeof(T), (size_t)x.ni * sizeof(T), (size_t)x.ni, hipMemcpyDeviceToDevice, s));
        hsn[i].LF = dst;
        hsn[i].ldl = x.ldc;
      }
      HS_HIP(hipMemcpyAsync((void*)sn, hsn.data(), sizeof(SolveNode<T>) * count, hipMemcpyHostToDevice, s));  // (hsn outlives the synchronisation below)
    }
    HS_HIP(hipStreamSynchronize(s));
    lap("F: Z_L U^-1");
    for (int i = 0; i < count; ++i) {
      if (ZLo[i]) std::swap(LL[i]->Z, ZLo[i]);
      for (LowRank<T>* lr : {LL[i], RR[i]}) {  // the dense factor replaces the trapezoid form
        if (!lr->Cd) continue;
        hs_lr_free(lr->Lp);
        hs_lr_free(lr->rperm);
        lr->Lp = nullptr;
        lr->rperm = nullptr;
      }
    }
  } catch (...) {
    (void)hipStreamSynchronize(s);
    cleanup();
    throw;
  }
  cleanup();
}

template <class T>
static void ensure_lr_workspace(hs_handle* h, int r, int cols) {
  const size_t need_t = (size_t)r + 1, need_p = (size_t)((cols + 511) / 512 + 1) * (size_t)(r + 1);
  if (need_t > h->lr_t_elems) {
    if (h->d_lr_t) (void)hipFree(h->d_lr_t);
    dmalloc(&h->d_lr_t, need_t * sizeof(T), "low-rank workspace");
    h->lr_t_elems = need_t;
  }
  if (need_p > h->lr_part_elems) {
    if (h->d_lr_part) (void)hipFree(h->d_lr_part);
    dmalloc(&h->d_lr_part, need_p * sizeof(T), "low-rank workspace");
    h->lr_part_elems = need_p;
  }
}

// u = C * t subtracted from dst (C dense in original row order, or the trapezoid form of hs_lowrank.hip)
template <class T>
static void lr_apply_C(const LowRank<T>& lr, const T* t, T* dst, const int* didx, hipStream_t s) {
  if (lr.Cd)
    launch_lr_dense<T>(lr.Cd, lr.ldc, lr.rows, lr.r, t, dst, didx, s);
  else
    launch_lr_trap<T>(lr.Lp, lr.ldp, lr.rows, lr.r, lr.rperm, t, dst, didx, s);
}

// forward sweep, after the triangular solves of level lv:  rhs[bnd] -= C_L * (Z_L' * y)
template <class T>
static void solve_lr_fwd(hs_handle* h, int lv, T* db, hipStream_t s) {
  const LevelH& L = h->levels[lv];
  T* w2 = (T*)h->d_w2;  // y = L11^-1 P rhs[int]
  for (int id : L.mine) {
    const NodeH& x = h->nodes[id];
    if (!(x.compressed || x.mfd) || !x.lrL || x.hssd || (x.mf && !x.mfd)) continue;
    const LowRank<T>& lr = *(const LowRank<T>*)x.lrL;
    if (lr.r == 0) continue;
    ensure_lr_workspace<T>(h, lr.r, lr.cols);
    launch_lr_zmul<T>(lr.Z, lr.ldz, lr.r, lr.cols, w2 + x.woff, nullptr, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
    lr_apply_C<T>(lr, (const T*)h->d_lr_t, db, h->d_int + x.off_fidx + x.ni, s);
  }
}

// backward sweep, after w1 = y for the compressed fronts of level lv:  w1 -= G * (Z_R * rhs[bnd])
template <class T>
static void solve_lr_bwd(hs_handle* h, int lv, T* db, hipStream_t s) {
  const LevelH& L = h->levels[lv];
  T* w1 = (T*)h->d_w1;
  for (int id : L.mine) {
    const NodeH& x = h->nodes[id];
    if (!(x.compressed || x.mfd) || !x.lrR || x.hssd || (x.mf && !x.mfd)) continue;
    const LowRank<T>& lr = *(const LowRank<T>*)x.lrR;
    if (lr.r == 0) continue;
    ensure_lr_workspace<T>(h, lr.r, lr.cols);
    launch_lr_zmul<T>(lr.Z, lr.ldz, lr.r, lr.cols, db, h->d_int + x.off_fidx + x.ni, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
    lr_apply_C<T>(lr, (const T*)h->d_lr_t, w1 + x.woff, nullptr, s);
  }
}

// dense reconstruction of a compressed Gauss transform on the host (parity tests): out = C * Z (rows x cols)
template <class T>
static void lowrank_to_dense(const LowRank<T>& lr, T* out) {
  std::vector<T> hZ((size_t)lr.ldz * std::max(lr.cols, 1));
  if (lr.r > 0) HS_HIP(hipMemcpy(hZ.data(), lr.Z, sizeof(T) * (size_t)lr.ldz * lr.cols, hipMemcpyDeviceToHost));
  if (lr.Cd) {
    std::vector<T> hC((size_t)lr.ldc * std::max(lr.r, 1));
    if (lr.r > 0) HS_HIP(hipMemcpy(hC.data(), lr.Cd, sizeof(T) * (size_t)lr.ldc * lr.r, hipMemcpyDeviceToHost));
    for (int c = 0; c < lr.cols; ++c)
      for (int i = 0; i < lr.rows; ++i) {
        T acc = Scal<T>::zero();
        for (int j = 0; j < lr.r; ++j) acc = Scal<T>::fma(hC[(size_t)i + (size_t)j * lr.ldc], hZ[(size_t)j + (size_t)c * lr.ldz], acc);
        out[(size_t)i + (size_t)c * lr.rows] = acc;
      }
    return;
  }
  std::vector<T> hL((size_t)lr.ldp * std::max(lr.k, 1));
  std::vector<int> rp(lr.rows);
  HS_HIP(hipMemcpy(hL.data(), lr.Lp, sizeof(T) * (size_t)lr.ldp * lr.k, hipMemcpyDeviceToHost));
  HS_HIP(hipMemcpy(rp.data(), lr.rperm, sizeof(int) * lr.rows, hipMemcpyDeviceToHost));
  for (int c = 0; c < lr.cols; ++c)
    for (int i = 0; i < lr.rows; ++i) {
      T acc = Scal<T>::zero();
      const int jmax = std::min(lr.r, i + 1);
      for (int j = 0; j < jmax; ++j) {
        T l = (j == i) ? Scal<T>::one() : hL[(size_t)i + (size_t)j * lr.ldp];
        acc = Scal<T>::fma(l, hZ[(size_t)j + (size_t)c * lr.ldz], acc);
      }
      out[(size_t)rp[i] + (size_t)c * lr.rows] = acc;
    }
}
