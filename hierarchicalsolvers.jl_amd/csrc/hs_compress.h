// hs_compress.h -- compressed fronts, phase 1: low-rank Gauss transforms.
//
// Reference: `_factor_branch(..., Val(true))` (src/factorization.jl:78-112) keeps, for a front at a level
// <= swlevel with |bnd| >= swsize (:15), L and R as LowRankMatrix objects (`_lgauss_transform` /
// `_rgauss_transform`, :171-209, tolerances 0.5*atol / 0.5*rtol, :99-100), D as an HSS block
// factorization and S as an HSS matrix (randcompress_adaptive, :110).
//
// What is built here (DESIGN.md section 1 lists the rest as open): for such a front the elimination
// itself runs exactly like the dense path (D = P'LU dense, S exact and dense), then the two Gauss
// transforms  Lbi = Abi*U^-1  and  Uib = L^-1*P*Aib  are replaced by low-rank factors
// (hs_lowrank.hip) and `ldiv!` applies them as  rhs[bnd] -= C*(Z*y),  rhs[int] -= C2*(Z2*rhs[bnd]).
// `maxrank(F)` reports the largest rank like the reference (src/factornode.jl:49-57).
// HSS compression of D and S (rows B2', C2, C3, C6 of SURVEY.md section 8) is NOT part of this phase.
#pragma once
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
  h->d_lr_t = h->d_lr_part = nullptr;
  h->lr_t_elems = h->lr_part_elems = 0;
}

// after the fronts of level `lv` are eliminated: compress the Gauss transforms of the flagged fronts
template <class T>
static void compress_level(hs_handle* h, int lv) {
  LevelH& L = h->levels[lv];
  hipStream_t s = h->stream;
  T* dfac = (T*)h->d_fac;
  const int kinit = h->opts.kest > 0 ? (int)h->opts.kest : 128;
  for (int id : L.mine) {
    NodeH& x = h->nodes[id];
    if (!x.compressed || x.ni == 0 || x.nb == 0) continue;
    LowRank<T>* lrL = new LowRank<T>();
    LowRank<T>* lrR = new LowRank<T>();
    x.lrL = lrL;
    x.lrR = lrR;
    // tolerances of the Gauss transforms: 0.5*atol, 0.5*rtol (factorization.jl:99-100)
    int st = lowrank_compress<T>(dfac + x.off_LF + x.ni, x.ldl, x.nb, x.ni, 0.5 * h->opts.atol, 0.5 * h->opts.rtol, kinit,
                                 (uint64_t)h->opts.seed * 2654435761ull + (uint64_t)id * 2 + 0, s, lrL);
    if (st == 0)
      st = lowrank_compress<T>(dfac + x.off_UR, x.ldu, x.ni, x.nb, 0.5 * h->opts.atol, 0.5 * h->opts.rtol, kinit,
                               (uint64_t)h->opts.seed * 2654435761ull + (uint64_t)id * 2 + 1, s, lrR);
    if (st != 0) throw HsError{st};
    h->maxrank = std::max<int64_t>(h->maxrank, std::max(lrL->r, lrR->r));
    if (h->opts.verbose)
      fprintf(stderr, "[hs] node %d (level %d, ni=%d, nb=%d): rank(L)=%d rank(R)=%d\n", id, x.level, x.ni, x.nb, lrL->r, lrR->r);
  }
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

// forward sweep, after the triangular solves of level lv:  rhs[bnd] -= C * (Z * y)
template <class T>
static void solve_lr_fwd(hs_handle* h, int lv, T* db, hipStream_t s) {
  const LevelH& L = h->levels[lv];
  T* w2 = (T*)h->d_w2;  // y = L11^-1 P rhs[int]
  for (int id : L.mine) {
    const NodeH& x = h->nodes[id];
    if (!x.compressed || !x.lrL) continue;
    const LowRank<T>& lr = *(const LowRank<T>*)x.lrL;
    if (lr.r == 0) continue;
    ensure_lr_workspace<T>(h, lr.r, lr.cols);
    launch_lr_zmul<T>(lr.Z, lr.ldz, lr.r, lr.cols, w2 + x.woff, nullptr, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
    launch_lr_trap<T>(lr.Lp, lr.ldp, lr.rows, lr.r, lr.rperm, (const T*)h->d_lr_t, db, h->d_int + x.off_fidx + x.ni, s);
  }
}

// backward sweep, after w1 = y for the compressed fronts of level lv:  w1 -= C2 * (Z2 * rhs[bnd])
template <class T>
static void solve_lr_bwd(hs_handle* h, int lv, T* db, hipStream_t s) {
  const LevelH& L = h->levels[lv];
  T* w1 = (T*)h->d_w1;
  for (int id : L.mine) {
    const NodeH& x = h->nodes[id];
    if (!x.compressed || !x.lrR) continue;
    const LowRank<T>& lr = *(const LowRank<T>*)x.lrR;
    if (lr.r == 0) continue;
    ensure_lr_workspace<T>(h, lr.r, lr.cols);
    launch_lr_zmul<T>(lr.Z, lr.ldz, lr.r, lr.cols, db, h->d_int + x.off_fidx + x.ni, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
    launch_lr_trap<T>(lr.Lp, lr.ldp, lr.rows, lr.r, lr.rperm, (const T*)h->d_lr_t, w1 + x.woff, nullptr, s);
  }
}

// dense reconstruction of a compressed Gauss transform on the host (parity tests): out = C * Z (rows x cols)
template <class T>
static void lowrank_to_dense(const LowRank<T>& lr, T* out) {
  std::vector<T> hL((size_t)lr.ldp * std::max(lr.k, 1)), hZ((size_t)lr.ldz * std::max(lr.cols, 1));
  std::vector<int> rp(lr.rows);
  HS_HIP(hipMemcpy(hL.data(), lr.Lp, sizeof(T) * (size_t)lr.ldp * lr.k, hipMemcpyDeviceToHost));
  if (lr.r > 0) HS_HIP(hipMemcpy(hZ.data(), lr.Z, sizeof(T) * (size_t)lr.ldz * lr.cols, hipMemcpyDeviceToHost));
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
