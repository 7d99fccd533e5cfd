// hs_hssfront.h -- compressed fronts whose interior block D = Aii is kept as an HSS matrix (hs_options.hss_d).
//
// Reference: `_factor_branch(..., Val(true))` (src/factorization.jl:78-112): `D = blockfactor(Aii)` over HssMatrix blocks
// (src/blockmatrix.jl:121-130), every later `Aii^-1 * X` goes through the HSS solve (`blockldiv!`, :134-156), L and R are
// LowRankMatrix objects (:99-100,171-182) and `S = Abb - Abi*R` (:228-242).  The root is never flagged (:15) but assembles
// its children's HSS blocks (:67,126) and so factors an HSS `D` as well; here it is one of these fronts (nb = 0).
//
// Per front, on the assembled dense front [Aii Aib; Abi Abb] (LF = [Aii; Abi], UR = Aib, SB = Abb):
//   A'. H ~= Aii[q, q]  (hs_hss.hip: randomized compression; q = recursive bisection of the graph of A on the interior
//       DOFs, hss_bisect_perm below) and its ULV-type elimination                           -- replaces the (2/3) ni^3 LU
//   B.  Aib ~= C_R*Z_R,  Abi ~= C_L*Z_L     randomized row IDs (one batch per level, hs_lowrank_batch.hip)
//   C'. W = H^-1 * C_R                       HSS solve with rR right-hand sides
//   E.  S = Abb - C_L*((Z_L*W)*Z_R)          three grouped GEMMs
// `ldiv!`:  forward  t = H^-1 rhs[int];  rhs[bnd] -= C_L*(Z_L*t)      backward  rhs[int] = t - W*(Z_R*rhs[bnd]).
// The dense sweeps skip these fronts (their SolveNode has ni = 0).
#pragma once
#include "../../include/hs_hss.h"
#include "hs_lowrank.h"

// order of the interior positions in which every DOF of int1 is followed by its neighbours in int2 (pattern of A):
// perm[new position] = original position; empty = identity
static std::vector<int64_t> hss_interleave_perm(const int* I, int ni, int ni1, int64_t n, const int64_t* colptr, const int64_t* rowval,
                                                std::vector<int>& where) {
  std::vector<int64_t> perm;
  if (ni1 <= 0 || ni1 >= ni || !colptr || !rowval || (int64_t)where.size() < n) return perm;
  for (int e = 0; e < ni; ++e)
    if (I[e] < 0 || I[e] >= n) return perm;
  for (int e = ni1; e < ni; ++e) where[I[e]] = e;
  std::vector<char> placed(ni, 0);
  perm.reserve(ni);
  for (int e = 0; e < ni1; ++e) {
    placed[e] = 1;
    perm.push_back(e);
    const int64_t g = I[e];
    for (int64_t a = colptr[g] - 1; a < colptr[g + 1] - 1; ++a) {
      const int64_t rr = rowval[a] - 1;
      if (rr < 0 || rr >= n) continue;
      const int q = where[rr];
      if (q >= 0 && !placed[q]) {
        placed[q] = 1;
        perm.push_back(q);
      }
    }
  }
  for (int e = ni1; e < ni; ++e) {
    if (!placed[e]) perm.push_back(e);
    where[I[e]] = -1;
  }
  return perm;
}

// Order of the interior positions by RECURSIVE BISECTION of the graph of A restricted to them (both separator layers and the
// couplings between them): a set is split into the first half of a breadth-first sweep from a pseudo-peripheral vertex and
// the rest, exactly where the cluster tree of the HSS matrix splits its index range (at ceil(size/2)), down to sets of
// `small` vertices.  Every index range of the cluster tree is then a compact patch of the separator: its coupling to the rest
// goes through the patch boundary only.  (The order the elimination tree hands down lists thin strips: measured on the
// 32,768 root of Poisson 128^3, leaves of 256 had rank 220.)  perm[new position] = original position.
static std::vector<int64_t> hss_bisect_perm(const int* I, int ni, int64_t n, const int64_t* colptr, const int64_t* rowval, std::vector<int>& where,
                                            int small = 32) {
  std::vector<int64_t> perm;
  if (ni <= small || !colptr || !rowval || (int64_t)where.size() < n) return perm;
  for (int e = 0; e < ni; ++e)
    if (I[e] < 0 || I[e] >= n) return perm;
  for (int e = 0; e < ni; ++e) where[I[e]] = e;
  // local adjacency (CSR)
  std::vector<int> ap(ni + 1, 0), aj;
  for (int e = 0; e < ni; ++e) {
    const int64_t g = I[e];
    for (int64_t a = colptr[g] - 1; a < colptr[g + 1] - 1; ++a) {
      const int64_t rr = rowval[a] - 1;
      if (rr < 0 || rr >= n || rr == g) continue;
      if (where[rr] >= 0) aj.push_back(where[rr]);
    }
    ap[e + 1] = (int)aj.size();
  }
  for (int e = 0; e < ni; ++e) where[I[e]] = -1;
  std::vector<int> ord(ni), mark(ni, -1), queue(ni), tmp(ni);
  for (int e = 0; e < ni; ++e) ord[e] = e;
  int stamp = 0;
  // breadth-first order of the vertices ord[lo:hi) (restricted to that set), started at `start`; unreached parts follow
  auto bfs = [&](int lo, int hi, int start, int inset) {
    ++stamp;
    int qh = 0, qt = 0, scan = lo;
    auto push = [&](int v) {
      mark[v] = stamp;
      queue[qt++] = v;
    };
    push(start);
    while (qt < hi - lo) {
      if (qh == qt) {  // another component
        while (mark[ord[scan]] == stamp) ++scan;
        push(ord[scan]);
      }
      const int v = queue[qh++];
      for (int a = ap[v]; a < ap[v + 1]; ++a) {
        const int w = aj[a];
        if (tmp[w] == inset && mark[w] != stamp) push(w);
      }
    }
    return qt;
  };
  // explicit stack of ranges; tmp[v] = id of the set v currently belongs to
  std::vector<std::pair<int, int>> st{{0, ni}};
  std::fill(tmp.begin(), tmp.end(), 0);
  int nextid = 1;
  std::vector<int> setid_of_range;
  while (!st.empty()) {
    const auto [lo, hi] = st.back();
    st.pop_back();
    const int sz = hi - lo;
    if (sz <= small) continue;
    const int id = tmp[ord[lo]];
    // pseudo-peripheral start: last vertex of a sweep from an arbitrary one, twice
    int start = ord[lo];
    for (int rep = 0; rep < 2; ++rep) {
      bfs(lo, hi, start, id);
      start = queue[sz - 1];
    }
    bfs(lo, hi, start, id);
    for (int t = 0; t < sz; ++t) ord[lo + t] = queue[t];
    const int mid = lo + (sz + 1) / 2;
    const int idl = nextid++, idr = nextid++;
    for (int t = lo; t < mid; ++t) tmp[ord[t]] = idl;
    for (int t = mid; t < hi; ++t) tmp[ord[t]] = idr;
    st.push_back({lo, mid});
    st.push_back({mid, hi});
  }
  perm.assign(ord.begin(), ord.end());
  return perm;
}

template <class T>
static void free_hss_nodes(hs_handle* h) {
  for (auto& x : h->nodes) {
    if (x.hss) {
      hs_hss_free((hs_hss*)x.hss);
      x.hss = nullptr;
    }
    if (x.hW) {
      (void)hipFree(x.hW);
      x.hW = nullptr;
    }
    if (x.hss2) {
      hs_hss_free((hs_hss*)x.hss2);
      x.hss2 = nullptr;
    }
    if (x.hss_keep) {
      hs_hss_free((hs_hss*)x.hss_keep);
      x.hss_keep = nullptr;
    }
    for (void** q : {&x.bW12, &x.bZ12, &x.bC21, &x.bZ21})
      if (*q) {
        (void)hipFree(*q);
        *q = nullptr;
      }
  }
}
static void free_hss_any(hs_handle* h) {
  free_hss_nodes<double>(h);
  for (auto& x : h->nodes)
    if (x.ht) {
      (void)hipFree(x.ht);
      x.ht = nullptr;
    }
}

template <class T>
static void factor_hss_fronts(hs_handle* h, const int* ids, int count, const NodeDesc<T>* dn) {
  if (count <= 0) return;
  hipStream_t s = h->stream;
  static const bool vt = getenv("HS_VERBOSE_COMPRESS") != nullptr;
  static const int leaf_env = getenv("HS_HSS_LEAF") ? atoi(getenv("HS_HSS_LEAF")) : 256;
  std::vector<NodeDesc<T>> hd(count);
  HS_HIP(hipMemcpy(hd.data(), dn, sizeof(NodeDesc<T>) * count, hipMemcpyDeviceToHost));
  auto tlast = std::chrono::steady_clock::now();
  auto lap = [&](const char* what, int id) {
    if (!vt) return;
    (void)hipStreamSynchronize(s);
    auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[hs hss] node %d (level %d): %-26s %9.3f ms\n", id, h->nodes[id].level, what,
            std::chrono::duration<double, std::milli>(now - tlast).count());
    tlast = now;
  };
  // A'. HSS form of every Aii and its elimination
  for (int i = 0; i < count; ++i) {
    NodeH& x = h->nodes[ids[i]];
    hs_hss_options o;
    hs_hss_options_default(&o);
    o.leafsize = std::max(32, leaf_env);
    const double dsc = std::pow(10.0, -(double)(h->opts.hss_dexp == 0 ? 2 : h->opts.hss_dexp - 1));  // hs_options.hss_dexp
    o.atol = h->opts.atol * dsc;
    o.rtol = h->opts.rtol * dsc;
    // samples to start from: what the previous factorization of this front ended with, else opts.kest, else a guess from the size (the
    // HSS rank of a 2-D separator block grows like its perimeter, ~ sqrt(ni)): every doubling repeats the 4*ni^2*k flops of the samples
    int64_t k0 = 128;
    while (k0 < 4.0 * std::sqrt((double)x.ni)) k0 *= 2;
    o.kest = x.last_k > 0 ? x.last_k : (h->opts.kest > 0 ? h->opts.kest : k0);
    o.seed = h->opts.seed + 31 * (int64_t)ids[i];
    hs_hss* H = nullptr;
    const int64_t* q = x.ilv.empty() ? nullptr : x.ilv.data();
    int st = h->is_complex ? hs_hss_compress_ex_z(hd[i].ni, (const double*)hd[i].LF, hd[i].ldl, 1, q, &o, s, &H)
                           : hs_hss_compress_ex_d(hd[i].ni, (const double*)hd[i].LF, hd[i].ldl, 1, q, &o, s, &H);
    if (st != 0) throw HsError{st};
    x.hss = H;
    lap("HSS compress(Aii)", ids[i]);
    st = hs_hss_factor(H);
    if (st != 0) {
      if (st == HS_ERR_SINGULAR) hs_set_error(HS_ERR_SINGULAR, ids[i], "SingularException: the HSS form of the interior block of node %d is singular", ids[i]);
      throw HsError{st};
    }
    lap("HSS elimination", ids[i]);
    x.last_k = (int)hs_hss_samples(H);
    h->maxrank = std::max<int64_t>(h->maxrank, hs_hss_rank(H));
    if (h->opts.verbose || vt)
      fprintf(stderr, "[hs] node %d (level %d, ni=%d, nb=%d): hssrank(D)=%lld (%lld samples)\n", ids[i], x.level, x.ni, x.nb, (long long)hs_hss_rank(H),
              (long long)hs_hss_samples(H));
    if (!x.ht) dmalloc(&x.ht, ((size_t)x.ni + 32) * sizeof(T), "HSS solve vector");
  }
  // B. low-rank forms of the off-diagonal blocks of the fronts that have a boundary (one batch)
  std::vector<int> wb;
  for (int i = 0; i < count; ++i)
    if (hd[i].nb > 0) wb.push_back(i);
  const int nw = (int)wb.size();
  if (nw == 0) return;
  std::vector<LowRank<T>*> LL(nw), RR(nw);
  {
    std::vector<LowRankJob<T>> jobs(2 * nw);
    const int k0 = h->opts.kest > 0 ? (int)h->opts.kest : 128;
    for (int a = 0; a < nw; ++a) {
      const int i = wb[a];
      NodeH& x = h->nodes[ids[i]];
      x.lrL = LL[a] = new LowRank<T>();
      x.lrR = RR[a] = new LowRank<T>();
      const int kL = x.last_rL > 0 ? (x.last_rL + 8 + 31) / 32 * 32 + 32 : k0, kR = x.last_rR > 0 ? (x.last_rR + 8 + 31) / 32 * 32 + 32 : k0;
      const uint64_t seed = (uint64_t)h->opts.seed * 2654435761ull + (uint64_t)ids[i] * 2;
      jobs[2 * a] = LowRankJob<T>{hd[i].UR, hd[i].ldu, hd[i].ni, hd[i].nb, kR, seed, RR[a], 0};
      jobs[2 * a + 1] = LowRankJob<T>{hd[i].LF + hd[i].ni, hd[i].ldl, hd[i].nb, hd[i].ni, kL, seed + 1, LL[a], 0};
    }
    static const bool lr_qr = !(getenv("HS_LR_QR") && getenv("HS_LR_QR")[0] == '0');
    const int st = lr_qr ? lowrank_id_batch<T>(jobs.data(), 2 * nw, 0.5 * h->opts.atol, 0.5 * h->opts.rtol, s)
                         : lowrank_compress_batch<T>(jobs.data(), 2 * nw, 0.5 * h->opts.atol, 0.5 * h->opts.rtol, s);  // factorization.jl:99-100
    if (st != 0) throw HsError{st};
  }
  std::vector<T*> W2(nw, nullptr), W3(nw, nullptr);
  GemmProb<T>* dgp = nullptr;
  auto cleanup = [&]() {
    for (int a = 0; a < nw; ++a) {
      if (W2[a]) (void)hipFree(W2[a]);
      if (W3[a]) (void)hipFree(W3[a]);
    }
    if (dgp) (void)hipFree(dgp);
  };
  try {
    int maxnb = 0, mrL = 0, mrR = 0;
    std::vector<GemmProb<T>> gp(3 * nw);
    for (int a = 0; a < nw; ++a) {
      const int i = wb[a];
      NodeH& x = h->nodes[ids[i]];
      LowRank<T>*lrL = LL[a], *lrR = RR[a];
      x.last_rL = lrL->r;
      x.last_rR = lrR->r;
      h->maxrank = std::max<int64_t>(h->maxrank, std::max(lrL->r, lrR->r));
      if (h->opts.verbose || vt)
        fprintf(stderr, "[hs] node %d (level %d, ni=%d, nb=%d): rank(L)=%d rank(R)=%d\n", ids[i], x.level, x.ni, x.nb, lrL->r, lrR->r);
      lowrank_expand<T>(*lrR, s);
      lowrank_expand<T>(*lrL, s);
      if ((lrR->r > 0 && !lrR->Cd) || (lrL->r > 0 && !lrL->Cd)) HS_FAIL(HS_ERR_NOMEM, ids[i], "hipMalloc of a low-rank factor of node %d failed", ids[i]);
      const int rL = lrL->r, rR = lrR->r, ni = hd[i].ni, nb = hd[i].nb;
      gp[a] = gp[nw + a] = gp[2 * nw + a] = GemmProb<T>{nullptr, nullptr, nullptr, 0, 0, 0, 2, 2, 2};
      if (rR == 0) continue;
      // C'. W = Aii^-1 * C_R through the HSS elimination
      const size_t wel = (size_t)lrR->ldc * rR;
      dmalloc(&x.hW, (wel + 32) * sizeof(T), "Aii^-1*C_R");
      x.hldw = lrR->ldc;
      HS_HIP(hipMemcpyAsync(x.hW, lrR->Cd, wel * sizeof(T), hipMemcpyDeviceToDevice, s));
      const int st = hs_hss_ldiv((hs_hss*)x.hss, (double*)x.hW, x.hldw, rR, 1);
      if (st != 0) throw HsError{st};
      if (rL == 0) continue;
      const int ldw2 = (rL + 1) / 2 * 2;
      dmalloc((void**)&W2[a], ((size_t)ldw2 * rR + 32) * sizeof(T), "Z_L*W");
      dmalloc((void**)&W3[a], ((size_t)ldw2 * nb + 32) * sizeof(T), "(Z_L*W)*Z_R");
      gp[a] = GemmProb<T>{lrL->Z, (const T*)x.hW, W2[a], rL, rR, ni, lrL->ldz, x.hldw, ldw2};
      gp[nw + a] = GemmProb<T>{W2[a], lrR->Z, W3[a], rL, nb, rR, ldw2, lrR->ldz, ldw2};
      gp[2 * nw + a] = GemmProb<T>{lrL->Cd, W3[a], hd[i].SB, nb, nb, rL, lrL->ldc, ldw2, hd[i].lds};
      mrL = std::max(mrL, rL);
      mrR = std::max(mrR, rR);
      maxnb = std::max(maxnb, nb);
    }
    lap("B, C': IDs, W = D^-1 C_R", ids[wb[0]]);
    if (mrL > 0 && mrR > 0) {  // E. S -= C_L * ((Z_L * W) * Z_R)
      dmalloc((void**)&dgp, sizeof(GemmProb<T>) * gp.size(), "GEMM descriptors");
      HS_HIP(hipMemcpy(dgp, gp.data(), sizeof(GemmProb<T>) * gp.size(), hipMemcpyHostToDevice));
      launch_gemm_probs<T>(dgp, nw, mrL, mrR, 0, s);
      launch_gemm_probs<T>(dgp + nw, nw, mrL, maxnb, 0, s);
      launch_gemm_probs<T>(dgp + 2 * nw, nw, maxnb, maxnb, 1, s);
    }
    HS_HIP(hipStreamSynchronize(s));
    lap("E: Schur update", ids[wb[0]]);
    for (int a = 0; a < nw; ++a)
      for (LowRank<T>* lr : {LL[a], RR[a]}) {  // the dense factor replaces the trapezoid form
        if (!lr->Cd) continue;
        hs_lr_free(lr->Lp);
        hs_lr_free(lr->rperm);
        lr->Lp = nullptr;
        lr->rperm = nullptr;
      }
  } catch (...) {
    (void)hipStreamSynchronize(s);
    cleanup();
    throw;
  }
  cleanup();
}

// B (ni x q, device) <- D^-1 B for a front whose interior block is held in HSS form: one HSS solve, or the reference's block solve
// `blockldiv!` (src/blockmatrix.jl:134-144) over [A11 A12; A21 A22] with S22 = A22 - A21*A11^-1*A12:
//   y1 = A11^-1 B1;  x2 = S22^-1 (B2 - A21 y1);  x1 = y1 - (A11^-1 A12) x2          (A12 = C12*Z12, A21 = C21*Z21, W12 = A11^-1*C12)
template <class T>
static void hss_d_solve(hs_handle* h, const NodeH& x, T* B, int ldb, int q, hipStream_t s) {
  auto hsolve = [&](void* H, T* b) {
    int st = hs_hss_set_stream((hs_hss*)H, (void*)s);
    if (st == 0) st = hs_hss_ldiv((hs_hss*)H, (double*)b, ldb, q, 1);
    if (st != 0) throw HsError{st};
  };
  if (!x.mfb) {
    hsolve(x.hss, B);
    return;
  }
  const int n1 = x.ni1, n2 = x.ni - x.ni1, k12 = x.bk12, k21 = x.bk21;
  T *B1 = B, *B2 = B + n1;
  hsolve(x.hss, B1);
  if (q == 1) {
    if (k21 > 0) {
      ensure_lr_workspace<T>(h, k21, n1);
      launch_lr_zmul<T>((const T*)x.bZ21, x.bldz21, k21, n1, B1, nullptr, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
      launch_lr_dense<T>((const T*)x.bC21, x.bldc21, n2, k21, (const T*)h->d_lr_t, B2, nullptr, s);
    }
    hsolve(x.hss2, B2);
    if (k12 > 0) {
      ensure_lr_workspace<T>(h, k12, n2);
      launch_lr_zmul<T>((const T*)x.bZ12, x.bldz12, k12, n2, B2, nullptr, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
      launch_lr_dense<T>((const T*)x.bW12, x.bldw, n1, k12, (const T*)h->d_lr_t, B1, nullptr, s);
    }
    return;
  }
  // blocks of right-hand sides (W = D^-1 C_R during the factorization): the same steps as MFMA products
  const int kmax = std::max(std::max(k12, k21), 1), ldt = (kmax + 1) / 2 * 2;
  T* t = nullptr;
  GemmProb<T>* dgp = nullptr;
  if (hs_lr_alloc((void**)&t, ((size_t)ldt * q + 32) * sizeof(T)) != 0 || hs_lr_alloc((void**)&dgp, 2 * sizeof(GemmProb<T>)) != 0) {
    hs_lr_free(t);
    HS_FAIL(HS_ERR_NOMEM, 0, "hipMalloc of the block-solve scratch failed");
  }
  auto pair = [&](const T* Z, int ldz, int k, int cols, const T* X, const T* C, int ldc, int rows, T* Y) {  // Y -= C * (Z * X)
    if (k <= 0) return;
    GemmProb<T> gp[2] = {GemmProb<T>{Z, X, t, k, q, cols, ldz, ldb, ldt}, GemmProb<T>{C, t, Y, rows, q, k, ldc, ldt, ldb}};
    HS_HIP(hipMemcpyAsync(dgp, gp, sizeof gp, hipMemcpyHostToDevice, s));
    HS_HIP(hipStreamSynchronize(s));
    launch_gemm_probs<T>(dgp, 1, k, q, 0, s);
    launch_gemm_probs<T>(dgp + 1, 1, rows, q, 1, s);
  };
  try {
    pair((const T*)x.bZ21, x.bldz21, k21, n1, B1, (const T*)x.bC21, x.bldc21, n2, B2);
    hsolve(x.hss2, B2);
    pair((const T*)x.bZ12, x.bldz12, k12, n2, B2, (const T*)x.bW12, x.bldw, n1, B1);
    HS_HIP(hipStreamSynchronize(s));
  } catch (...) {
    (void)hipStreamSynchronize(s);
    hs_lr_free(t);
    hs_lr_free(dgp);
    throw;
  }
  hs_lr_free(t);
  hs_lr_free(dgp);
}

// forward sweep of level lv:  t = H^-1 rhs[int];  rhs[bnd] -= C_L * (Z_L * t)
template <class T>
static void solve_hss_fwd(hs_handle* h, int lv, T* db, hipStream_t s) {
  const LevelH& L = h->levels[lv];
  for (int id : L.mine) {
    const NodeH& x = h->nodes[id];
    if (!(x.hssd || x.mf) || !x.hss) continue;
    T* t = (T*)x.ht;
    launch_pack_idx(h->d_int + x.off_fidx, x.ni, db, t, (int)sizeof(T), s);
    hss_d_solve<T>(h, x, t, x.ni, 1, s);
    if (x.nb == 0 || !x.lrL) continue;
    const LowRank<T>& lr = *(const LowRank<T>*)x.lrL;
    if (lr.r == 0) continue;
    ensure_lr_workspace<T>(h, lr.r, lr.cols);
    launch_lr_zmul<T>(lr.Z, lr.ldz, lr.r, lr.cols, t, nullptr, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
    lr_apply_C<T>(lr, (const T*)h->d_lr_t, db, h->d_int + x.off_fidx + x.ni, s);
  }
}

// backward sweep of level lv:  rhs[int] = t - W * (Z_R * rhs[bnd])
template <class T>
static void solve_hss_bwd(hs_handle* h, int lv, T* db, hipStream_t s) {
  const LevelH& L = h->levels[lv];
  for (int id : L.mine) {
    const NodeH& x = h->nodes[id];
    if (!(x.hssd || x.mf) || !x.hss) continue;
    T* t = (T*)x.ht;
    if (x.nb > 0 && x.lrR && x.hW) {
      const LowRank<T>& lr = *(const LowRank<T>*)x.lrR;
      if (lr.r > 0) {
        ensure_lr_workspace<T>(h, lr.r, lr.cols);
        launch_lr_zmul<T>(lr.Z, lr.ldz, lr.r, lr.cols, db, h->d_int + x.off_fidx + x.ni, (T*)h->d_lr_part, (T*)h->d_lr_t, s);
        launch_lr_dense<T>((const T*)x.hW, x.hldw, x.ni, lr.r, (const T*)h->d_lr_t, t, nullptr, s);
      }
    }
    launch_unpack_idx(h->d_int + x.off_fidx, x.ni, db, t, (int)sizeof(T), s);
  }
}
