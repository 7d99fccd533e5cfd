// hs_hss_op.h -- operators that are never formed, for the HSS compression (included by hs_hss.hip inside its namespace).
//
// Reference: the compressed branch assembles its blocks from the children's HSS Schur complements WITHOUT densifying them
// (`_assemble_blocks`, src/factorization.jl:126-140):
//     Aii = [S1.A11  A[int1,int2]; A[int2,int1]  S2.A11]      Abb = [S1.A22  A[bnd1,bnd2]; A[bnd2,bnd1]  S2.A22]
// and hands `S = P (Abb - Abi*R) P'` to `randcompress_adaptive` as an operator with products (`_sample_schur!`, :238-244) and
// entries (`_getindex_schur`, :246-249).  `BlockOp` is that operator on the device: two diagonal HSS blocks (views of the
// children's HSS matrices, hs_hss_child) + the sparse couplings of `A` between their index sets, optionally minus a low-rank
// product C*M*Z (Lru, hs_hss.hip).  Products: two HSS products + a sparse product (CSR rows for A*X, CSC columns for A^T*X:
// both gather-form, deterministic); entries: batched HSS entry access (hss_getindex_batch) + a sparse entry gather.
#pragma once

// ---- sparse kernels -------------------------------------------------------------------------------------------------
// Y[i, c] += sum over the stored entries (g_i, q) of row g_i (CSR; or of column g_i of the CSC form for the transposed
// product) with q in the OTHER part of the operator: val * X[lpos[q], c].  lpos: global id -> operator index, -1 = outside.
template <class T>
__global__ __launch_bounds__(64) void blockop_spmm_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const T* __restrict__ val,
                                                          const int* __restrict__ gid, const int* __restrict__ lpos, int n, int n1, const T* __restrict__ X,
                                                          int ldx, T* __restrict__ Y, int ldy, int k) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, k);
  const int64_t g = gid[i];
  const bool first = i < n1;
  const int64_t e0 = ptr[g], e1 = ptr[g + 1];
  for (int c = c0; c < c1; ++c) {
    T acc = Scal<T>::zero();
    bool any = false;
    for (int64_t e = e0; e < e1; ++e) {
      const int q = lpos[idx[e]];
      if (q < 0 || (q < n1) == first) continue;
      acc = Scal<T>::fma(val[e], X[(size_t)q + (size_t)c * ldx], acc);
      any = true;
    }
    if (any) Y[(size_t)i + (size_t)c * ldy] = Y[(size_t)i + (size_t)c * ldy] + acc;
  }
}

// out[ro[a], co[c]] (or transposed) = A[gr[a], gc[c]] from the CSC form: one thread per entry scans the column
template <class T>
struct SpGetJob {
  const int* gr;  // global row ids (cnt rows)
  const int* gc;
  const int* ro;  // output positions of the rows / columns
  const int* co;
  int rows, cols;
  T* out;
  int ldo, trans;
};
template <class T>
__global__ __launch_bounds__(64) void blockop_spget_kernel(const SpGetJob<T>* __restrict__ jobs, const int64_t* __restrict__ colptr,
                                                           const int32_t* __restrict__ rowval, const T* __restrict__ nz) {
  const SpGetJob<T> j = jobs[blockIdx.z];
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= j.rows) return;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, j.cols);
  const int gr = j.gr[a], orow = j.ro[a];
  for (int c = c0; c < c1; ++c) {
    const int64_t g = j.gc[c];
    T v = Scal<T>::zero();
    for (int64_t e = colptr[g]; e < colptr[g + 1]; ++e)
      if (rowval[e] == gr) v = v + nz[e];
    const int ocol = j.co[c];
    if (j.trans)
      j.out[(size_t)ocol + (size_t)orow * j.ldo] = v;
    else
      j.out[(size_t)orow + (size_t)ocol * j.ldo] = v;
  }
}

// out[ro[a], co[c]] (or transposed) = src[a, c]
template <class T>
struct ScatJob {
  const T* src;
  int lds;
  const int* ro;
  const int* co;
  int rows, cols;
  T* out;
  int ldo, trans;
};
template <class T>
__global__ __launch_bounds__(64) void sub_scatter_kernel(const ScatJob<T>* __restrict__ jobs) {
  const ScatJob<T> j = jobs[blockIdx.z];
  const int a = blockIdx.x * 64 + threadIdx.x;
  if (a >= j.rows) return;
  const int c0 = blockIdx.y * 16, c1 = min(c0 + 16, j.cols);
  const int orow = j.ro[a];
  for (int c = c0; c < c1; ++c) {
    const T v = j.src[(size_t)a + (size_t)c * j.lds];
    const int ocol = j.co[c];
    if (j.trans)
      j.out[(size_t)ocol + (size_t)orow * j.ldo] = v;
    else
      j.out[(size_t)orow + (size_t)ocol * j.ldo] = v;
  }
}

__global__ __launch_bounds__(256) void lpos_set_kernel(const int* __restrict__ gid, int n, int* __restrict__ lpos, int clear) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) lpos[gid[i]] = clear ? -1 : i;
}

template <class J, class K>
void run_z_batched(Pool& tmp, std::vector<J>& jobs, K kernel, hipStream_t s) {
  std::vector<J> live;
  int mr = 0, mc = 0;
  for (auto& j : jobs)
    if (j.rows > 0 && j.cols > 0) {
      live.push_back(j);
      mr = std::max(mr, j.rows);
      mc = std::max(mc, j.cols);
    }
  jobs.clear();
  if (live.empty()) return;
  J* d = upload(tmp, live, s);
  for (size_t b = 0; b < live.size(); b += 32768) {
    const unsigned cnt = (unsigned)std::min<size_t>(32768, live.size() - b);
    kernel(dim3((mr + 63) / 64, (mc + 15) / 16, cnt), (const J*)(d + b));
  }
}

// ints that many small descriptors point to: filled on the host, uploaded once
struct IntArena {
  std::vector<int> h;
  int* d = nullptr;
  size_t cap = 0;
  void reserve(Pool& pool, size_t n) {
    cap = n + 16;
    d = pool.get<int>(cap);
    h.clear();
    h.reserve(cap);
  }
  // returns the DEVICE address the values will have after flush()
  int* put(const int* v, size_t n) {
    if (h.size() + n > cap) {
      hs_set_error(HS_ERR_NOMEM, 0, "internal: index arena of the HSS module too small");
      throw (int)HS_ERR_NOMEM;
    }
    int* at = d + h.size();
    h.insert(h.end(), v, v + n);
    return at;
  }
  int* put(const std::vector<int>& v) { return put(v.data(), v.size()); }
  void flush(Pool& pool, hipStream_t s) {  // asynchronous: staged through a pinned block owned by `pool`
    if (h.empty()) return;
    void* ph = pool.get_pinned(sizeof(int) * h.size());
    memcpy(ph, h.data(), sizeof(int) * h.size());
    HSS_HIP(hipMemcpyAsync(d, ph, sizeof(int) * h.size(), hipMemcpyHostToDevice, s));
  }
};

// ---- entries of an HSS matrix for MANY index pairs at once ---------------------------------------------------------------
// out_b = H[I_b, J_b] for every job b (index lists on the host, in the caller's index order of H).  Same algorithm as the
// single-pair access (basis rows of the requested indices carried up the tree as E on the row side and F = E^T on the
// column side, couplings E_l*B12*F_r per inner node), but every tree level is ONE group of launches over all jobs: the
// matrix-free compression asks for the blocks of a whole level of its cluster tree at a time.
template <class T>
struct GiJob {
  const int* I;
  int ni;
  const int* J;
  int nj;
  T* out;  // ni x nj, leading dimension ldo (device)
  int ldo;
};

template <class T>
void hss_getindex_batch(HssT<T>& H, const std::vector<GiJob<T>>& jobs) {
  hipStream_t s = H.s;
  auto& nd = H.nd;
  const int N = (int)nd.size();
  const int B = (int)jobs.size();
  if (B == 0) return;
  Pool tmp(global_cache());
  struct JS {
    std::vector<int> oI, oJ, sI, sJ;  // order and sorted tree positions
    T* outS = nullptr;
    int ldS = 2;
  };
  std::vector<JS> js(B);
  size_t nints = 0;
  for (int b = 0; b < B; ++b) {
    const GiJob<T>& g = jobs[b];
    if (g.ni <= 0 || g.nj <= 0) continue;
    auto prep = [&](const int* X, int cnt, std::vector<int>& srt, std::vector<int>& ord) {
      std::vector<int> pos(cnt);
      ord.resize(cnt);
      for (int a = 0; a < cnt; ++a) {
        if (X[a] < 0 || X[a] >= H.n) {
          hs_set_error(HS_ERR_ARGUMENT, a, "BoundsError: index %d outside 0:%d", X[a], H.n - 1);
          throw (int)HS_ERR_ARGUMENT;
        }
        pos[a] = H.hinvperm.empty() ? X[a] : H.hinvperm[(size_t)X[a]];
        ord[a] = a;
      }
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int c) { return pos[a] < pos[c]; });
      srt.resize(cnt);
      for (int a = 0; a < cnt; ++a) srt[a] = pos[ord[a]];
    };
    prep(g.I, g.ni, js[b].sI, js[b].oI);
    prep(g.J, g.nj, js[b].sJ, js[b].oJ);
    js[b].ldS = ev(g.ni);
    js[b].outS = tmp.getz<T>((size_t)js[b].ldS * g.nj, s);
    nints += (size_t)4 * (g.ni + g.nj);
  }
  IntArena ia;
  ia.reserve(tmp, nints);
  auto range = [](const std::vector<int>& v, int lo, int hi, int& b0, int& e0) {
    b0 = (int)(std::lower_bound(v.begin(), v.end(), lo) - v.begin());
    e0 = (int)(std::lower_bound(v.begin(), v.end(), hi) - v.begin());
  };
  // per (job, node): ranges of the sorted lists and the carried basis rows
  std::vector<int> bI((size_t)B * N), eI((size_t)B * N), bJ((size_t)B * N), eJ((size_t)B * N);
  std::vector<T*> E((size_t)B * N, nullptr), F((size_t)B * N, nullptr);
  auto at = [&](int b, int i) { return (size_t)b * N + i; };
  for (int b = 0; b < B; ++b) {
    if (!js[b].outS) continue;
    for (int i = 0; i < N; ++i) {
      range(js[b].sI, nd[i].lo, nd[i].hi, bI[at(b, i)], eI[at(b, i)]);
      range(js[b].sJ, nd[i].lo, nd[i].hi, bJ[at(b, i)], eJ[at(b, i)]);
    }
  }
  std::vector<SubJob<T>> subs;
  std::vector<BasisJob<T>> bjobs;
  int maxcnt = 0, maxr = 0;
  // ---- leaves: diagonal blocks and basis rows ----------------------------------------------------------------------
  for (int i = 0; i < N; ++i) {
    HNode<T>& x = nd[i];
    if (x.left >= 0) continue;
    bool touched = false;
    for (int b = 0; b < B && !touched; ++b) touched = js[b].outS && (eI[at(b, i)] > bI[at(b, i)] || eJ[at(b, i)] > bJ[at(b, i)]);
    if (!touched) continue;
    if (i != 0 && x.hinvp.empty()) {
      std::vector<int> hp(x.m);
      HSS_HIP(hipMemcpyAsync(hp.data(), x.p, sizeof(int) * x.m, hipMemcpyDeviceToHost, s));
      HSS_HIP(hipStreamSynchronize(s));
      x.hinvp.assign(x.m, 0);
      for (int a = 0; a < x.m; ++a) x.hinvp[hp[a]] = a;
    }
    for (int b = 0; b < B; ++b) {
      if (!js[b].outS) continue;
      const int cI = eI[at(b, i)] - bI[at(b, i)], cJ = eJ[at(b, i)] - bJ[at(b, i)];
      if (cI == 0 && cJ == 0) continue;
      std::vector<int> li(cI), lj(cJ);
      for (int a = 0; a < cI; ++a) li[a] = js[b].sI[bI[at(b, i)] + a] - x.lo;
      for (int a = 0; a < cJ; ++a) lj[a] = js[b].sJ[bJ[at(b, i)] + a] - x.lo;
      if (cI > 0 && cJ > 0) {
        int* dli = ia.put(li);
        int* dlj = ia.put(lj);
        subs.push_back(SubJob<T>{x.D, x.ldd, dli, dlj, 0, 0, cI, cJ, js[b].outS + bI[at(b, i)] + (size_t)js[b].ldS * bJ[at(b, i)], js[b].ldS, 0});
      }
      if (i == 0) continue;
      for (int side = 0; side < 2; ++side) {
        const std::vector<int>& l = side == 0 ? li : lj;
        const int cnt = (int)l.size();
        if (cnt == 0) continue;
        std::vector<int> ip(cnt);
        for (int a = 0; a < cnt; ++a) ip[a] = x.hinvp[l[a]];
        int* dip = ia.put(ip);
        T* o = side == 0 ? (E[at(b, i)] = tmp.get<T>((size_t)ev(cnt) * x.r)) : (F[at(b, i)] = tmp.get<T>((size_t)ev(x.r) * cnt));
        bjobs.push_back(BasisJob<T>{x.Tm, x.ldt, x.r, cnt, dip, o, side == 0 ? ev(cnt) : ev(x.r), side});
        maxcnt = std::max(maxcnt, cnt);
        maxr = std::max(maxr, x.r);
      }
    }
  }
  // back-to-caller-order lists (used at the very end) go into the same arena
  std::vector<int*> drI(B, nullptr), drJ(B, nullptr);
  for (int b = 0; b < B; ++b) {
    if (!js[b].outS) continue;
    std::vector<int> rI(jobs[b].ni), rJ(jobs[b].nj);
    for (int a = 0; a < jobs[b].ni; ++a) rI[js[b].oI[a]] = a;
    for (int a = 0; a < jobs[b].nj; ++a) rJ[js[b].oJ[a]] = a;
    drI[b] = ia.put(rI);
    drJ[b] = ia.put(rJ);
  }
  ia.flush(tmp, s);
  run_subs(tmp, subs, s);
  if (!bjobs.empty()) {
    BasisJob<T>* dj = upload(tmp, bjobs, s);
    for (size_t b0 = 0; b0 < bjobs.size(); b0 += 32768) {
      const unsigned cnt = (unsigned)std::min<size_t>(32768, bjobs.size() - b0);
      hipLaunchKernelGGL(basis_rows_kernel<T>, dim3((maxcnt + 63) / 64, (maxr + 15) / 16, cnt), dim3(64), 0, s, (const BasisJob<T>*)(dj + b0));
    }
  }
  // ---- inner nodes, deepest level first ----------------------------------------------------------------------------
  for (int lv = H.nlev - 2; lv >= 0; --lv) {
    std::vector<GemmProb<T>> ga, gb, ge, gf;
    std::vector<RowJob<T>> negs, rj;
    std::vector<SubJob<T>> blocks, cg;
    for (int i : H.lev[lv]) {
      HNode<T>& x = nd[i];
      if (x.left < 0) continue;
      const int l = x.left, r = x.right, rl = nd[l].r, rr = nd[r].r;
      const int m = x.m, rk = x.r, nR = m - rk;
      for (int b = 0; b < B; ++b) {
        if (!js[b].outS) continue;
        const int cIl = eI[at(b, l)] - bI[at(b, l)], cIr = eI[at(b, r)] - bI[at(b, r)], cJl = eJ[at(b, l)] - bJ[at(b, l)], cJr = eJ[at(b, r)] - bJ[at(b, r)];
        if (cIl + cIr + cJl + cJr == 0) continue;
        T* outS = js[b].outS;
        const int ldS = js[b].ldS;
        // out[I_l, J_r] = E_l * B12 * F_r,  out[I_r, J_l] = E_r * B21 * F_l
        if (cIl > 0 && cJr > 0 && rl > 0 && rr > 0) {
          T* t = tmp.getz<T>((size_t)ev(cIl) * rr, s);
          ga.push_back(GemmProb<T>{E[at(b, l)], x.B12, t, cIl, rr, rl, ev(cIl), x.ld12, ev(cIl)});
          gb.push_back(GemmProb<T>{t, F[at(b, r)], outS + bI[at(b, l)] + (size_t)ldS * bJ[at(b, r)], cIl, cJr, rr, ev(cIl), ev(rr), ldS});
        }
        if (cIr > 0 && cJl > 0 && rl > 0 && rr > 0) {
          T* t = tmp.getz<T>((size_t)ev(cIr) * rl, s);
          ga.push_back(GemmProb<T>{E[at(b, r)], x.B21, t, cIr, rl, rr, ev(cIr), x.ld21, ev(cIr)});
          gb.push_back(GemmProb<T>{t, F[at(b, l)], outS + bI[at(b, r)] + (size_t)ldS * bJ[at(b, l)], cIr, cJl, rl, ev(cIr), ev(rl), ldS});
        }
        if (i == 0) continue;
        // the node's own rows: W = [E_l 0; 0 E_r] -> E = W[:, p_S] + W[:, p_R] * T;  Wt = [F_l 0; 0 F_r] -> F = Wt[p_S, :] + T^T * Wt[p_R, :]
        if (!x.NTm && nR > 0) {
          x.NTm = H.keep.template get<T>((size_t)x.ldt * rk);
          negs.push_back(RowJob<T>{x.Tm, x.ldt, x.NTm, x.ldt, nullptr, nR, rk, ROW_GATHER_NEG});
        }
        const int cI = cIl + cIr, cJ = cJl + cJr;
        if (cI > 0) {
          const int ldw = ev(cI);
          T* W = tmp.getz<T>((size_t)ldw * m, s);
          if (cIl > 0) blocks.push_back(SubJob<T>{E[at(b, l)], ev(cIl), nullptr, nullptr, 0, 0, cIl, rl, W, ldw, 0});
          if (cIr > 0) blocks.push_back(SubJob<T>{E[at(b, r)], ev(cIr), nullptr, nullptr, 0, 0, cIr, rr, W + cIl + (size_t)ldw * rl, ldw, 0});
          T* Ei = E[at(b, i)] = tmp.get<T>((size_t)ldw * rk);
          cg.push_back(SubJob<T>{W, ldw, nullptr, x.p, 0, 0, cI, rk, Ei, ldw, 0});
          if (nR > 0) {
            T* Wr = tmp.get<T>((size_t)ldw * nR);
            cg.push_back(SubJob<T>{W, ldw, nullptr, x.p + rk, 0, 0, cI, nR, Wr, ldw, 0});
            ge.push_back(GemmProb<T>{Wr, x.NTm, Ei, cI, rk, nR, ldw, x.ldt, ldw});  // E -= Wr * (-T)
          }
        }
        if (cJ > 0) {
          const int ldw = ev(m);
          T* Wt = tmp.getz<T>((size_t)ldw * cJ, s);
          if (cJl > 0) blocks.push_back(SubJob<T>{F[at(b, l)], ev(rl), nullptr, nullptr, 0, 0, rl, cJl, Wt, ldw, 0});
          if (cJr > 0) blocks.push_back(SubJob<T>{F[at(b, r)], ev(rr), nullptr, nullptr, 0, 0, rr, cJr, Wt + rl + (size_t)ldw * cJl, ldw, 0});
          T* Fi = F[at(b, i)] = tmp.get<T>((size_t)ev(rk) * cJ);
          rj.push_back(RowJob<T>{Wt, ldw, Fi, ev(rk), x.p, rk, cJ, ROW_GATHER});
          if (nR > 0) {
            T* t = tmp.get<T>((size_t)ev(nR) * cJ);
            rj.push_back(RowJob<T>{Wt, ldw, t, ev(nR), x.p + rk, nR, cJ, ROW_GATHER_NEG});
            gf.push_back(GemmProb<T>{x.Tt, t, Fi, rk, cJ, nR, x.ldtt, ev(nR), ev(rk)});  // F += T^T * Wt[p_R, :]
          }
        }
      }
    }
    run_rows(tmp, negs, s);  // -T copies
    run_gemms(tmp, ga, 0, s);
    run_gemms(tmp, gb, 0, s);
    run_subs(tmp, blocks, s);  // W, Wt blocks
    run_subs(tmp, cg, s);      // column gathers W[:, p_S] -> E, W[:, p_R] -> Wr
    run_gemms(tmp, ge, 1, s);
    run_rows(tmp, rj, s);
    run_gemms(tmp, gf, 1, s);
  }
  // back to the callers' orders
  for (int b = 0; b < B; ++b)
    if (js[b].outS) subs.push_back(SubJob<T>{js[b].outS, js[b].ldS, drI[b], drJ[b], 0, 0, jobs[b].ni, jobs[b].nj, jobs[b].out, jobs[b].ldo, 0});
  run_subs(tmp, subs, s);
  HSS_HIP(hipStreamSynchronize(s));
}

// ---- the operator ---------------------------------------------------------------------------------------------------
template <class T>
struct SparseDev {
  int64_t n = 0;
  const int64_t* colptr = nullptr;  // CSC, 0-based
  const int32_t* rowval = nullptr;
  const T* nz = nullptr;
  const int64_t* rowptr = nullptr;  // CSR of the same matrix
  const int32_t* colind = nullptr;
  const T* nzr = nullptr;
};

template <class T>
struct BlockOp {
  int n = 0, n1 = 0;            // index space [part 1 (n1) ; part 2 (n - n1)]
  HssT<T>* H1 = nullptr;        // diagonal blocks (null when the part is empty); their caller index order is the part's order
  HssT<T>* H2 = nullptr;
  std::vector<int> hgid;        // host: 0-based global DOF id of every index
  int* gid = nullptr;           // device copy
  int* lpos = nullptr;          // device, A.n entries, -1 outside the operator (set by begin(), restored by end())
  SparseDev<T> A;
  Pool own{global_cache()};

  void begin(hipStream_t s) {
    gid = own.get<int>((size_t)std::max(n, 1));
    HSS_HIP(hipMemcpyAsync(gid, hgid.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
    HSS_HIP(hipStreamSynchronize(s));
    if (n > 0) hipLaunchKernelGGL(lpos_set_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const int*)gid, n, lpos, 0);
  }
  void end(hipStream_t s) {
    if (n > 0 && gid) hipLaunchKernelGGL(lpos_set_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const int*)gid, n, lpos, 1);
    HSS_HIP(hipStreamSynchronize(s));
  }
  // Y = Op * X (trans: Op^T * X), n x k blocks in the operator's index order
  void mul(const T* X, int ldx, T* Y, int ldy, int k, bool trans, hipStream_t s) {
    const int n2 = n - n1;
    if (H1 && n1 > 0) {
      H1->s = s;
      hss_mul<T>(*H1, X, ldx, Y, ldy, k, trans);
    }
    if (H2 && n2 > 0) {
      H2->s = s;
      hss_mul<T>(*H2, X + n1, ldx, Y + n1, ldy, k, trans);
    }
    if (n1 > 0 && n2 > 0) {
      const int64_t* ptr = trans ? A.colptr : A.rowptr;
      const int32_t* idx = trans ? A.rowval : A.colind;
      const T* val = trans ? A.nz : A.nzr;
      hipLaunchKernelGGL(blockop_spmm_kernel<T>, dim3((n + 63) / 64, (k + 15) / 16), dim3(64), 0, s, ptr, idx, val, (const int*)gid, (const int*)lpos, n, n1, X,
                         ldx, Y, ldy, k);
    }
  }
  // the blocks Op[I, J] of a list of jobs (host index lists hri / hci in operator indices; null: r0 + i)
  void gather(Pool& tmp, const std::vector<SubJob<T>>& jobs, hipStream_t s) {
    std::vector<GiJob<T>> g1, g2;
    std::vector<ScatJob<T>> scat;
    std::vector<SpGetJob<T>> sp;
    std::vector<std::vector<int>> keep;  // host lists the batched entry access reads until it returns
    keep.reserve(jobs.size() * 8 + 8);
    size_t nints = 0;
    for (const SubJob<T>& j : jobs) nints += (size_t)4 * (std::max(j.rows, 0) + std::max(j.cols, 0));
    IntArena ia;
    ia.reserve(tmp, nints);
    for (const SubJob<T>& j : jobs) {
      if (j.rows <= 0 || j.cols <= 0) continue;
      // split the index lists by part; remember where every entry sits in the job's block
      std::vector<int> I1, I2, J1, J2, pI1, pI2, pJ1, pJ2, gI1, gI2, gJ1, gJ2;
      for (int a = 0; a < j.rows; ++a) {
        const int v = j.hri ? j.hri[a] : j.r0 + a;
        if (v < n1) { I1.push_back(v); pI1.push_back(a); gI1.push_back(hgid[v]); }
        else { I2.push_back(v - n1); pI2.push_back(a); gI2.push_back(hgid[v]); }
      }
      for (int c = 0; c < j.cols; ++c) {
        const int v = j.hci ? j.hci[c] : j.c0 + c;
        if (v < n1) { J1.push_back(v); pJ1.push_back(c); gJ1.push_back(hgid[v]); }
        else { J2.push_back(v - n1); pJ2.push_back(c); gJ2.push_back(hgid[v]); }
      }
      int* dI1 = ia.put(pI1); int* dI2 = ia.put(pI2); int* dJ1 = ia.put(pJ1); int* dJ2 = ia.put(pJ2);
      auto diag = [&](std::vector<GiJob<T>>& gl, std::vector<int>& I, std::vector<int>& J, int* dpI, int* dpJ) {
        if (I.empty() || J.empty()) return;
        const int ci = (int)I.size(), cj = (int)J.size(), ld = ev(ci);
        T* t = tmp.get<T>((size_t)ld * cj);
        keep.push_back(std::move(I));
        const int* hi = keep.back().data();
        keep.push_back(std::move(J));
        const int* hj = keep.back().data();
        gl.push_back(GiJob<T>{hi, ci, hj, cj, t, ld});
        scat.push_back(ScatJob<T>{t, ld, dpI, dpJ, ci, cj, j.out, j.ldo, j.trans});
      };
      const int c11 = (int)I1.size(), c12 = (int)J2.size(), c21 = (int)I2.size(), c22 = (int)J1.size();
      if (c11 > 0 && c12 > 0) sp.push_back(SpGetJob<T>{ia.put(gI1), ia.put(gJ2), dI1, dJ2, c11, c12, j.out, j.ldo, j.trans});
      if (c21 > 0 && c22 > 0) sp.push_back(SpGetJob<T>{ia.put(gI2), ia.put(gJ1), dI2, dJ1, c21, c22, j.out, j.ldo, j.trans});
      diag(g1, I1, J1, dI1, dJ1);
      diag(g2, I2, J2, dI2, dJ2);
    }
    ia.flush(tmp, s);
    if (H1 && !g1.empty()) {
      H1->s = s;
      hss_getindex_batch<T>(*H1, g1);
    }
    if (H2 && !g2.empty()) {
      H2->s = s;
      hss_getindex_batch<T>(*H2, g2);
    }
    const SparseDev<T> Ad = A;
    run_z_batched(tmp, sp, [&](dim3 grid, const SpGetJob<T>* d) {
      hipLaunchKernelGGL(blockop_spget_kernel<T>, grid, dim3(64), 0, s, d, Ad.colptr, Ad.rowval, Ad.nz);
    }, s);
    run_z_batched(tmp, scat, [&](dim3 grid, const ScatJob<T>* d) { hipLaunchKernelGGL(sub_scatter_kernel<T>, grid, dim3(64), 0, s, d); }, s);
    HSS_HIP(hipStreamSynchronize(s));  // `keep` and the descriptors die with this call
  }
};
