// hs_mffront.h -- the MATRIX-FREE compressed branch: Schur complements travel between fronts as HSS matrices (hs_options.mf).
//
// Reference: `_factor_branch(..., Val(true))` (src/factorization.jl:78-112) with HSS children:
//   C3  `_assemble_blocks` for HssMatrix children (:126-140): Aii = [S1.A11 A[int1,int2]; A[int2,int1] S2.A11], Aib / Abi = the
//       children's off-diagonal generators (`U*B12`, `V`) + sparse couplings, Abb = [S1.A22 A[bnd1,bnd2]; A[bnd2,bnd1] S2.A22] --
//       nothing is densified;
//   B2' `D = blockfactor(Aii)` over HSS blocks (src/blockmatrix.jl:121-130), every later `Aii^-1 X` through HSS solves (:134-156);
//   C5  Gauss transforms from the children's generators, the sparse coupling `X = [0 A12; A21 0]` factored on its own and
//       appended (:184-209); nothing is recompressed (`_recompress!` is commented out, :194,207);
//   C6  `S = P (Abb - Abi*R) P'` handed to `randcompress_adaptive` as an operator with products (`_sample_schur!`, :238-244) and
//       entries (`_getindex_schur`, :246-249), over `bisection_cluster((|int_loc|, |bnd|))` (:109-110);
//   F2  a flagged leaf: `compress(S[perm,perm], cl, cl)` (:45-59);
// and the TRANSITION (a flagged branch whose children are dense): dense blocks, `pqrfact` Gauss transforms, `S` compressed from the
// same operator (:99-110) -- hs_compress.h eliminates such a front, then `S` leaves through hs_hss_compress_lru instead of being formed.
//
// Formulation of `D`: ONE HSS matrix over a recursive-bisection order of the front's interior graph, compressed matrix-free from the
// operator [S1.A11 A12; A21 S2.A11] (hs_hss_blockop: two HSS products + a sparse product per sample, batched HSS entry access + a sparse
// gather per block) -- the `dmode="single"` formulation of the CPU restatement used by the tests.  The reference's 2x2 `BlockFactorization` needs `A11 \ A12` in HSS-by-HSS
// arithmetic (HssMatrices.jl, absent from the reference tree); `_equilibrate_clusters` (C2, :143-168) exists to make the two children's
// cluster trees compatible for that arithmetic and has no role here: every compression samples an operator.  `L = Abi*Aii^-1` is applied
// as `Abi_lr * (D^-1 x)` in `ldiv!` (one HSS solve serves `_dsolve!` and `_lsolve!`), so no transposed HSS solve is needed.
// PARITY UNPINNED (HssMatrices.jl / LowRankApprox.jl are not part of the reference tree); checked against a CPU restatement of the same data flow (tests/test_mf_gpu.py).
#pragma once
#include <atomic>
#include <mutex>
#include <thread>
#include "../../include/hs_hss.h"
#include "hs_lowrank.h"

// MfCoupling = NodeH::Coupling (hs_api.hip): entries of a sparse coupling block in block coordinates, e = 0-based position in nzval

static int mf_hss_leaf(const hs_options& o) {
  static const int env = getenv("HS_HSS_LEAF") ? atoi(getenv("HS_HSS_LEAF")) : 0;
  // SolverOptions.leafsize is honoured from 64 up; below that (the reference's default is 32) the leaves are 512 wide: measured on the
  // 32,768 root of Poisson 128^3 at 1e-6, leaves of 128 / 256 indices have full rank (the bottom two levels of the tree compress nothing
  // and cost a third of the time: 12.8 s with 128, 9.9 s with 256, 8.6 s with 512)
  return env > 0 ? std::max(32, env) : (o.leafsize >= 64 ? (int)o.leafsize : 512);
}

// ---- analysis: which Schur complements leave as HSS, which fronts are matrix-free, sparse coupling lists ---------------------------------
static void mf_plan(hs_handle* h, int64_t swlevel) {
  std::vector<NodeH>& N = h->nodes;
  const int leaf = mf_hss_leaf(h->opts);
  auto wants = [&](const NodeH& x) {  // compression_flag of factorization.jl:15 (a flagged LEAF compresses its S too, :45-59)
    return x.level >= 2 && x.level <= swlevel && x.nb >= h->opts.swsize && x.nb > leaf && x.ni > 0 && x.parent >= 0 && N[x.parent].level >= 1;
  };
  for (int i = 0; i < h->nreal; ++i) {
    NodeH& x = N[i];
    if (x.leaf || x.right < 0) continue;
    if (wants(N[x.left]) && wants(N[x.right])) {  // both or none: a parent assembles matrix-free from two HSS children
      N[x.left].s_hss = N[x.right].s_hss = true;
      x.mf = true;
    }
  }
}

// the order [int_loc; bnd_loc] of a node's boundary (factorization.jl:41,56-57,108) and |int_loc|, from its cmap
static void mf_sperm(hs_handle* h, const std::vector<int>& hint) {
  for (int i = 0; i < h->nreal; ++i) {
    NodeH& x = h->nodes[i];
    if (!x.s_hss || !(x.mine || x.ghost)) continue;
    const int* cm = hint.data() + x.off_cmap;
    const NodeH& p = h->nodes[x.parent];
    x.sperm.resize((size_t)x.nb);
    for (int e = 0; e < x.nb; ++e) x.sperm[(size_t)e] = e;
    std::stable_sort(x.sperm.begin(), x.sperm.end(), [&](int64_t a, int64_t b) { return cm[a] < cm[b]; });
    x.n1p = 0;
    for (int e = 0; e < x.nb; ++e) x.n1p += cm[e] >= 0 && cm[e] < p.ni;
  }
}

// sparse couplings A[int, bnd] / A[bnd, int] between the two children's parts of a matrix-free front (1-based CSC pattern of A)
static void mf_couplings(hs_handle* h, int64_t n, const int64_t* colptr, const int64_t* rowval) {
  std::vector<int> where((size_t)n, -1);
  for (int i = 0; i < h->nreal; ++i) {
    NodeH& x = h->nodes[i];
    if (!x.mf || !x.mine) continue;
    const int* F = h->fidx_host.data() + x.off_fidx;
    for (int p = 0; p < x.m; ++p) where[F[p]] = p;
    auto part = [&](int p) { return p < x.ni ? (p < x.ni1 ? 1 : 2) : ((p - x.ni) < x.nb1 ? 1 : 2); };
    for (int c = 0; c < x.m; ++c) {
      const int64_t g = F[c];
      for (int64_t e = colptr[g] - 1; e < colptr[g + 1] - 1; ++e) {
        const int p = where[rowval[e] - 1];
        if (p < 0 || part(p) == part(c)) continue;
        if (p < x.ni && c >= x.ni) {
          x.xr.row.push_back(p); x.xr.col.push_back(c - x.ni); x.xr.e.push_back(e);
        } else if (p >= x.ni && c < x.ni) {
          x.xl.row.push_back(p - x.ni); x.xl.col.push_back(c); x.xl.e.push_back(e);
        } else if ((x.mfb || x.mfd) && p < x.ni && c < x.ni) {  // A[int1, int2] / A[int2, int1]: the 2x2 block form of D, the dense expansion
          if (p < x.ni1) {
            x.x12.row.push_back(p); x.x12.col.push_back(c - x.ni1); x.x12.e.push_back(e);
          } else {
            x.x21.row.push_back(p - x.ni1); x.x21.col.push_back(c); x.x21.e.push_back(e);
          }
        }
      }
    }
    for (int p = 0; p < x.m; ++p) where[F[p]] = -1;
  }
}

// CSR form of A (pattern on the host -> device; the values follow by a fixed permutation of the CSC values in hs_numeric_begin)
static void mf_build_csr(hs_handle* h, int64_t n, const int64_t* colptr, const int64_t* rowval) {
  const int64_t nnz = colptr[n] - 1;
  std::vector<int64_t> rp((size_t)n + 1, 0), tp((size_t)nnz);
  std::vector<int32_t> ci((size_t)nnz);
  for (int64_t e = 0; e < nnz; ++e) rp[(size_t)rowval[e]]++;  // rowval is 1-based: counts land at index r+1
  for (int64_t r = 0; r < n; ++r) rp[(size_t)r + 1] += rp[(size_t)r];
  std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
  for (int64_t c = 0; c < n; ++c)
    for (int64_t e = colptr[c] - 1; e < colptr[c + 1] - 1; ++e) {
      const int64_t r = rowval[e] - 1, at = fill[(size_t)r]++;
      ci[(size_t)at] = (int32_t)c;
      tp[(size_t)at] = e;
    }
  dmalloc((void**)&h->d_rowptr, sizeof(int64_t) * ((size_t)n + 1), "rowptr");
  dmalloc((void**)&h->d_colind, sizeof(int32_t) * (size_t)nnz, "colind");
  dmalloc((void**)&h->d_tperm, sizeof(int64_t) * (size_t)nnz, "CSR value permutation");
  dmalloc(&h->d_nzr, (size_t)nnz * (h->is_complex ? 16 : 8), "CSR values");
  dmalloc((void**)&h->d_lpos, sizeof(int) * (size_t)n, "operator index map");
  HS_HIP(hipMemcpy(h->d_rowptr, rp.data(), sizeof(int64_t) * ((size_t)n + 1), hipMemcpyHostToDevice));
  HS_HIP(hipMemcpy(h->d_colind, ci.data(), sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
  HS_HIP(hipMemcpy(h->d_tperm, tp.data(), sizeof(int64_t) * (size_t)nnz, hipMemcpyHostToDevice));
  HS_HIP(hipMemset(h->d_lpos, 0xff, sizeof(int) * (size_t)n));
}

template <class T>
static void free_mf_nodes(hs_handle* h) {
  for (auto& x : h->nodes) {
    if (x.S_hss) {
      hs_hss_free((hs_hss*)x.S_hss);
      x.S_hss = nullptr;
    }
  }
}

// a matrix-free front with a dense interior block (NodeH::mfd): D with Z_L*U^-1 below it, L^-1*P*C_R and the small Schur block are
// allocated per factorization; its low-rank objects borrow the two factors that live inside those buffers
static void free_mfd_buffers(hs_handle* h) {
  for (auto& x : h->nodes) {
    if (!x.mfd_LF && !x.mfd_UR && !x.mfd_SB) continue;
    if (x.lrL) {
      if (h->is_complex) ((LowRank<cplx>*)x.lrL)->Z = nullptr; else ((LowRank<double>*)x.lrL)->Z = nullptr;
    }
    if (x.lrR) {
      if (h->is_complex) ((LowRank<cplx>*)x.lrR)->Cd = nullptr; else ((LowRank<double>*)x.lrR)->Cd = nullptr;
    }
    if (x.mfd_LF) (void)hipFree(x.mfd_LF);
    if (x.mfd_UR) (void)hipFree(x.mfd_UR);
    if (x.mfd_SB) (void)hipFree(x.mfd_SB);
    x.mfd_LF = x.mfd_UR = x.mfd_SB = nullptr;
  }
}

// rank_L: rank(L) of the front whose Schur complement is being compressed, or 0.  SolverOptions.kest < 0 (the reference's default, -1) means
// "start the adaptive compression of S with ceil(rank(L) / 2) samples" (src/factorization.jl:102-104) -- here: with at least that many; a front
// without a low-rank L (a flagged leaf, the blocks of D) starts from 4 sqrt(n).  A re-factorization of the same handle starts from the count the last one ended with.
static hs_hss_options mf_options(const hs_handle* h, int node, double scale, int64_t first_split, int last_k, int n, int rank_L = 0) {
  hs_hss_options o;
  hs_hss_options_default(&o);
  o.leafsize = mf_hss_leaf(h->opts);
  o.first_split = first_split;
  o.atol = h->opts.atol * scale;
  o.rtol = h->opts.rtol * scale;
  int64_t k0 = 128;
  while (k0 < 4.0 * std::sqrt((double)n)) k0 *= 2;
  // factorization.jl:102-104 -- but never fewer than the 4 sqrt(n) above: the sufficiency rule of the compression (a rank is trusted up to 0.8 k - pad)
  // needs the oversampling; measured with ceil(rank(L)/2) alone on Poisson 32^3 at 1e-2: 64 samples, error 7.0e-2 instead of 3.9e-2
  if (h->opts.kest < 0 && rank_L > 0) k0 = std::max<int64_t>(k0, ((rank_L + 1) / 2 + 31) / 32 * 32);
  o.kest = last_k > 0 ? last_k : (h->opts.kest > 0 ? h->opts.kest : k0);
  o.seed = h->opts.seed + 31 * (int64_t)node;
  return o;
}

static void mf_check(int st) {
  if (st != 0) throw HsError{st};
}

// The compressions of the fronts of one level are independent chains of small dependent launches (tree levels x windows of rows): run
// alone, each leaves most of the chip idle.  Up to HS_MF_THREADS (default 2: the chains are bound by the host side of the launches, which does not scale over threads) of them run concurrently, one host thread and one HIP stream
// each (the HSS module is synchronous per call; its block caches are mutex-guarded, its error state thread-local).  body(i, stream) may
// throw HsError / int; the first failure is re-raised on the calling thread with its message.
static std::mutex g_mf_mu;  // guards hs_handle::maxrank and verbose output ordering
template <class F>
static void mf_parallel(hs_handle* h, int count, F&& body) {
  static const int nth_env = getenv("HS_MF_THREADS") ? atoi(getenv("HS_MF_THREADS")) : 2;  // measured at Poisson 128^3, mf = 1: 4.32 / 3.88 / 3.98 / 4.03 / 4.51 s with 1 / 2 / 3 / 4 / 8 threads
  const int nth = std::max(1, std::min(count, nth_env));
  HS_HIP(hipStreamSynchronize(h->stream));  // everything the fronts read has been produced
  if (nth == 1) {
    for (int i = 0; i < count; ++i) body(i, h->stream);
    return;
  }
  int dev = 0;
  HS_HIP(hipGetDevice(&dev));
  std::atomic<int> next{0};
  std::vector<int> codes(nth, 0);
  std::vector<std::string> msgs(nth);
  std::vector<long long> infos(nth, 0);
  std::vector<std::thread> th;
  for (int t = 0; t < nth; ++t)
    th.emplace_back([&, t]() {
      hipStream_t st = nullptr;
      if (hipSetDevice(dev) != hipSuccess || hipStreamCreate(&st) != hipSuccess) {
        codes[t] = HS_ERR_DEVICE;
        msgs[t] = "hipSetDevice / hipStreamCreate failed in a worker of the matrix-free level";
        return;
      }
      try {
        for (int i = next.fetch_add(1); i < count; i = next.fetch_add(1)) {
          body(i, st);
          if (hipStreamSynchronize(st) != hipSuccess) throw HsError{HS_ERR_DEVICE};
        }
      } catch (const HsError& e) {
        codes[t] = e.code;
      } catch (int c) {
        codes[t] = c;
      } catch (const std::bad_alloc&) {
        codes[t] = HS_ERR_NOMEM;
      }
      if (codes[t] != 0) {
        msgs[t] = hs_last_error();
        infos[t] = hs_last_error_info();
        next.store(count);  // the others stop at their next front
      }
      (void)hipStreamSynchronize(st);
      (void)hipStreamDestroy(st);
    });
  for (auto& t : th) t.join();
  for (int t = 0; t < nth; ++t)
    if (codes[t] != 0) {
      hs_set_error(codes[t], infos[t], "%s", msgs[t].c_str());
      throw HsError{codes[t]};
    }
}
static void mf_maxrank(hs_handle* h, int64_t r) {
  std::lock_guard<std::mutex> lk(g_mf_mu);
  h->maxrank = std::max<int64_t>(h->maxrank, r);
}

// S of a front that was eliminated on its dense front (a transition branch or a flagged leaf) leaves as an HSS matrix:
// H ~= (SB - C*M*Z)[perm, perm], perm = [int_loc; bnd_loc], first split at |int_loc| (factorization.jl:56-57,108-110)
template <class T>
static void mf_compress_schur_dense(hs_handle* h, int id, const T* SB, int lds, const T* C_, int ldc, const T* M, int ldm, const T* Z, int ldz, int r1, int r2,
                                    hipStream_t stream) {
  NodeH& x = h->nodes[id];
  const int64_t fs = (x.n1p > 0 && x.n1p < x.nb) ? x.n1p : 0;
  hs_hss_options o = mf_options(h, id, 1.0, fs, x.last_ks, x.nb, (C_ && Z) ? r1 : 0);
  hs_hss* H = nullptr;
  const int st = h->is_complex ? hs_hss_compress_lru_z(x.nb, (const double*)SB, lds, (const double*)C_, ldc, (const double*)M, ldm, (const double*)Z, ldz, r1, r2, 1,
                                                       x.sperm.data(), &o, stream, &H)
                               : hs_hss_compress_lru_d(x.nb, (const double*)SB, lds, (const double*)C_, ldc, (const double*)M, ldm, (const double*)Z, ldz, r1, r2, 1,
                                                       x.sperm.data(), &o, stream, &H);
  mf_check(st);
  x.S_hss = H;
  x.last_ks = (int)hs_hss_samples(H);
  mf_maxrank(h, hs_hss_rank(H));
  if (h->opts.verbose) fprintf(stderr, "[hs] node %d (level %d, nb=%d): hssrank(S)=%lld (%lld samples)\n", id, x.level, x.nb, (long long)hs_hss_rank(H), (long long)hs_hss_samples(H));
}

// The same for several fronts of one level at once: ONE batched compression (hs_hss_compress_lru_multi) -- compressed one at a time each is a
// chain of ~10,000 small dependent launches (110 ms for nb = 12,097; the eight transition fronts of Poisson 128^3: 641 ms with two at a time).
template <class T>
struct MfSchurArgs {
  int id;
  const T* SB; int lds;
  const T* C; int ldc;
  const T* M; int ldm;
  const T* Z; int ldz;
  int r1, r2;
};
template <class T>
static void mf_compress_schur_dense_group(hs_handle* h, const std::vector<MfSchurArgs<T>>& a, hipStream_t stream);
template <class T>
static void mf_compress_schur_dense_batch(hs_handle* h, const std::vector<MfSchurArgs<T>>& a, hipStream_t stream) {
  static const bool batch_on = getenv("HS_MF_BATCH") && getenv("HS_MF_BATCH")[0] == '1';  // off by default: measured no faster than one compression per host thread
  const int cnt = (int)a.size();
  if (cnt == 0) return;
  if (cnt == 1 || !batch_on) {
    mf_parallel(h, cnt, [&](int t, hipStream_t st) {
      const MfSchurArgs<T>& q = a[t];
      mf_compress_schur_dense<T>(h, q.id, q.SB, q.lds, q.C, q.ldc, q.M, q.ldm, q.Z, q.ldz, q.r1, q.r2, st);
    });
    return;
  }
  // a single chain leaves the chip idle at every host round trip (rank decisions): two or more forests side by side fill each other's gaps
  static const int groups_env = getenv("HS_MF_BATCH_GROUPS") ? atoi(getenv("HS_MF_BATCH_GROUPS")) : 2;
  const int groups = std::max(1, std::min(groups_env, cnt / 2));
  if (groups > 1) {
    std::vector<std::vector<MfSchurArgs<T>>> part((size_t)groups);
    for (int t = 0; t < cnt; ++t) part[(size_t)(t % groups)].push_back(a[t]);
    mf_parallel(h, groups, [&](int g, hipStream_t st) { mf_compress_schur_dense_group<T>(h, part[(size_t)g], st); });
    return;
  }
  mf_compress_schur_dense_group<T>(h, a, stream);
}
template <class T>
static void mf_compress_schur_dense_group(hs_handle* h, const std::vector<MfSchurArgs<T>>& a, hipStream_t stream) {
  const int cnt = (int)a.size();
  HS_HIP(hipStreamSynchronize(h->stream));
  std::vector<int64_t> n(cnt), ldb(cnt), ldc(cnt), ldm(cnt), ldz(cnt), r1(cnt), r2(cnt);
  std::vector<const double*> B(cnt), Cp(cnt), Mp(cnt), Zp(cnt);
  std::vector<const int64_t*> perm(cnt);
  std::vector<hs_hss_options> opt(cnt);
  std::vector<const hs_hss_options*> po(cnt);
  std::vector<hs_hss*> out(cnt, nullptr);
  for (int t = 0; t < cnt; ++t) {
    const MfSchurArgs<T>& q = a[t];
    NodeH& x = h->nodes[q.id];
    const int64_t fs = (x.n1p > 0 && x.n1p < x.nb) ? x.n1p : 0;
    opt[t] = mf_options(h, q.id, 1.0, fs, x.last_ks, x.nb, (q.C && q.Z) ? q.r1 : 0);
    po[t] = &opt[t];
    n[t] = x.nb; B[t] = (const double*)q.SB; ldb[t] = q.lds;
    Cp[t] = (const double*)q.C; ldc[t] = q.ldc; Mp[t] = (const double*)q.M; ldm[t] = q.ldm; Zp[t] = (const double*)q.Z; ldz[t] = q.ldz;
    r1[t] = (q.C && q.M && q.Z) ? q.r1 : 0; r2[t] = (q.C && q.M && q.Z) ? q.r2 : 0;
    perm[t] = x.sperm.data();
  }
  const int st = h->is_complex ? hs_hss_compress_lru_multi_z(cnt, n.data(), B.data(), ldb.data(), Cp.data(), ldc.data(), Mp.data(), ldm.data(), Zp.data(), ldz.data(), r1.data(),
                                                             r2.data(), perm.data(), po.data(), (void*)stream, out.data())
                               : hs_hss_compress_lru_multi_d(cnt, n.data(), B.data(), ldb.data(), Cp.data(), ldc.data(), Mp.data(), ldm.data(), Zp.data(), ldz.data(), r1.data(),
                                                             r2.data(), perm.data(), po.data(), (void*)stream, out.data());
  mf_check(st);
  for (int t = 0; t < cnt; ++t) {
    NodeH& x = h->nodes[a[t].id];
    x.S_hss = out[t];
    x.last_ks = (int)hs_hss_samples(out[t]);
    mf_maxrank(h, hs_hss_rank(out[t]));
    if (h->opts.verbose)
      fprintf(stderr, "[hs] node %d (level %d, nb=%d): hssrank(S)=%lld (%lld samples, batch of %d)\n", a[t].id, x.level, x.nb, (long long)hs_hss_rank(out[t]), (long long)hs_hss_samples(out[t]), cnt);
  }
}

// the blocks of one child's Schur complement a parent reads (factorization.jl:127-135): S.A11 / S.A22 as views sharing the generators
struct MfChild {
  hs_hss* S = nullptr;
  hs_hss* a11 = nullptr;
  hs_hss* a22 = nullptr;
  int n1 = 0, n2 = 0, r12 = 0, r21 = 0;  // sizes of the two blocks, ranks of A12 = C*Z (r12 = rank of the bnd basis) and A21
  ~MfChild() { close(); }
  void close() {
    if (a11) hs_hss_free(a11);
    if (a22) hs_hss_free(a22);
    a11 = a22 = nullptr;
  }
  void open(hs_hss* S_, int n1_, int nb, hipStream_t s) {
    S = S_;
    n1 = n1_;
    n2 = nb - n1_;
    mf_check(hs_hss_set_stream(S, (void*)s));  // the stream it was compressed on may belong to a worker that is gone; views inherit this one
    if (n1 > 0 && n2 > 0) {
      mf_check(hs_hss_child(S, 0, &a11));
      mf_check(hs_hss_child(S, 1, &a22));
      int64_t i1[8], i2[8];
      mf_check(hs_hss_node_info(S, 1, i1));
      mf_check(hs_hss_node_info(S, 2, i2));
      r12 = (int)i2[6];  // A12 = (U_1 B12) U_2^T
      r21 = (int)i1[6];  // A21 = (U_2 B21) U_1^T
    } else if (n1 > 0) {
      mf_check(hs_hss_child(S, 2, &a11));
    } else {
      mf_check(hs_hss_child(S, 2, &a22));
    }
  }
};

template <class T>
struct MfBuf {  // device buffers of one front's elimination, released on every exit path (recycled blocks: a hipFree would synchronise the
                // device under the other fronts that are being compressed)
  std::vector<void*> p;
  ~MfBuf() {
    for (void* q : p) hs_lr_free(q);
  }
  T* get(size_t elems, const char* what) {
    void* q = nullptr;
    const size_t bytes = (elems + 32) * sizeof(T);
    if (hs_lr_alloc(&q, bytes) != 0) HS_FAIL(HS_ERR_NOMEM, 0, "hipMalloc of %.3f GiB for %s failed", bytes / 1073741824.0, what);
    p.push_back(q);
    return (T*)q;
  }
};

// exact factorization of a sparse coupling block X (rows x cols, few entries) by its nonzero rows or columns, whichever are fewer:
// X = C*Z with unit entries in one factor and the values in the other; returns the rank and the fill lists (positions relative to
// the column offset c_off of C and the row offset c_off of Z)
static int mf_factor_coupling(const MfCoupling& X, int c_off, std::vector<HsFillEntry>& fc, std::vector<HsFillEntry>& fz) {
  if (X.size() == 0) return 0;
  std::vector<int> rows(X.row), cols(X.col);
  std::sort(rows.begin(), rows.end());
  rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
  std::sort(cols.begin(), cols.end());
  cols.erase(std::unique(cols.begin(), cols.end()), cols.end());
  const bool by_rows = rows.size() <= cols.size();
  const std::vector<int>& key = by_rows ? rows : cols;
  auto idx = [&](int v) { return (int)(std::lower_bound(key.begin(), key.end(), v) - key.begin()); };
  if (by_rows) {  // C[r, k(r)] = 1, Z[k(r), c] = value
    for (size_t k = 0; k < rows.size(); ++k) fc.push_back(HsFillEntry{rows[k], c_off + (int)k, -1});
    for (size_t t = 0; t < X.size(); ++t) fz.push_back(HsFillEntry{c_off + idx(X.row[t]), X.col[t], X.e[t]});
  } else {  // C[r, k(c)] = value, Z[k(c), c] = 1
    for (size_t t = 0; t < X.size(); ++t) fc.push_back(HsFillEntry{X.row[t], c_off + idx(X.col[t]), X.e[t]});
    for (size_t k = 0; k < cols.size(); ++k) fz.push_back(HsFillEntry{c_off + (int)k, cols[k], -1});
  }
  return (int)key.size();
}

template <class T>
static void mf_fill(hs_handle* h, const std::vector<HsFillEntry>& f, T* out, int ld, MfBuf<T>& buf, hipStream_t s) {
  if (f.empty()) return;
  HsFillEntry* d = (HsFillEntry*)buf.get((f.size() * sizeof(HsFillEntry) + sizeof(T) - 1) / sizeof(T), "coupling entries");
  HS_HIP(hipMemcpyAsync(d, f.data(), f.size() * sizeof(HsFillEntry), hipMemcpyHostToDevice, s));
  HS_HIP(hipStreamSynchronize(s));
  launch_fill_entries<T>(d, (int)f.size(), (const T*)h->d_nz, out, ld, s);
}

// ---- numeric: the matrix-free fronts of one level, one at a time (they are the few large fronts at the top of the tree) --------------
template <class T>
static void factor_mf_fronts(hs_handle* h, const int* ids, int count) {
  static const bool vt = getenv("HS_VERBOSE_COMPRESS") != nullptr;
  const bool say = h->opts.verbose || vt;
  hs_sparse_dev As{h->n, h->d_colptr, h->d_rowval, h->d_nz, h->d_rowptr, h->d_colind, h->d_nzr};
  const double dsc = std::pow(10.0, -(double)(h->opts.hss_dexp == 0 ? 2 : h->opts.hss_dexp - 1));  // hs_options.hss_dexp
  mf_parallel(h, count, [&](int k, hipStream_t s) {
    const int id = ids[k];
    NodeH& x = h->nodes[id];
    NodeH &c1 = h->nodes[x.left], &c2 = h->nodes[x.right];
    if (!c1.S_hss || !c2.S_hss) HS_FAIL(HS_ERR_ARGUMENT, id, "internal: node %d is matrix-free but a child holds no HSS Schur complement", id);
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
      if (!vt) return;
      (void)hipStreamSynchronize(s);
      auto now = std::chrono::steady_clock::now();
      fprintf(stderr, "[hs mf] node %d (level %d, ni=%d, nb=%d): %-30s %9.3f ms\n", id, x.level, x.ni, x.nb, what, std::chrono::duration<double, std::milli>(now - t0).count());
      t0 = now;
    };
    MfChild ch1, ch2;
    ch1.open((hs_hss*)c1.S_hss, c1.n1p, c1.nb, s);
    ch2.open((hs_hss*)c2.S_hss, c2.n1p, c2.nb, s);
    if (ch1.n1 != x.ni1 || ch1.n1 + ch2.n1 != x.ni || ch1.n2 != x.nb1 || ch1.n2 + ch2.n2 != x.nb)
      HS_FAIL(HS_ERR_DIMENSION, id, "internal: children of node %d contribute (%d+%d, %d+%d) DOFs, expected (%d, %d)", id, ch1.n1, ch2.n1, ch1.n2, ch2.n2, x.ni, x.nb);
    std::vector<int64_t> gid((size_t)x.m);
    for (int p = 0; p < x.m; ++p) gid[(size_t)p] = h->fidx_host[x.off_fidx + p];
    MfBuf<T> buf;
    // the children's Schur complements are released once absorbed (the reference keeps every S although only the root's is used)
    auto release_children = [&]() {
      if (h->opts.keep_schur) return;
      ch1.close();
      ch2.close();
      hs_hss_free((hs_hss*)c1.S_hss);
      hs_hss_free((hs_hss*)c2.S_hss);
      c1.S_hss = c2.S_hss = nullptr;
    };
    // ---- Aib = C_R*Z_R and Abi = C_L*Z_L from the children's generators + the sparse couplings (nothing is recompressed): ranks ------------
    std::vector<HsFillEntry> fcR, fzR, fcL, fzL;
    const int oR2 = ch1.r12, oRx = ch1.r12 + ch2.r12;
    const int rR = x.nb > 0 ? oRx + mf_factor_coupling(x.xr, oRx, fcR, fzR) : 0;
    const int oL2 = ch1.r21, oLx = ch1.r21 + ch2.r21;
    const int rL = x.nb > 0 ? oLx + mf_factor_coupling(x.xl, oLx, fcL, fzL) : 0;
    LowRank<T>*lrL = nullptr, *lrR = nullptr;
    auto zalloc = [&](T** p, size_t elems, const char* what) {
      dmalloc((void**)p, (elems + 32) * sizeof(T), what);
      HS_HIP(hipMemsetAsync(*p, 0, (elems + 32) * sizeof(T), s));
    };
    if (x.nb > 0) {
      lrL = new LowRank<T>();
      lrR = new LowRank<T>();
      x.lrL = lrL;
      x.lrR = lrR;
      lrL->rows = x.nb; lrL->cols = x.ni; lrL->r = rL; lrL->k = rL;
      lrR->rows = x.ni; lrR->cols = x.nb; lrR->r = rR; lrR->k = rR;
      lrL->ldc = rup(std::max(x.nb, 1), 2);
      lrR->ldz = rup(std::max(rR, 1), 2);
      zalloc(&lrL->Cd, (size_t)lrL->ldc * std::max(rL, 1), "C_L");
      zalloc(&lrR->Z, (size_t)lrR->ldz * std::max(x.nb, 1), "Z_R");
      x.last_rL = rL;
      x.last_rR = rR;
      mf_maxrank(h, std::max(rL, rR));
      if (say)
        fprintf(stderr, "[hs] node %d (level %d, ni=%d, nb=%d): rank(L)=%d (%d+%d from the children, %d sparse) rank(R)=%d (%d+%d, %d sparse)\n", id, x.level, x.ni,
                x.nb, rL, ch1.r21, ch2.r21, rL - oLx, rR, ch1.r12, ch2.r12, rR - oRx);
    }
    auto offd = [&](MfChild& c, int which, T* Cp, int ldc, T* Zp, int ldz) {
      if (c.n1 == 0 || c.n2 == 0) return;
      mf_check(hs_hss_set_stream(c.S, (void*)s));
      mf_check(hs_hss_offdiag(c.S, which, (double*)Cp, ldc, (double*)Zp, ldz, 1));
    };
    // fills C_L, Z_R (owned by the low-rank objects) and Z_L (rL x ni), C_R (ni x rR) wherever the formulation of D wants them
    auto generators = [&](T* ZL, int ldzl, T* CR, int ldcr) {
      if (x.nb == 0) return;
      // Aib: rows int, columns bnd;  block (1,1) = A12 of S1, block (2,2) = A12 of S2
      offd(ch1, 0, CR, ldcr, lrR->Z, lrR->ldz);
      offd(ch2, 0, CR + ch1.n1 + (size_t)oR2 * ldcr, ldcr, lrR->Z + oR2 + (size_t)ch1.n2 * lrR->ldz, lrR->ldz);
      // Abi: rows bnd, columns int;  block (1,1) = A21 of S1, block (2,2) = A21 of S2
      offd(ch1, 1, lrL->Cd, lrL->ldc, ZL, ldzl);
      offd(ch2, 1, lrL->Cd + ch1.n2 + (size_t)oL2 * lrL->ldc, lrL->ldc, ZL + oL2 + (size_t)ch1.n1 * ldzl, ldzl);
      mf_fill<T>(h, fcR, CR, ldcr, buf, s);
      mf_fill<T>(h, fzR, lrR->Z, lrR->ldz, buf, s);
      mf_fill<T>(h, fcL, lrL->Cd, lrL->ldc, buf, s);
      mf_fill<T>(h, fzL, ZL, ldzl, buf, s);
    };
    T* Mm = nullptr;  // Abi*R = C_L * M * Z_R, M = Z_L * Aii^-1 * C_R  (rL x rR)
    int ldm = 2;
    hs_hss_blockop opd{ch1.n1, ch2.n1, ch1.a11, ch2.a11, gid.data(), &As, (int32_t*)h->d_lpos};
    if (x.mfd) {
      // ---- D = Aii expanded from the operator [S1.A11 A[int1,int2]; A[int2,int1] S2.A11] and eliminated DENSELY ---------------------------------
      // The front [D C_R; Z_L 0] of size ni + max(rL, rR) is an ordinary front for the elimination kernels: its LU leaves Z_L*U^-1 below D,
      // L^-1*P*C_R beside it and -M in the corner -- the low-rank Gauss transforms in the form hs_compress.h applies in ldiv!.
      const int nbp = std::max(2, rup(std::max(rL, rR), 2));
      const int m = x.ni + nbp, ldl = rup(m, 2), ldu = rup(std::max(x.ni, 1), 2), ldsb = nbp;
      dmalloc(&x.mfd_LF, ((size_t)ldl * x.ni + 32) * sizeof(T), "dense interior block of a matrix-free front");
      dmalloc(&x.mfd_UR, ((size_t)ldu * nbp + 32) * sizeof(T), "L^-1*P*C_R of a matrix-free front");
      dmalloc(&x.mfd_SB, ((size_t)ldsb * nbp + 32) * sizeof(T), "Z_L*Aii^-1*C_R of a matrix-free front");
      x.mfd_nbp = nbp; x.mfd_ldl = ldl; x.mfd_ldu = ldu;
      T *LF = (T*)x.mfd_LF, *UR = (T*)x.mfd_UR, *SB = (T*)x.mfd_SB;
      const LevelH& L = h->levels[x.level];
      NodeDesc<T>* dslot = (NodeDesc<T>*)h->d_nodes + L.desc_off + x.batch_pos;
      SolveNode<T>* sslot = (SolveNode<T>*)h->d_solve + L.desc_off + x.batch_pos;
      T* dinv = (T*)h->d_inv;
      const int nblk = (x.ni + HS_PB - 1) / HS_PB;
      NodeDesc<T> d;
      memset(&d, 0, sizeof d);
      d.LF = LF; d.UR = UR; d.SB = SB;
      d.invL = dinv + x.off_inv;
      d.invU = d.invL + (size_t)nblk * HS_PB * HS_PB;
      d.inv256L = dinv + x.off_inv256;
      d.inv256U = d.inv256L + (size_t)((x.ni + 255) / 256) * 256 * 256;
      d.ipiv = h->d_int + x.off_ipiv;
      d.rperm = h->d_int + x.off_rperm;
      d.cand0 = h->d_tmpi + x.off_cand;
      d.cand1 = d.cand0 + x.ncand;
      d.pivlist = d.cand1 + x.ncand;
      d.info = h->d_info + id;
      d.growth = h->d_growth + id;
      d.fidx = h->d_int + x.off_fidx;
      d.ni = x.ni; d.nb = nbp; d.m = m;
      d.ldl = ldl; d.ldu = ldu; d.lds = ldsb;
      d.ni1 = x.ni; d.nb1 = nbp;
      d.isleaf = 1;
      d.node = id;
      d.finalize();
      SolveNode<T> q;
      memset(&q, 0, sizeof q);
      q.LF = LF; q.UR = UR; q.invL = d.invL; q.invU = d.invU; q.inv256L = d.inv256L; q.inv256U = d.inv256U;
      q.rperm = d.rperm; q.fidx = d.fidx;
      q.ni = x.ni; q.nb = 0; q.m = x.ni; q.ldl = ldl; q.ldu = ldu;
      q.compressed = 1;
      q.mrows = x.ni;
      q.woff = x.woff;
      HS_HIP(hipMemcpyAsync(dslot, &d, sizeof d, hipMemcpyHostToDevice, s));
      HS_HIP(hipMemcpyAsync(sslot, &q, sizeof q, hipMemcpyHostToDevice, s));
      HS_HIP(hipStreamSynchronize(s));  // d, q are stack objects
      static const int cb_env = getenv("HS_MF_EXPAND_COLS") ? atoi(getenv("HS_MF_EXPAND_COLS")) : 4096;
      const int cb = std::max(256, cb_env), ldi = rup(x.ni, 2);  // columns of the identity per application of the operator
      T* I_ = nullptr;
      std::vector<HsFillEntry> f12, f21;
      const bool lone = (s == h->stream);  // the look-ahead streams belong to the handle: only a front that runs alone may use them
      Profiler prof;                       // h->prof is not shared between concurrent fronts
      for (int attempt = 0; attempt < 2; ++attempt) {
        HS_HIP(hipMemsetAsync(LF, 0, ((size_t)ldl * x.ni + 32) * sizeof(T), s));
        HS_HIP(hipMemsetAsync(UR, 0, ((size_t)ldu * nbp + 32) * sizeof(T), s));
        HS_HIP(hipMemsetAsync(SB, 0, ((size_t)ldsb * nbp + 32) * sizeof(T), s));
        static const bool by_apply = getenv("HS_MF_EXPAND_APPLY") != nullptr;  // diagnostics: expand D by applying the operator to the identity
        if (by_apply) {
          if (!I_) I_ = buf.get((size_t)ldi * cb, "identity block");
          for (int c0 = 0; c0 < x.ni; c0 += cb) {
            const int nc = std::min(cb, x.ni - c0);
            launch_identity_cols<T>(I_, ldi, x.ni, c0, nc, s);
            mf_check(hs_hss_blockop_apply(&opd, h->is_complex, (const double*)I_, ldi, (double*)(LF + (size_t)c0 * ldl), ldl, nc, 0, s));
          }
        } else {  // the two HSS blocks expanded in place (2 n^2 r flops each), the sparse couplings A[int1,int2], A[int2,int1] entry by entry
          if (ch1.n1 > 0) {
            mf_check(hs_hss_set_stream(ch1.a11, (void*)s));
            mf_check(hs_hss_expand(ch1.a11, (double*)LF, ldl, 1));
          }
          if (ch2.n1 > 0) {
            mf_check(hs_hss_set_stream(ch2.a11, (void*)s));
            mf_check(hs_hss_expand(ch2.a11, (double*)(LF + ch1.n1 + (size_t)ch1.n1 * ldl), ldl, 1));
          }
          if (attempt == 0 && f12.empty() && f21.empty()) {
            for (size_t t = 0; t < x.x12.size(); ++t) f12.push_back(HsFillEntry{x.x12.row[t], ch1.n1 + x.x12.col[t], x.x12.e[t]});
            for (size_t t = 0; t < x.x21.size(); ++t) f21.push_back(HsFillEntry{ch1.n1 + x.x21.row[t], x.x21.col[t], x.x21.e[t]});
          }
          mf_fill<T>(h, f12, LF, ldl, buf, s);
          mf_fill<T>(h, f21, LF, ldl, buf, s);
        }
        generators(LF + x.ni, ldl, UR, ldu);
        if (attempt == 0) lap("D: Aii expanded from the generators; Aib, Abi");
        launch_init_fronts<T>(dslot, 1, m, s);
        bool opt;
        {
          std::lock_guard<std::mutex> lk(g_mf_mu);
          opt = h->optimistic;
        }
        static const bool opt_env = !(getenv("HS_OPTIMISTIC") && getenv("HS_OPTIMISTIC")[0] == '0');
        opt = opt && opt_env && attempt == 0;
        int hni = x.ni, hnb = nbp;
        Sched<T> sch{dslot, 1, x.ni, nbp, m, s, &prof, &hni, &hnb, lone ? h->stream2 : nullptr, 0, lone ? h->stream_la : nullptr, lone ? h->stream2m : nullptr};
        sch.sn = sslot;
        sch.optimistic = opt;
        sch.factor_fronts();
        if (!opt) break;
        int gr = 0;
        HS_HIP(hipMemcpyAsync(&gr, h->d_growth + id, sizeof(int), hipMemcpyDeviceToHost, s));
        HS_HIP(hipStreamSynchronize(s));
        if (gr == 0) break;
        {
          std::lock_guard<std::mutex> lk(g_mf_mu);
          h->optimistic = false;
        }
        if (h->opts.verbose) fprintf(stderr, "[hs] node %d: a pivot outside the diagonal block was needed; redoing its interior block with tournament pivoting\n", id);
      }
      lap("D: dense elimination");
      if (x.nb > 0) {
        launch_negate<T>(SB, ldsb, rL, rR, s);  // the elimination left Abb - Abi*Aii^-1*Aib = -M in the corner
        Mm = SB;
        ldm = ldsb;
        lrL->Z = LF + x.ni;  // Z_L*U^-1 (borrowed: free_mfd_buffers)
        lrL->ldz = ldl;
        lrR->Cd = UR;        // L^-1*P*C_R (borrowed)
        lrR->ldc = ldu;
      }
    } else {
      const bool blockD = x.mfb && ch1.n1 > 0 && ch2.n1 > 0 && ch1.a11 && ch2.a11;
      x.mfb = blockD;
      if (blockD) {
        // ---- D = blockfactor([A11 A12; A21 A22]) over HSS blocks (src/blockmatrix.jl:121-130): A11 = S1.A11 is the left child's own HSS block
        // (nothing is compressed again), A12 / A21 are the sparse couplings factored exactly by their nonzero rows / columns, and the Schur
        // complement S22 = A22 - A21*A11^-1*A12 = S2.A11 - C21*(Z21*A11^-1*C12)*Z12 is RECOMPRESSED from that operator (the role of `recompress!`)
        const int n1 = ch1.n1, n2 = ch2.n1;
        std::vector<HsFillEntry> fc12, fz12, fc21, fz21;
        const int k12 = mf_factor_coupling(x.x12, 0, fc12, fz12), k21 = mf_factor_coupling(x.x21, 0, fc21, fz21);
        x.bk12 = k12; x.bk21 = k21;
        x.bldw = rup(n1, 2); x.bldz12 = rup(std::max(k12, 1), 2); x.bldc21 = rup(n2, 2); x.bldz21 = rup(std::max(k21, 1), 2);
        T *W12 = nullptr, *Z12 = nullptr, *C21 = nullptr, *Z21 = nullptr;
        zalloc(&W12, (size_t)x.bldw * std::max(k12, 1), "A11^-1*C12");
        x.bW12 = W12;
        zalloc(&Z12, (size_t)x.bldz12 * n2, "Z12");
        x.bZ12 = Z12;
        zalloc(&C21, (size_t)x.bldc21 * std::max(k21, 1), "C21");
        x.bC21 = C21;
        zalloc(&Z21, (size_t)x.bldz21 * n1, "Z21");
        x.bZ21 = Z21;
        mf_fill<T>(h, fc12, W12, x.bldw, buf, s);
        mf_fill<T>(h, fz12, Z12, x.bldz12, buf, s);
        mf_fill<T>(h, fc21, C21, x.bldc21, buf, s);
        mf_fill<T>(h, fz21, Z21, x.bldz21, buf, s);
        x.hss = ch1.a11;  // the view now belongs to this front; the matrix it shares its generators with stays alive with it
        ch1.a11 = nullptr;
        if (!h->opts.keep_schur) {
          x.hss_keep = c1.S_hss;
          c1.S_hss = nullptr;
        }
        mf_check(hs_hss_set_stream((hs_hss*)x.hss, (void*)s));
        int st = hs_hss_factor((hs_hss*)x.hss);
        if (st == 0 && k12 > 0) st = hs_hss_ldiv((hs_hss*)x.hss, (double*)W12, x.bldw, k12, 1);
        if (st != 0) {
          if (st == HS_ERR_SINGULAR) hs_set_error(HS_ERR_SINGULAR, id, "SingularException: the block A11 of the interior block of node %d is singular", id);
          throw HsError{st};
        }
        lap("D: A11 eliminated, A11^-1 A12");
        T* M22 = nullptr;
        int ldm22 = 2;
        if (k12 > 0 && k21 > 0) {  // Z21 * (A11^-1 C12)
          ldm22 = rup(k21, 2);
          M22 = buf.get((size_t)ldm22 * k12, "Z21*A11^-1*C12");
          HS_HIP(hipMemsetAsync(M22, 0, sizeof(T) * (size_t)ldm22 * k12, s));
          GemmProb<T> gp{Z21, W12, M22, k21, k12, n1, x.bldz21, x.bldw, ldm22};
          GemmProb<T>* dgp = (GemmProb<T>*)buf.get((sizeof(GemmProb<T>) + sizeof(T) - 1) / sizeof(T), "GEMM descriptor");
          HS_HIP(hipMemcpyAsync(dgp, &gp, sizeof gp, hipMemcpyHostToDevice, s));
          HS_HIP(hipStreamSynchronize(s));
          launch_gemm_probs<T>(dgp, 1, k21, k12, 0, s);
        }
        // the couplings of a two-layer separator have FULL rank (k12 = k21 = ni/2): the update is folded into two factors, C21 * (M22*Z12),
        // so that the compression corrects a block of entries with one product instead of two through a 16,384 x 16,384 middle factor
        T* MZ = nullptr;
        if (k12 > 0 && k21 > 0) {
          MZ = buf.get((size_t)ldm22 * n2, "Z21*A11^-1*A12");
          HS_HIP(hipMemsetAsync(MZ, 0, sizeof(T) * (size_t)ldm22 * n2, s));
          GemmProb<T> gp{M22, Z12, MZ, k21, n2, k12, ldm22, x.bldz12, ldm22};
          GemmProb<T>* dgp = (GemmProb<T>*)buf.get((sizeof(GemmProb<T>) + sizeof(T) - 1) / sizeof(T), "GEMM descriptor");
          HS_HIP(hipMemcpyAsync(dgp, &gp, sizeof gp, hipMemcpyHostToDevice, s));
          HS_HIP(hipStreamSynchronize(s));
          launch_gemm_probs<T>(dgp, 1, k21, n2, 0, s);
        }
        hs_hss_blockop op22{n2, 0, ch2.a11, nullptr, gid.data() + n1, &As, (int32_t*)h->d_lpos};
        hs_hss_options o = mf_options(h, id, dsc, 0, x.last_k, n2);
        hs_hss* S22 = nullptr;
        const bool u22 = k12 > 0 && k21 > 0;
        mf_check(h->is_complex ? hs_hss_compress_blockop_z(&op22, (const double*)C21, x.bldc21, nullptr, 0, (const double*)MZ, ldm22, u22 ? k21 : 0, u22 ? k21 : 0, nullptr, &o, s, &S22)
                               : hs_hss_compress_blockop_d(&op22, (const double*)C21, x.bldc21, nullptr, 0, (const double*)MZ, ldm22, u22 ? k21 : 0, u22 ? k21 : 0, nullptr, &o, s, &S22));
        x.hss2 = S22;
        lap("D: S22 = A22 - A21 A11^-1 A12 recompressed");
        st = hs_hss_factor(S22);
        if (st != 0) {
          if (st == HS_ERR_SINGULAR) hs_set_error(HS_ERR_SINGULAR, id, "SingularException: the Schur complement S22 of the interior block of node %d is singular", id);
          throw HsError{st};
        }
        lap("D: S22 eliminated");
        x.last_k = (int)hs_hss_samples(S22);
        const long long rD = std::max<long long>(hs_hss_rank((hs_hss*)x.hss), hs_hss_rank(S22));
        mf_maxrank(h, rD);
        if (say)
          fprintf(stderr, "[hs] node %d (level %d, ni=%d, nb=%d): blockfactor(D): hssrank(A11)=%lld, couplings of rank %d / %d, hssrank(S22)=%lld (%lld samples)\n", id, x.level,
                  x.ni, x.nb, (long long)hs_hss_rank((hs_hss*)x.hss), k12, k21, (long long)hs_hss_rank(S22), (long long)hs_hss_samples(S22));
        if (!x.ht) dmalloc(&x.ht, ((size_t)x.ni + 32) * sizeof(T), "HSS solve vector");
      } else
      // ---- D = Aii as one HSS matrix, compressed from the operator [S1.A11 A[int1,int2]; A[int2,int1] S2.A11] ------------------------------
      {
        hs_hss_options o = mf_options(h, id, dsc, 0, x.last_k, x.ni);
        hs_hss* D = nullptr;
        const int64_t* q = x.ilv.empty() ? nullptr : x.ilv.data();
        mf_check(h->is_complex ? hs_hss_compress_blockop_z(&opd, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, q, &o, s, &D)
                               : hs_hss_compress_blockop_d(&opd, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, q, &o, s, &D));
        x.hss = D;
        lap("D: compress(Aii) matrix-free");
        int st = hs_hss_factor(D);
        if (st != 0) {
          if (st == HS_ERR_SINGULAR) hs_set_error(HS_ERR_SINGULAR, id, "SingularException: the HSS form of the interior block of node %d is singular", id);
          throw HsError{st};
        }
        lap("D: HSS elimination");
        x.last_k = (int)hs_hss_samples(D);
        mf_maxrank(h, hs_hss_rank(D));
        if (say) fprintf(stderr, "[hs] node %d (level %d, ni=%d, nb=%d): hssrank(D)=%lld (%lld samples), matrix-free\n", id, x.level, x.ni, x.nb, (long long)hs_hss_rank(D), (long long)hs_hss_samples(D));
        if (!x.ht) dmalloc(&x.ht, ((size_t)x.ni + 32) * sizeof(T), "HSS solve vector");
      }
      if (x.nb > 0) {
        lrL->ldz = rup(std::max(rL, 1), 2);
        lrR->ldc = rup(std::max(x.ni, 1), 2);
        zalloc(&lrL->Z, (size_t)lrL->ldz * std::max(x.ni, 1), "Z_L");
        zalloc(&lrR->Cd, (size_t)lrR->ldc * std::max(rR, 1), "C_R");
        generators(lrL->Z, lrL->ldz, lrR->Cd, lrR->ldc);
        lap("Aib, Abi from the generators");
        // ---- W = Aii^-1 * C_R (the R transform; C_R itself is not needed again) -------------------------------------------------------------------
        if (rR > 0) {
          hss_d_solve<T>(h, x, lrR->Cd, lrR->ldc, rR, s);
          x.hW = lrR->Cd;  // ownership moves to the node (freed with the HSS objects); the low-rank object keeps only Z_R
          x.hldw = lrR->ldc;
          lrR->Cd = nullptr;
          lap("W = D^-1 C_R");
          if (rL > 0) {  // M = Z_L * W  (rL x rR):  Abi*R = C_L * M * Z_R
            ldm = rup(rL, 2);
            Mm = buf.get((size_t)ldm * rR, "Z_L*W");
            HS_HIP(hipMemsetAsync(Mm, 0, sizeof(T) * (size_t)ldm * rR, s));
            GemmProb<T> gp{lrL->Z, (const T*)x.hW, Mm, rL, rR, x.ni, lrL->ldz, x.hldw, ldm};
            GemmProb<T>* dgp = (GemmProb<T>*)buf.get((sizeof(GemmProb<T>) + sizeof(T) - 1) / sizeof(T), "GEMM descriptor");
            HS_HIP(hipMemcpyAsync(dgp, &gp, sizeof gp, hipMemcpyHostToDevice, s));
            HS_HIP(hipStreamSynchronize(s));
            launch_gemm_probs<T>(dgp, 1, rL, rR, 0, s);
          }
        }
      }
    }
    if (x.nb == 0) {
      HS_HIP(hipStreamSynchronize(s));
      release_children();
      return;
    }
    const bool upd = rL > 0 && rR > 0;
    // ---- S = P (Abb - Abi*R) P' from products and entries of the operator [S1.A22 A[bnd1,bnd2]; A[bnd2,bnd1] S2.A22] - C_L*M*Z_R ----------
    hs_hss_blockop opb{ch1.n2, ch2.n2, ch1.a22, ch2.a22, gid.data() + x.ni, &As, (int32_t*)h->d_lpos};
    if (x.s_hss) {
      const int64_t fs = (x.n1p > 0 && x.n1p < x.nb) ? x.n1p : 0;
      hs_hss_options o = mf_options(h, id, 1.0, fs, x.last_ks, x.nb, rL);
      hs_hss* Sh = nullptr;
      mf_check(h->is_complex ? hs_hss_compress_blockop_z(&opb, (const double*)lrL->Cd, lrL->ldc, (const double*)Mm, ldm, (const double*)lrR->Z, lrR->ldz, upd ? rL : 0,
                                                         upd ? rR : 0, x.sperm.data(), &o, s, &Sh)
                             : hs_hss_compress_blockop_d(&opb, (const double*)lrL->Cd, lrL->ldc, (const double*)Mm, ldm, (const double*)lrR->Z, lrR->ldz, upd ? rL : 0,
                                                         upd ? rR : 0, x.sperm.data(), &o, s, &Sh));
      x.S_hss = Sh;
      x.last_ks = (int)hs_hss_samples(Sh);
      mf_maxrank(h, hs_hss_rank(Sh));
      if (say) fprintf(stderr, "[hs] node %d (level %d, nb=%d): hssrank(S)=%lld (%lld samples), matrix-free\n", id, x.level, x.nb, (long long)hs_hss_rank(Sh), (long long)hs_hss_samples(Sh));
      lap("S: compress(Abb - Abi R) matrix-free");
    } else {
      // the parent assembles a dense front: S is formed by applying the operator to the identity, a block of columns at a time
      T* SB = x.ext_sb ? (T*)x.ext_sb : (T*)h->d_sb + x.off_SB;
      const int cb = 1024, ldi = rup(x.nb, 2);
      T* I_ = buf.get((size_t)ldi * cb, "identity block");
      T* t1 = upd ? buf.get((size_t)ldm * cb, "M*Z_R block") : nullptr;
      GemmProb<T>* dgp = (GemmProb<T>*)buf.get((2 * sizeof(GemmProb<T>) + sizeof(T) - 1) / sizeof(T), "GEMM descriptors");
      for (int c0 = 0; c0 < x.nb; c0 += cb) {
        const int nc = std::min(cb, x.nb - c0);
        launch_identity_cols<T>(I_, ldi, x.nb, c0, nc, s);
        mf_check(hs_hss_blockop_apply(&opb, h->is_complex, (const double*)I_, ldi, (double*)(SB + (size_t)c0 * x.lds), x.lds, nc, 0, s));
        if (upd) {
          HS_HIP(hipMemsetAsync(t1, 0, sizeof(T) * (size_t)ldm * nc, s));
          GemmProb<T> gp[2] = {GemmProb<T>{Mm, lrR->Z + (size_t)c0 * lrR->ldz, t1, rL, nc, rR, ldm, lrR->ldz, ldm},
                               GemmProb<T>{lrL->Cd, t1, SB + (size_t)c0 * x.lds, x.nb, nc, rL, lrL->ldc, ldm, x.lds}};
          HS_HIP(hipMemcpyAsync(dgp, gp, sizeof gp, hipMemcpyHostToDevice, s));
          HS_HIP(hipStreamSynchronize(s));
          launch_gemm_probs<T>(dgp, 1, rL, nc, 0, s);
          launch_gemm_probs<T>(dgp + 1, 1, x.nb, nc, 1, s);
        }
      }
      lap("S: formed densely for a dense parent");
    }
    HS_HIP(hipStreamSynchronize(s));
    release_children();
  });
}
