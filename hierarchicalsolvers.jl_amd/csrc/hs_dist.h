// hs_dist.h -- fronts above the rank cut eliminated by their whole group of ranks (hs_options.dist_top).
//
// The reference factors the two subtrees of a node one after the other although they are independent (src/factorization.jl:20-21);
// the subtree-per-rank partition uses that independence below the cut.  Above it a front has a GROUP of 2^k ranks (the owners of the
// subtrees below it) and one elimination: here the independent units are the block columns of the front.
//
//   storage   every rank of the group holds the whole front [LF | UR | SB] (the children's Schur complements reach every member: the
//             two sibling groups swap them pairwise, rank r <-> r +- |child group|) and assembles it like a rank-local front;
//   compute   block columns of NB interior DOFs are dealt round-robin over the group (1-D block-cyclic).  The owner of block j brings
//             it up to date, factors it (the same recursive panel code as a rank-local front: pivoting among the rows of Aii only),
//             and fans the finished block column out to the group -- [L\U column block, pivots, accumulated row permutation, the 32- and
//             256-wide inverse diagonal blocks] -- while every rank applies block j-1 to ITS block columns and to ITS slice of the
//             boundary columns (UR), look-ahead as in Sched::factor_fronts_lookahead;
//   result    after the last block the column slices of UR and of the Schur complement are gathered inside the group, so every member
//             holds the complete factors: `ldiv!` sweeps run replicated above the cut (no communication in the backward sweep at all)
//             and the parent's group finds both Schur complements after one pairwise swap.
//
// On MI355X the fan-out of one block column to g-1 peers is g-1 concurrent point-to-point transfers over distinct xGMI links.
// Bound of this 1-D form: the panel chain of a block column runs on one rank (a 2-D distribution would split its rows as well).
#pragma once

template <class T>
struct DistFront {
  hs_comm* comm;
  int glo, gcnt, rank;  // group [glo, glo+gcnt), this rank's world rank
  int NB;
  int period;           // consecutive block columns one rank owns (HS_DIST_PERIOD, default 1): owner(j) = glo + (j / period) % gcnt
  hipStream_t sc;       // every transfer of the handle is enqueued here (one order of operations on the communicator)
  // the front
  int ni, nb, m, ldl, ldu, lds;
  T *LF, *UR, *SB, *invL, *invU, *inv256L, *inv256U;
  int* ipiv;            // [ipiv; rperm], 2*ni ints
  T *stage_s, *stage_r; // packing buffers: every transfer carries at most ONE message per peer and direction
  T* stage_u;           // sender's buffer of the U parts (they travel behind the block-column messages and must not hold up the next pack)
  int owner(int j) const { return glo + (j / period) % gcnt; }
};

// Elements of T one block-column message may take: the L part of the widest block column + its inverse diagonal blocks + [ipiv; rperm]
static inline size_t dist_stage_elems(int m, int ni, int NB, size_t esz) {
  return (size_t)m * NB + (size_t)2 * (NB / HS_PB) * HS_PB * HS_PB + (size_t)2 * (NB / 256) * 65536 + ((size_t)2 * ni * sizeof(int) + esz - 1) / esz + 64;
}

// The message of block column j: what a rank needs to APPLY it -- rows c0.. of the column block (the finished L part with Abi*U^-1
// below it), the 32- and 256-wide inverse diagonal blocks, the pivots and the accumulated row permutation.  `pack` != 0: LF -> stage,
// else stage -> LF.  Returns the message size in bytes (the same on every rank).
template <class T>
static size_t dist_panel_message(const DistFront<T>& D, int j, T* stage, int pack, hipStream_t s) {
  const int c0 = j * D.NB, w = std::min(D.NB, D.ni - c0), rows = D.m - c0;
  const int b32 = c0 / HS_PB, n32 = (w + HS_PB - 1) / HS_PB, b256 = c0 / 256, n256 = (w + 255) / 256;
  T* q = stage;
  auto blk = [&](T* dev, size_t cnt) {
    if (pack == 1) HS_HIP(hipMemcpyAsync(q, dev, cnt * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (pack == 0) HS_HIP(hipMemcpyAsync(dev, q, cnt * sizeof(T), hipMemcpyDeviceToDevice, s));
    q += cnt;
  };
  T* Lp = D.LF + (size_t)c0 * D.ldl + c0;
  if (pack == 1) HS_HIP(hipMemcpy2DAsync(q, (size_t)rows * sizeof(T), Lp, (size_t)D.ldl * sizeof(T), (size_t)rows * sizeof(T), w, hipMemcpyDeviceToDevice, s));
  if (pack == 0) HS_HIP(hipMemcpy2DAsync(Lp, (size_t)D.ldl * sizeof(T), q, (size_t)rows * sizeof(T), (size_t)rows * sizeof(T), w, hipMemcpyDeviceToDevice, s));
  q += (size_t)rows * w;
  blk(D.invL + (size_t)b32 * HS_PB * HS_PB, (size_t)n32 * HS_PB * HS_PB);
  blk(D.invU + (size_t)b32 * HS_PB * HS_PB, (size_t)n32 * HS_PB * HS_PB);
  blk(D.inv256L + (size_t)b256 * 65536, (size_t)n256 * 65536);
  blk(D.inv256U + (size_t)b256 * 65536, (size_t)n256 * 65536);
  const size_t ibytes = (size_t)2 * D.ni * sizeof(int);
  if (pack == 1) HS_HIP(hipMemcpyAsync(q, D.ipiv, ibytes, hipMemcpyDeviceToDevice, s));
  if (pack == 0) HS_HIP(hipMemcpyAsync(D.ipiv, q, ibytes, hipMemcpyDeviceToDevice, s));
  return (size_t)((char*)q - (char*)stage) + ibytes;
}

// fan-out of a packed message from the owner of block j to the rest of the group
template <class T>
static void dist_fanout(const DistFront<T>& D, int j, T* from, size_t bytes, hipStream_t s) {
  const int owner = D.owner(j);
  std::vector<HsPiece> sends, recvs;
  if (owner == D.rank) {
    for (int r = D.glo; r < D.glo + D.gcnt; ++r)
      if (r != D.rank) sends.push_back({r, from, bytes});
  } else {
    recvs.push_back({owner, D.stage_r, bytes});
  }
  D.comm->transfer(sends, recvs, s);
}

// The owner has packed block j into stage_s on the stream that factored it; everything else happens on the comm stream.
template <class T>
static void dist_bcast_panel(const DistFront<T>& D, int j, hipStream_t s) {
  const size_t bytes = dist_panel_message(D, j, (T*)nullptr, -1, s);  // size only
  dist_fanout(D, j, D.stage_s, bytes, s);
  if (D.owner(j) != D.rank) dist_panel_message(D, j, D.stage_r, 0, s);
}

// The rows ABOVE the diagonal block of block column j (its part of U) are not needed to apply it; they follow off the critical path
// (every rank ends with the complete factors: the sweeps of ldiv! run replicated above the cut).
template <class T>
static void dist_bcast_upper(const DistFront<T>& D, int j, hipStream_t s) {
  const int c0 = j * D.NB, w = std::min(D.NB, D.ni - c0);
  if (c0 == 0) return;
  const bool mine = D.owner(j) == D.rank;
  const size_t bytes = (size_t)c0 * w * sizeof(T);
  T* Up = D.LF + (size_t)c0 * D.ldl;
  if (mine) HS_HIP(hipMemcpy2DAsync(D.stage_u, (size_t)c0 * sizeof(T), Up, (size_t)D.ldl * sizeof(T), (size_t)c0 * sizeof(T), w, hipMemcpyDeviceToDevice, s));
  dist_fanout(D, j, D.stage_u, bytes, s);
  if (!mine) HS_HIP(hipMemcpy2DAsync(Up, (size_t)D.ldl * sizeof(T), D.stage_r, (size_t)c0 * sizeof(T), (size_t)c0 * sizeof(T), w, hipMemcpyDeviceToDevice, s));
}

// slice of the boundary columns rank (glo + q) computes
static inline void dist_bnd_slice(int nb, int gcnt, int q, int& b0, int& b1) {
  int per = ((nb + gcnt - 1) / gcnt + 127) / 128 * 128;
  b0 = std::min(q * per, nb);
  b1 = std::min(b0 + per, nb);
}

template <class T>
static void dist_allgather_bnd(const DistFront<T>& D, hipStream_t s) {
  if (D.nb == 0) return;
  int mb0, mb1;
  dist_bnd_slice(D.nb, D.gcnt, D.rank - D.glo, mb0, mb1);
  for (int which = 0; which < 2; ++which) {  // column slices are contiguous: no packing; one message per peer and transfer
    T* base = which == 0 ? D.UR : D.SB;
    const size_t ld = which == 0 ? D.ldu : D.lds;
    std::vector<HsPiece> sends, recvs;
    for (int q = 0; q < D.gcnt; ++q) {
      const int r = D.glo + q;
      if (r == D.rank) continue;
      int b0, b1;
      dist_bnd_slice(D.nb, D.gcnt, q, b0, b1);
      if (mb1 > mb0) sends.push_back({r, base + (size_t)mb0 * ld, ld * (mb1 - mb0) * sizeof(T)});
      if (b1 > b0) recvs.push_back({r, base + (size_t)b0 * ld, ld * (b1 - b0) * sizeof(T)});
    }
    D.comm->transfer(sends, recvs, s);
  }
}

// The schedule.  `base` is the one-front Sched of the level (stream s = the handle's stream, s2 = the high-priority side stream).
template <class T>
static void factor_front_dist(Sched<T>& base, const DistFront<T>& D) {
  const int NB = D.NB, ni = D.ni, g = D.gcnt, me = D.rank - D.glo;
  const int nblk = (ni + NB - 1) / NB;
  if (nblk <= 0) return;
  auto mine = [&](int j) { return D.owner(j) == D.rank; };
  int last_packed = -1;  // my latest block column whose message sits in stage_s: its send must be over before the next pack
  Sched<T> mn = base, side = base;
  mn.s2 = nullptr;
  mn.s_la = nullptr;
  side.s = base.s2 ? base.s2 : base.s;
  side.s2 = nullptr;
  side.s_la = nullptr;
  side.hiprio = base.s2 ? 1 : 0;
  hipStream_t s = mn.s, s2 = side.s, sc = D.sc;
  int b0, b1;
  dist_bnd_slice(D.nb, g, me, b0, b1);
  std::vector<hipEvent_t> ev_have(nblk), ev_fact(nblk);
  for (auto& e : ev_have) HS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : ev_fact) HS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t ev_main, ev_iter[3];
  HS_HIP(hipEventCreateWithFlags(&ev_main, hipEventDisableTiming));
  for (auto& e : ev_iter) HS_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  // the front is assembled on s: nothing may arrive in it, and the side stream may not touch it, before that
  HS_HIP(hipEventRecord(ev_main, s));
  HS_HIP(hipStreamWaitEvent(sc, ev_main, 0));
  if (s2 != s) HS_HIP(hipStreamWaitEvent(s2, ev_main, 0));
  if (mine(0)) {
    side.lu_rec(0, NB);
    dist_panel_message(D, 0, D.stage_s, 1, s2);
    last_packed = 0;
    HS_HIP(hipEventRecord(ev_fact[0], s2));
    HS_HIP(hipStreamWaitEvent(sc, ev_fact[0], 0));
  }
  dist_bcast_panel(D, 0, sc);
  HS_HIP(hipEventRecord(ev_have[0], sc));
  for (int j = 0; j < nblk; ++j) {
    const int c0 = j * NB, c1 = c0 + NB, c2 = c1 + NB;
    const bool has_next = j + 1 < nblk;
    if (j >= 3) HS_HIP(hipEventSynchronize(ev_iter[j % 3]));  // bounded host run-ahead (see factor_fronts_lookahead)
    HS_HIP(hipStreamWaitEvent(s, mine(j) ? ev_fact[j] : ev_have[j], 0));
    const bool next_mine = has_next && mine(j + 1);
    if (next_mine) {  // look-ahead: my next block column first, factored on the side stream while the rest is updated
      mn.laswp(HS_MAT_LF, c1, c2, c0, c1);
      mn.trsm_rec(HS_MAT_LF, c0, c1, c1, c2);
      mn.gemm(HS_MAT_LF, HS_MAT_LF, c1, HS_BIG, c1, c2, c0, c1);
      HS_HIP(hipEventRecord(ev_main, s));
      HS_HIP(hipStreamWaitEvent(s2, ev_main, 0));
      side.lu_rec(c1, c2);
      if (last_packed >= 0) HS_HIP(hipStreamWaitEvent(s2, ev_have[last_packed], 0));  // recorded behind that block's fan-out on the comm stream
      dist_panel_message(D, j + 1, D.stage_s, 1, s2);
      last_packed = j + 1;
      HS_HIP(hipEventRecord(ev_fact[j + 1], s2));
      HS_HIP(hipStreamWaitEvent(sc, ev_fact[j + 1], 0));
    }
    if (has_next) {
      dist_bcast_panel(D, j + 1, sc);
      HS_HIP(hipEventRecord(ev_have[j + 1], sc));
      dist_bcast_upper(D, j + 1, sc);  // behind the critical message: it travels while the next block column is being factored
    }
    for (int k = j + (next_mine ? 2 : 1); k < nblk;) {  // my other block columns, one launch group per run of consecutive ones
      if (!mine(k)) {
        ++k;
        continue;
      }
      int ke = k;
      while (ke < nblk && mine(ke)) ++ke;
      const int k0 = k * NB, k1 = ke * NB;
      mn.laswp(HS_MAT_LF, k0, k1, c0, c1);
      mn.trsm_rec(HS_MAT_LF, c0, c1, k0, k1);
      mn.gemm(HS_MAT_LF, HS_MAT_LF, c1, HS_BIG, k0, k1, c0, c1);
      k = ke;
    }
    if (b1 > b0) {  // my slice of the boundary columns
      mn.laswp(HS_MAT_UR, b0, b1, c0, c1);
      mn.trsm_rec(HS_MAT_UR, c0, c1, b0, b1);
      mn.gemm(HS_MAT_UR, HS_MAT_UR, c1, HS_BIG, b0, b1, c0, c1);
    }
    HS_HIP(hipEventRecord(ev_iter[j % 3], s));
  }
  // every transfer of this front has completed before the block columns change again (left swaps)
  HS_HIP(hipEventRecord(ev_main, sc));
  HS_HIP(hipStreamWaitEvent(s, ev_main, 0));
  for (int c0 = 0; c0 + NB < ni; c0 += NB) mn.laswp(HS_MAT_LF, c0, c0 + NB, c0 + NB, HS_BIG);
  if (b1 > b0) mn.gemm(HS_MAT_SB, HS_MAT_UR, 0, HS_BIG, b0, b1, 0, HS_BIG);
  if (D.nb > 0) {
    HS_HIP(hipEventRecord(ev_main, s));
    HS_HIP(hipStreamWaitEvent(sc, ev_main, 0));
    dist_allgather_bnd(D, sc);
    HS_HIP(hipEventRecord(ev_main, sc));
    HS_HIP(hipStreamWaitEvent(s, ev_main, 0));
  }
  HS_HIP(hipStreamSynchronize(s));  // the events below are destroyed; the level's flags are read next anyway
  for (auto& e : ev_have) (void)hipEventDestroy(e);
  for (auto& e : ev_fact) (void)hipEventDestroy(e);
  for (auto& e : ev_iter) (void)hipEventDestroy(e);
  (void)hipEventDestroy(ev_main);
}

// logical OR of one flag over the group [glo, glo+gcnt) (host value in, host value out; synchronises the comm stream)
static int dist_group_or(hs_comm* comm, int* d_flags, int glo, int gcnt, int rank, int flag, hipStream_t sc) {
  std::vector<int> hf(gcnt, 0);
  hf[rank - glo] = flag ? 1 : 0;
  HS_HIP(hipMemcpyAsync(d_flags + (rank - glo), &hf[rank - glo], sizeof(int), hipMemcpyHostToDevice, sc));
  std::vector<HsPiece> sends, recvs;
  for (int r = glo; r < glo + gcnt; ++r) {
    if (r == rank) continue;
    sends.push_back({r, d_flags + (rank - glo), sizeof(int)});
    recvs.push_back({r, d_flags + (r - glo), sizeof(int)});
  }
  comm->transfer(sends, recvs, sc);
  HS_HIP(hipMemcpyAsync(hf.data(), d_flags, sizeof(int) * gcnt, hipMemcpyDeviceToHost, sc));
  HS_HIP(hipStreamSynchronize(sc));
  int any = 0;
  for (int v : hf) any |= v;
  return any;
}
