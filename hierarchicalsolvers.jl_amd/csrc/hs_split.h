// hs_split.h -- the interior block of a large compressed front is eliminated in SLICES (host-side tree rewrite).
//
// Reference: a compressed branch stores D = Aii as a 2x2 `BlockFactorization` (src/blockmatrix.jl:106-130:
// A11, A12, A21 and the Schur complement S22 = A22 - A21*(A11 \ A12)) of HSS blocks, i.e. D is never factored as one
// dense matrix.  The counterpart here needs no new kernel: a front X with interior DOFs int = [s_1; s_2; ...; s_k] is
// replaced, in the elimination tree, by the chain
//     X^(1): eliminates s_1, boundary [s_2; ...; s_k; bnd]   (children: X's children -- the SAME assembled front,
//                                                             only the int/bnd border moved)
//     X^(j): eliminates s_j, boundary [s_j+1; ...; bnd]      (single child X^(j-1); its front is that child's Schur
//                                                             complement, nothing else is assembled)
// Each X^(j) is an ordinary front: with compression its off-diagonal blocks [A_{j,j+1..k}  A_{j,bnd}] are low rank
// (hs_compress.h) and its Schur update is a rank-r GEMM.  The result is the block LU of D with low-rank off-diagonal
// panels -- the `blockfactor` structure applied k-1 times -- at (2/3)*ni^3/k^2 LU flops instead of (2/3)*ni^3, and the
// flops that remain are GEMMs, not the latency-bound panel chain of one huge LU.  Pivoting stays inside a slice, exactly
// as `\` on A11 and on S22 pivots inside those blocks in the reference.  In exact arithmetic nothing changes.
//
// Enabled by hs_options.split (slice width in units of 256 columns, 0 = off) for fronts at levels <= swlevel with
// ni >= 2 slices; single-rank factorizations only (the rank cut of hs_analyze counts levels of the original tree).
#pragma once
#include <algorithm>
#include <cstdint>
#include <vector>

struct SplitTree {
  bool active = false;
  hs_tree view;
  std::vector<int64_t> left, right, int_ptr, int_idx, bnd_ptr, bnd_idx, iloc_ptr, iloc_idx, bloc_ptr, bloc_idx;
  std::vector<int> kind;            // 0 ordinary node, 1 first slice of a chain, 2 later slice
  std::vector<int> user;            // user (post-order) node id behind each internal node
  std::vector<int> oni, oni1, onb1; // kind 1: split points of the ORIGINAL front (child -> front maps, sides of the gather)
  std::vector<char> cflag;          // compression flag of each internal node
  std::vector<std::vector<int>> newpos;  // kind 1 with a re-ordered interior: position in the slice order of each ORIGINAL int position
  // per user node
  std::vector<int> last_of_user;    // internal id of the node that carries the user's bnd (the last slice)
  std::vector<int> first_of_user;   // internal id of the first slice
  std::vector<int> u_ni, u_nb, u_level;
};

// compression_flag of factorization.jl:15 with swlevel resolved as in factorization.jl:8
static inline bool hs_compression_flag(int level, int ni, int nb, bool leaf, int64_t swlevel, int64_t swsize) {
  return (level <= swlevel) && (nb >= swsize) && nb > 0 && ni > 0 && !leaf;  // a compressed LEAF keeps dense L, R (factorization.jl:45-59)
}

// Fills `st` from the user's tree.  With splitting off (or nothing to split) st.active stays false and the caller uses `tr`.
// `colptr` / `rowval` (1-based CSC pattern of A, may be null): used to order the interior DOFs of a sliced front so that a
// slice is a compact patch of the separator.  int = [int1; int2] lists the two grid layers of the separator one after the
// other; they are coupled by an identity-like block of A -- sparse but of FULL rank -- so a slice of int1 alone has full-rank
// coupling to "the rest".  Interleaved (every DOF of int1 followed by its neighbours in int2) a slice holds both layers of a
// patch and couples to the rest only through the patch boundary: measured on Poisson 32^3, slices of 256 of the 2048 root
// DOFs have rank 256 in the natural order and 21-55 (tol 1e-2) / 60-128 (tol 1e-6) interleaved.
static void make_split_tree(const hs_tree* tr, const hs_options& opts, int nranks, SplitTree& st, int64_t n = 0, const int64_t* colptr = nullptr,
                            const int64_t* rowval = nullptr) {
  st = SplitTree();
  const int slice = (int)opts.split * 256;
  if (!tr || tr->nnodes <= 0 || slice <= 0 || nranks != 1) return;
  const int nn = (int)tr->nnodes;
  auto len = [](const int64_t* p, int i) { return (int)(p[i + 1] - p[i]); };
  // levels of the user's tree (root = 1); the caller's build_plan validates the structure, here only enough not to crash
  std::vector<int> parent(nn, -1), level(nn, 0);
  for (int i = 0; i < nn; ++i) {
    const int64_t l = tr->left[i], r = tr->right[i];
    if (l >= i || r >= i || l < -1 || r < -1) return;  // not post-ordered: let build_plan report it
    if (l >= 0) parent[l] = i;
    if (r >= 0) parent[r] = i;
  }
  int maxlevel = 1;
  level[nn - 1] = 1;
  for (int i = nn - 2; i >= 0; --i) {
    if (parent[i] < 0) return;
    level[i] = level[parent[i]] + 1;
    maxlevel = std::max(maxlevel, level[i]);
  }
  const int64_t swlevel = opts.swlevel < 0 ? std::max<int64_t>(maxlevel + opts.swlevel, 0) : opts.swlevel;
  bool any = false;
  for (int i = 0; i < nn; ++i) {
    const bool leaf = tr->left[i] < 0 && tr->right[i] < 0;
    if (!leaf && level[i] <= swlevel && len(tr->int_ptr, i) >= 2 * slice) any = true;
  }
  if (!any) return;

  st.last_of_user.assign(nn, -1);
  st.first_of_user.assign(nn, -1);
  st.u_ni.resize(nn);
  st.u_nb.resize(nn);
  st.u_level = level;
  st.int_ptr.push_back(0);
  st.bnd_ptr.push_back(0);
  st.iloc_ptr.push_back(0);
  st.bloc_ptr.push_back(0);
  std::vector<int> where;  // DOF (0-based) -> position in int2 of the node being split, -1 elsewhere
  if (colptr && rowval && n > 0) where.assign((size_t)n, -1);
  auto emit = [&](int user, int kind, int64_t l, int64_t r, const int64_t* ib, const int64_t* ie, const int64_t* bb0, const int64_t* be0, const int64_t* bb1,
                  const int64_t* be1, int oni, int oni1, int onb1, bool cflag) {
    const int id = (int)st.left.size();
    st.left.push_back(l);
    st.right.push_back(r);
    st.int_idx.insert(st.int_idx.end(), ib, ie);
    st.bnd_idx.insert(st.bnd_idx.end(), bb0, be0);
    st.bnd_idx.insert(st.bnd_idx.end(), bb1, be1);
    st.int_ptr.push_back((int64_t)st.int_idx.size());
    st.bnd_ptr.push_back((int64_t)st.bnd_idx.size());
    st.kind.push_back(kind);
    st.user.push_back(user);
    st.oni.push_back(oni);
    st.oni1.push_back(oni1);
    st.onb1.push_back(onb1);
    st.cflag.push_back(cflag ? 1 : 0);
    st.newpos.emplace_back();
    return id;
  };
  for (int i = 0; i < nn; ++i) {
    const int ni = len(tr->int_ptr, i), nb = len(tr->bnd_ptr, i);
    st.u_ni[i] = ni;
    st.u_nb[i] = nb;
    const bool leaf = tr->left[i] < 0 && tr->right[i] < 0;
    const int64_t* I = tr->int_idx + tr->int_ptr[i];
    const int64_t* B = tr->bnd_idx + tr->bnd_ptr[i];
    const int64_t l = tr->left[i] >= 0 ? st.last_of_user[tr->left[i]] : -1, r = tr->right[i] >= 0 ? st.last_of_user[tr->right[i]] : -1;
    const bool split = !leaf && level[i] <= swlevel && ni >= 2 * slice && tr->left[i] >= 0 && tr->right[i] >= 0;
    int last;
    if (!split) {
      last = emit(i, 0, l, r, I, I + ni, B, B + nb, B, B, ni, 0, 0, hs_compression_flag(level[i], ni, nb, leaf, swlevel, opts.swsize));
      st.first_of_user[i] = last;
    } else {
      const int k = std::max(2, ni / slice);
      const int q = ((ni + k - 1) / k + 31) / 32 * 32;  // slice width, a multiple of the panel width
      const int oni1 = len(tr->iloc_ptr, (int)tr->left[i]), onb1 = len(tr->bloc_ptr, (int)tr->left[i]);
      // slice order of the interior DOFs: every DOF of int1 followed by its not yet placed neighbours in int2
      std::vector<int64_t> pint;
      std::vector<int> npos;
      if (!where.empty() && oni1 > 0 && oni1 < ni) {
        bool ok = true;
        for (int e = 0; e < ni; ++e) ok = ok && I[e] >= 1 && I[e] <= n;
        if (ok) {
          for (int e = oni1; e < ni; ++e) where[I[e] - 1] = e;
          pint.reserve(ni);
          npos.assign(ni, -1);
          for (int e = 0; e < oni1; ++e) {
            npos[e] = (int)pint.size();
            pint.push_back(I[e]);
            const int64_t g = I[e] - 1;
            for (int64_t a = colptr[g] - 1; a < colptr[g + 1] - 1; ++a) {
              const int64_t rr = rowval[a] - 1;
              if (rr < 0 || rr >= n) continue;
              const int q = where[rr];
              if (q >= 0 && npos[q] < 0) {
                npos[q] = (int)pint.size();
                pint.push_back(I[q]);
              }
            }
          }
          for (int e = oni1; e < ni; ++e) {
            if (npos[e] < 0) {
              npos[e] = (int)pint.size();
              pint.push_back(I[e]);
            }
            where[I[e] - 1] = -1;
          }
        }
      }
      const int64_t* J = pint.empty() ? I : pint.data();  // interior DOFs in slice order
      int off = 0, prev = -1;
      last = -1;
      for (int j = 0; off < ni; ++j) {
        const int w = std::min(q, ni - off);
        const int rest = ni - off - w;  // interior DOFs left for the later slices: they lead this slice's boundary
        const bool cf = (rest + nb) >= opts.swsize && (rest + nb) > 0;
        if (j == 0) {
          last = emit(i, 1, l, r, J, J + w, J + w, J + ni, B, B + nb, ni, oni1, onb1, cf);
          st.first_of_user[i] = last;
          st.newpos[last] = npos;
        } else {
          // the previous slice relates to this one: its first w boundary DOFs are this slice's interior, the rest its boundary
          for (int e = 0; e < w; ++e) st.iloc_idx.push_back(e + 1);
          st.iloc_ptr.push_back((int64_t)st.iloc_idx.size());
          const int pb = (ni - off) + nb;  // boundary length of the previous slice
          for (int e = w; e < pb; ++e) st.bloc_idx.push_back(e + 1);
          st.bloc_ptr.push_back((int64_t)st.bloc_idx.size());
          last = emit(i, 2, prev, -1, J + off, J + off + w, J + off + w, J + ni, B, B + nb, w, w, rest + nb, cf);
        }
        prev = last;
        off += w;
      }
    }
    // the node that carries the user's bnd relates to the user's parent exactly as the user's node did
    st.iloc_idx.insert(st.iloc_idx.end(), tr->iloc_idx + tr->iloc_ptr[i], tr->iloc_idx + tr->iloc_ptr[i + 1]);
    st.iloc_ptr.push_back((int64_t)st.iloc_idx.size());
    st.bloc_idx.insert(st.bloc_idx.end(), tr->bloc_idx + tr->bloc_ptr[i], tr->bloc_idx + tr->bloc_ptr[i + 1]);
    st.bloc_ptr.push_back((int64_t)st.bloc_idx.size());
    st.last_of_user[i] = last;
  }
  st.view.nnodes = (int64_t)st.left.size();
  st.view.left = st.left.data();
  st.view.right = st.right.data();
  st.view.int_ptr = st.int_ptr.data();
  st.view.int_idx = st.int_idx.data();
  st.view.bnd_ptr = st.bnd_ptr.data();
  st.view.bnd_idx = st.bnd_idx.data();
  st.view.iloc_ptr = st.iloc_ptr.data();
  st.view.iloc_idx = st.iloc_idx.data();
  st.view.bloc_ptr = st.bloc_ptr.data();
  st.view.bloc_idx = st.bloc_idx.data();
  st.active = true;
}
