// hs_symbolic.hip -- the reference's symbolic layer on the host, in C++ (host code only; no device calls).
//
//   parse_elimtree  (src/nesteddissection.jl:105-148)   7-array elimination-tree format -> binary tree
//   symfact!        (src/nesteddissection.jl:29-69)     parents' int/bnd := [left part; right part], local index maps
//   postorder       (src/nesteddissection.jl:73-79)     elimination permutation
//   permuted!       (src/nesteddissection.jl:82-88)     renumber the tree with the inverse permutation
//
// i.e. exactly the host pipeline of the reference's scenario (test/rungmres.jl:15-19), returning the flat post-ordered
// `hs_tree` that hs_factor / hs_analyze take (include/hs_solver.h).  The reference evaluates `findall(in(parent), child)`
// with vector `in` -- O(|parent|*|child|) per node; here membership goes through one scratch map over the DOFs, so the
// whole pass is O(sum of index-set lengths).  Added validation the reference lacks (nesteddissection.jl:63 silently
// drops DOFs no child carries): every DOF of a branch must come from one of its children.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/hs_solver.h"
#include "../../include/hs_symbolic.h"

void hs_set_error(int code, long long info, const char* fmt, ...);

struct hs_symbolic {
  int64_t n = 0;  // number of DOFs
  std::vector<int64_t> perm;  // 1-based: A[perm, perm] is the matrix the tree refers to
  std::vector<int64_t> left, right;
  std::vector<int64_t> int_ptr, int_idx, bnd_ptr, bnd_idx, iloc_ptr, iloc_idx, bloc_ptr, bloc_idx;
};

namespace {
struct SymErr {
  int code;
};
#define SYM_FAIL(code, info, ...)          \
  do {                                     \
    hs_set_error(code, info, __VA_ARGS__); \
    throw SymErr{code};                    \
  } while (0)

struct Node {
  int left = -1, right = -1;
  std::vector<int64_t> in, bd, iloc, bloc;
};
}  // namespace

extern "C" int hs_symbolic_from_elimtree(int64_t nnodes, const int64_t* fathers, const int64_t* lsons, const int64_t* rsons, const int64_t* ninter,
                                         const int64_t* inter, int64_t ld_inter, const int64_t* nbound, const int64_t* bound, int64_t ld_bound,
                                         hs_symbolic** out) {
  if (out) *out = nullptr;
  try {
    if (!out || nnodes <= 0 || !fathers || !lsons || !rsons || !ninter || !nbound) SYM_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: empty elimination tree");
    const int nn = (int)nnodes;
    // ---- parse_elimtree: node table, one root ------------------------------------------------------------------
    int root = -1, nroots = 0;
    for (int i = 0; i < nn; ++i)
      if (fathers[i] == -1) {
        root = i;
        ++nroots;
      }
    if (nroots != 1) SYM_FAIL(HS_ERR_ARGUMENT, nroots, "ArgumentError: found either less than or more than one root.");  // :111
    std::vector<Node> N(nn);
    int64_t maxdof = 0;
    for (int i = 0; i < nn; ++i) {
      const int64_t ls = lsons[i], rs = rsons[i];
      if ((ls != -1 && (ls < 1 || ls > nn)) || (rs != -1 && (rs < 1 || rs > nn))) SYM_FAIL(HS_ERR_ARGUMENT, i + 1, "BoundsError: son of node %d outside 1:%d", i + 1, nn);
      N[i].left = ls == -1 ? -1 : (int)ls - 1;
      N[i].right = rs == -1 ? -1 : (int)rs - 1;
      if (ninter[i] < 0 || ninter[i] > ld_inter || nbound[i] < 0 || nbound[i] > ld_bound)
        SYM_FAIL(HS_ERR_DIMENSION, i + 1, "DimensionMismatch: dimensions inconsistent among inputs (node %d)", i + 1);  // :107
      N[i].in.assign(inter + (size_t)i * ld_inter, inter + (size_t)i * ld_inter + ninter[i]);
      N[i].bd.assign(bound + (size_t)i * ld_bound, bound + (size_t)i * ld_bound + nbound[i]);
      for (int64_t g : N[i].in) {
        if (g < 1) SYM_FAIL(HS_ERR_DIMENSION, i + 1, "BoundsError: DOF id %lld of node %d", (long long)g, i + 1);
        maxdof = std::max(maxdof, g);
      }
      for (int64_t g : N[i].bd) {
        if (g < 1) SYM_FAIL(HS_ERR_DIMENSION, i + 1, "BoundsError: DOF id %lld of node %d", (long long)g, i + 1);
        maxdof = std::max(maxdof, g);
      }
    }
    // ---- post-order (left, right, self), iterative: trees of graded meshes may be deep ----------------------------
    std::vector<int> order;
    order.reserve(nn);
    {
      std::vector<char> visited(nn, 0);
      std::vector<std::pair<int, int>> stack;  // (node, state)
      stack.push_back({root, 0});
      while (!stack.empty()) {
        auto [x, st] = stack.back();
        stack.pop_back();
        if (st == 1) {
          order.push_back(x);
          continue;
        }
        if (visited[x]) SYM_FAIL(HS_ERR_TREE, x + 1, "ArgumentError: node %d is reachable twice (not a tree)", x + 1);
        visited[x] = 1;
        stack.push_back({x, 1});
        if (N[x].right >= 0) stack.push_back({N[x].right, 0});
        if (N[x].left >= 0) stack.push_back({N[x].left, 0});
      }
      if ((int)order.size() != nn) SYM_FAIL(HS_ERR_TREE, (long long)order.size(), "ArgumentError: %d of %d nodes are not reachable from the root", nn - (int)order.size(), nn);
    }
    // ---- symfact! ---------------------------------------------------------------------------------------------------
    std::vector<signed char> mark((size_t)maxdof + 2, 0);
    for (int x : order) {
      Node& P = N[x];
      if (P.left < 0 && P.right < 0) continue;  // leaf: int, bnd as given; its loc is set by its parent
      for (int64_t g : P.in) mark[g] = 1;
      for (int64_t g : P.bd) mark[g] = 2;
      std::vector<int64_t> ni, nb;
      ni.reserve(P.in.size());
      nb.reserve(P.bd.size());
      for (int side = 0; side < 2; ++side) {
        const int c = side == 0 ? P.left : P.right;
        if (c < 0) continue;
        Node& Cn = N[c];
        Cn.iloc.clear();
        Cn.bloc.clear();
        for (size_t q = 0; q < Cn.bd.size(); ++q) {  // findall(in(nd.int), child.bnd) / findall(in(nd.bnd), child.bnd)
          const signed char m = mark[Cn.bd[q]];
          if (m == 1) {
            Cn.iloc.push_back((int64_t)q + 1);
            ni.push_back(Cn.bd[q]);
          } else if (m == 2) {
            Cn.bloc.push_back((int64_t)q + 1);
            nb.push_back(Cn.bd[q]);
          }
        }
      }
      // validation the reference lacks: the children must carry every DOF of the node exactly once
      if (ni.size() != P.in.size() || nb.size() != P.bd.size())
        SYM_FAIL(HS_ERR_DIMENSION, x + 1, "DimensionMismatch: the children of node %d carry (%lld, %lld) of its (%lld, %lld) int/bnd DOFs", x + 1, (long long)ni.size(),
                 (long long)nb.size(), (long long)P.in.size(), (long long)P.bd.size());
      for (int64_t g : P.in) mark[g] = 0;
      for (int64_t g : P.bd) mark[g] = 0;
      P.in.swap(ni);
      P.bd.swap(nb);
    }
    {  // root: nd_loc.int = 1:|bnd|, nd_loc.bnd = [] (:31-32)
      Node& R = N[root];
      R.iloc.resize(R.bd.size());
      for (size_t q = 0; q < R.bd.size(); ++q) R.iloc[q] = (int64_t)q + 1;
      R.bloc.clear();
    }
    // ---- postorder permutation and permuted! --------------------------------------------------------------------------
    hs_symbolic* S = new hs_symbolic();
    for (int x : order) S->perm.insert(S->perm.end(), N[x].in.begin(), N[x].in.end());
    S->perm.insert(S->perm.end(), N[root].bd.begin(), N[root].bd.end());
    S->n = (int64_t)S->perm.size();
    std::vector<int64_t> iperm((size_t)maxdof + 1, 0);
    for (size_t k = 0; k < S->perm.size(); ++k) {
      const int64_t g = S->perm[k];
      if (iperm[g] != 0) {
        delete S;
        SYM_FAIL(HS_ERR_DIMENSION, g, "DimensionMismatch: DOF %lld is eliminated twice", (long long)g);
      }
      iperm[g] = (int64_t)k + 1;
    }
    // ---- flat post-ordered arrays (node ids = post-order positions) ------------------------------------------------------
    std::vector<int> pos(nn, -1);
    for (int k = 0; k < nn; ++k) pos[order[k]] = k;
    S->left.resize(nn);
    S->right.resize(nn);
    S->int_ptr.assign(nn + 1, 0);
    S->bnd_ptr.assign(nn + 1, 0);
    S->iloc_ptr.assign(nn + 1, 0);
    S->bloc_ptr.assign(nn + 1, 0);
    for (int k = 0; k < nn; ++k) {
      const Node& X = N[order[k]];
      S->left[k] = X.left >= 0 ? pos[X.left] : -1;
      S->right[k] = X.right >= 0 ? pos[X.right] : -1;
      S->int_ptr[k + 1] = S->int_ptr[k] + (int64_t)X.in.size();
      S->bnd_ptr[k + 1] = S->bnd_ptr[k] + (int64_t)X.bd.size();
      S->iloc_ptr[k + 1] = S->iloc_ptr[k] + (int64_t)X.iloc.size();
      S->bloc_ptr[k + 1] = S->bloc_ptr[k] + (int64_t)X.bloc.size();
      for (int64_t g : X.in) S->int_idx.push_back(iperm[g]);
      for (int64_t g : X.bd) {
        if (iperm[g] == 0) {
          delete S;
          SYM_FAIL(HS_ERR_DIMENSION, g, "DimensionMismatch: boundary DOF %lld of node %d is never eliminated", (long long)g, order[k] + 1);
        }
        S->bnd_idx.push_back(iperm[g]);
      }
      S->iloc_idx.insert(S->iloc_idx.end(), X.iloc.begin(), X.iloc.end());
      S->bloc_idx.insert(S->bloc_idx.end(), X.bloc.begin(), X.bloc.end());
    }
    *out = S;
    return HS_OK;
  } catch (const SymErr& e) {
    return e.code;
  } catch (const std::bad_alloc&) {
    hs_set_error(HS_ERR_NOMEM, 0, "host allocation failed");
    return HS_ERR_NOMEM;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// Nested dissection of a GENERAL sparse matrix from its graph (no coordinates): the elimination tree in the reference's
// disjoint-ownership form (every DOF in exactly one leaf; bnd(B) = DOFs of B with a neighbour outside B; a parent eliminates
// (bnd(l) | bnd(r)) - bnd(parent); src/nesteddissection.jl:19-21,105-148).  The reference only CONSUMES such trees -- the
// generator behind its .mat files is not part of it.  A set is halved by a breadth-first sweep of its induced subgraph from a
// pseudo-peripheral vertex (first half of the sweep against the rest), until it holds at most nmax DOFs; the result then goes
// through the same pipeline as a tree read from a file (symfact! -> postorder -> permuted!).
// ------------------------------------------------------------------------------------------------------------------------
namespace {
struct GNode {
  int left = -1, right = -1;
  std::vector<int64_t> in, bd;  // 1-based, ascending
};
struct GraphND {
  int64_t n;
  std::vector<int64_t> ap;
  std::vector<int> aj;       // symmetrised pattern without the diagonal, sorted rows
  std::vector<int> deg, tag, inside, seen, queue;
  std::vector<GNode> nodes;  // post-order
  int nmax, stamp = 0, sets = 0;

  // breadth-first order of the vertices of V (all tagged `id`), the component of `start` first, then the others by lowest id
  void sweep(const std::vector<int>& V, int id, int start, std::vector<int>& order) {
    ++stamp;
    order.clear();
    size_t scan = 0;
    auto push = [&](int v) {
      seen[v] = stamp;
      order.push_back(v);
    };
    push(start);
    size_t head = 0;
    while (order.size() < V.size()) {
      if (head == order.size()) {
        while (seen[V[scan]] == stamp) ++scan;
        push(V[scan]);
      }
      const int v = order[head++];
      for (int64_t a = ap[v]; a < ap[v + 1]; ++a) {
        const int w = aj[a];
        if (tag[w] == id && seen[w] != stamp) push(w);
      }
    }
  }
  // V ascending; returns the node index; bnd flags of V are left in `isb` (indexed like V)
  int build(const std::vector<int>& V, std::vector<char>& isb) {
    const int id = ++sets;
    for (int v : V) tag[v] = id;
    isb.assign(V.size(), 0);
    for (size_t k = 0; k < V.size(); ++k) {
      const int v = V[k];
      int in = 0;
      for (int64_t a = ap[v]; a < ap[v + 1]; ++a) in += tag[aj[a]] == id;
      isb[k] = deg[v] > in;
    }
    GNode g;
    if ((int)V.size() <= nmax || V.size() < 2) {
      for (size_t k = 0; k < V.size(); ++k) (isb[k] ? g.bd : g.in).push_back((int64_t)V[k] + 1);
      nodes.push_back(g);
      return (int)nodes.size() - 1;
    }
    std::vector<int> order;
    int start = V[0];
    for (int rep = 0; rep < 2; ++rep) {  // pseudo-peripheral vertex: the last vertex of the first component's sweep, twice
      ++stamp;
      std::vector<int> comp{start};
      seen[start] = stamp;
      for (size_t head = 0; head < comp.size(); ++head) {
        const int v = comp[head];
        for (int64_t a = ap[v]; a < ap[v + 1]; ++a) {
          const int w = aj[a];
          if (tag[w] == id && seen[w] != stamp) {
            seen[w] = stamp;
            comp.push_back(w);
          }
        }
      }
      start = comp.back();
    }
    sweep(V, id, start, order);
    const size_t half = (V.size() + 1) / 2;
    std::vector<int> V1(order.begin(), order.begin() + half), V2(order.begin() + half, order.end());
    std::sort(V1.begin(), V1.end());
    std::sort(V2.begin(), V2.end());
    std::vector<char> b1, b2;
    const int l = build(V1, b1);
    const int r = build(V2, b2);
    // the children re-tagged their vertices; this node's own boundary flags were computed before
    std::vector<char> cb(V.size(), 0);
    {
      size_t i1 = 0, i2 = 0;
      for (size_t k = 0; k < V.size(); ++k) {
        if (i1 < V1.size() && V1[i1] == V[k]) cb[k] = b1[i1++];
        else cb[k] = b2[i2++];
      }
    }
    for (size_t k = 0; k < V.size(); ++k) {
      if (isb[k]) g.bd.push_back((int64_t)V[k] + 1);
      else if (cb[k]) g.in.push_back((int64_t)V[k] + 1);
    }
    g.left = l;
    g.right = r;
    nodes.push_back(g);
    return (int)nodes.size() - 1;
  }
};
}  // namespace

extern "C" int hs_symbolic_from_graph(int64_t n, const int64_t* colptr, const int64_t* rowval, int64_t nmax, hs_symbolic** out) {
  if (out) *out = nullptr;
  try {
    if (!out || n <= 0 || !colptr || !rowval || nmax < 1) SYM_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: hs_symbolic_from_graph needs a pattern and nmax >= 1");
    if (n > 2000000000LL) SYM_FAIL(HS_ERR_ARGUMENT, 0, "ArgumentError: more than 2^31 DOFs");
    GraphND G;
    G.n = n;
    G.nmax = (int)std::min<int64_t>(nmax, n);
    // symmetrised pattern without the diagonal
    std::vector<int64_t> cnt((size_t)n + 1, 0);
    const int64_t nnz = colptr[n] - 1;
    for (int64_t c = 0; c < n; ++c)
      for (int64_t a = colptr[c] - 1; a < colptr[c + 1] - 1; ++a) {
        const int64_t r = rowval[a] - 1;
        if (r < 0 || r >= n) SYM_FAIL(HS_ERR_DIMENSION, a, "BoundsError: row index %lld outside 1:%lld", (long long)(r + 1), (long long)n);
        if (r == c) continue;
        cnt[(size_t)r + 1]++;
        cnt[(size_t)c + 1]++;
      }
    (void)nnz;
    for (int64_t i = 0; i < n; ++i) cnt[(size_t)i + 1] += cnt[(size_t)i];
    std::vector<int> adj((size_t)cnt[(size_t)n]);
    std::vector<int64_t> pos(cnt.begin(), cnt.end() - 1);
    for (int64_t c = 0; c < n; ++c)
      for (int64_t a = colptr[c] - 1; a < colptr[c + 1] - 1; ++a) {
        const int64_t r = rowval[a] - 1;
        if (r == c) continue;
        adj[(size_t)pos[(size_t)r]++] = (int)c;
        adj[(size_t)pos[(size_t)c]++] = (int)r;
      }
    G.ap.assign((size_t)n + 1, 0);
    G.aj.reserve(adj.size());
    for (int64_t v = 0; v < n; ++v) {
      auto b = adj.begin() + cnt[(size_t)v], e = adj.begin() + cnt[(size_t)v + 1];
      std::sort(b, e);
      e = std::unique(b, e);
      G.aj.insert(G.aj.end(), b, e);
      G.ap[(size_t)v + 1] = (int64_t)G.aj.size();
    }
    G.deg.resize((size_t)n);
    for (int64_t v = 0; v < n; ++v) G.deg[(size_t)v] = (int)(G.ap[(size_t)v + 1] - G.ap[(size_t)v]);
    G.tag.assign((size_t)n, 0);
    G.seen.assign((size_t)n, 0);
    std::vector<int> all((size_t)n);
    for (int64_t v = 0; v < n; ++v) all[(size_t)v] = (int)v;
    std::vector<char> isb;
    G.build(all, isb);
    // the 7 arrays of util/read_problem.jl:14-20, nodes in post-order
    const int nn = (int)G.nodes.size();
    std::vector<int64_t> fathers(nn, -1), lsons(nn, -1), rsons(nn, -1), ninter(nn), nbound(nn);
    int64_t mi = 1, mb = 1;
    for (int i = 0; i < nn; ++i) {
      ninter[i] = (int64_t)G.nodes[i].in.size();
      nbound[i] = (int64_t)G.nodes[i].bd.size();
      mi = std::max(mi, ninter[i]);
      mb = std::max(mb, nbound[i]);
      if (G.nodes[i].left >= 0) {
        lsons[i] = G.nodes[i].left + 1;
        rsons[i] = G.nodes[i].right + 1;
        fathers[G.nodes[i].left] = i + 1;
        fathers[G.nodes[i].right] = i + 1;
      }
    }
    std::vector<int64_t> inter((size_t)mi * nn, 0), bound((size_t)mb * nn, 0);
    for (int i = 0; i < nn; ++i) {
      std::copy(G.nodes[i].in.begin(), G.nodes[i].in.end(), inter.begin() + (size_t)mi * i);
      std::copy(G.nodes[i].bd.begin(), G.nodes[i].bd.end(), bound.begin() + (size_t)mb * i);
    }
    return hs_symbolic_from_elimtree(nn, fathers.data(), lsons.data(), rsons.data(), ninter.data(), inter.data(), mi, nbound.data(), bound.data(), mb, out);
  } catch (const SymErr& e) {
    return e.code;
  } catch (const std::bad_alloc&) {
    hs_set_error(HS_ERR_NOMEM, 0, "host allocation failed");
    return HS_ERR_NOMEM;
  }
}

extern "C" int64_t hs_symbolic_size(const hs_symbolic* S) { return S ? S->n : 0; }
extern "C" const int64_t* hs_symbolic_perm(const hs_symbolic* S) { return S ? S->perm.data() : nullptr; }
extern "C" int hs_symbolic_tree(const hs_symbolic* S, hs_tree* t) {
  if (!S || !t) {
    hs_set_error(HS_ERR_ARGUMENT, 0, "ArgumentError: null symbolic handle");
    return HS_ERR_ARGUMENT;
  }
  t->nnodes = (int64_t)S->left.size();
  t->left = S->left.data();
  t->right = S->right.data();
  t->int_ptr = S->int_ptr.data();
  t->int_idx = S->int_idx.data();
  t->bnd_ptr = S->bnd_ptr.data();
  t->bnd_idx = S->bnd_idx.data();
  t->iloc_ptr = S->iloc_ptr.data();
  t->iloc_idx = S->iloc_idx.data();
  t->bloc_ptr = S->bloc_ptr.data();
  t->bloc_idx = S->bloc_idx.data();
  return HS_OK;
}
extern "C" void hs_symbolic_free(hs_symbolic* S) { delete S; }
