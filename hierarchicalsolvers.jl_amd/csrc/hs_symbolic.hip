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
