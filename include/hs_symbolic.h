/* hs_symbolic.h -- C ABI of the symbolic layer (host only): what a Julia host would otherwise compute with
 * parse_elimtree / symfact! / postorder / permuted! (reference src/nesteddissection.jl:29-148, scenario
 * test/rungmres.jl:15-19) before calling factor.  Same error conventions as hs_solver.h (status codes,
 * hs_last_error).  All ids are 1-based Int64 as the Julia host holds them, except node ids inside hs_tree
 * (0-based post-order positions, -1 = no child). */
#ifndef HS_SYMBOLIC_H
#define HS_SYMBOLIC_H
#include <stdint.h>
#include "hs_solver.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hs_symbolic hs_symbolic; /* opaque: owns the permutation and the flat tree arrays */

/* The 7-array elimination-tree format of util/read_problem.jl:14-20 (decoder: nesteddissection.jl:105-148):
 * node ids 1-based, -1 = none, exactly one node with fathers == -1; column i (0-based i-1) of inter / bound
 * (column-major, leading dimensions ld_inter / ld_bound) holds the first ninter[i] / nbound[i] DOF ids.
 * Runs parse_elimtree -> symfact! -> postorder -> permuted!(nd, invperm(perm)).
 * Errors: HS_ERR_ARGUMENT (no / several roots: ArgumentError of :111), HS_ERR_DIMENSION (inconsistent inputs,
 * :107; a branch whose children do not carry all of its DOFs -- not checked by the reference), HS_ERR_TREE. */
int hs_symbolic_from_elimtree(int64_t nnodes, const int64_t* fathers, const int64_t* lsons, const int64_t* rsons,
                              const int64_t* ninter, const int64_t* inter, int64_t ld_inter,
                              const int64_t* nbound, const int64_t* bound, int64_t ld_bound, hs_symbolic** out);

/* The same from the GRAPH of a general sparse matrix alone (1-based CSC pattern colptr[n+1], rowval[nnz]; the values are not needed): nested
 * dissection by recursive breadth-first bisection into the reference's disjoint-ownership tree (every DOF in exactly one leaf of at most
 * nmax DOFs; bnd(B) = DOFs of B with a neighbour outside B), then symfact! -> postorder -> permuted!.  The reference only consumes such trees
 * (src/nesteddissection.jl:105-148); the generator behind its .mat files is not part of it. */
int hs_symbolic_from_graph(int64_t n, const int64_t* colptr, const int64_t* rowval, int64_t nmax, hs_symbolic** out);

int64_t hs_symbolic_size(const hs_symbolic* S);        /* number of DOFs n */
const int64_t* hs_symbolic_perm(const hs_symbolic* S); /* n entries, 1-based: factor A[perm, perm] (postorder, :73-79) */
/* the flat post-ordered tree in the PERMUTED numbering (what symfact! + permuted! leave in nd, nd_loc): valid for
 * hs_factor_* / hs_analyze as long as S lives */
int hs_symbolic_tree(const hs_symbolic* S, hs_tree* out);
void hs_symbolic_free(hs_symbolic* S);

#ifdef __cplusplus
}
#endif
#endif
