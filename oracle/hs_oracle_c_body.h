/* Body of the C restatement, compiled twice by hs_oracle_c.c: T = double (suffix _d) and T = double _Complex (suffix _z).
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see hs_oracle_c.c). */

/* ---- dense kernels: the BLAS / LAPACK the host has (function pointers), or the plain loops below ---------------------------------- */

/* C = beta*C + alpha*A*B, column-major, no transposes (everything in the reference's dense path is a plain product) */
static void NAME(gemm)(const hsc_blas* bl, int m, int n, int k, T alpha, const T* A, int lda, const T* B, int ldb, T beta, T* C, int ldc) {
  if (m <= 0 || n <= 0) return;
  g_flops += FLOPMUL * 2.0 * (double)m * (double)n * (double)k;
  if (k <= 0) {
    for (int j = 0; j < n; ++j)
      for (int i = 0; i < m; ++i) C[i + (size_t)j * ldc] *= beta;
    return;
  }
  if (bl && bl->BLASF(gemm)) {
    char nn = 'N';
    bl->BLASF(gemm)(&nn, &nn, &m, &n, &k, &alpha, (T*)A, &lda, (T*)B, &ldb, &beta, C, &ldc);
    return;
  }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < n; ++j) {
    T* c = C + (size_t)j * ldc;
    for (int i = 0; i < m; ++i) c[i] *= beta;
    for (int p = 0; p < k; ++p) {
      const T b = alpha * B[p + (size_t)j * ldb];
      const T* a = A + (size_t)p * lda;
      for (int i = 0; i < m; ++i) c[i] += a[i] * b;
    }
  }
}

/* LU with partial pivoting in place (what Julia's `\` does to a square Matrix before it solves: LinearAlgebra.lu) */
static int NAME(getrf)(const hsc_blas* bl, int n, T* A, int lda, int* ipiv) {
  g_flops += FLOPMUL * (2.0 / 3.0) * (double)n * (double)n * (double)n;
  g_getrf += 1;
  if (n <= 0) return 0;
  if (bl && bl->BLASF(getrf)) {
    int info = 0;
    bl->BLASF(getrf)(&n, &n, A, &lda, ipiv, &info);
    return info;
  }
  enum { NB = 48 };
  for (int k0 = 0; k0 < n; k0 += NB) {
    const int kb = n - k0 < NB ? n - k0 : NB;
    for (int k = k0; k < k0 + kb; ++k) { /* panel, unblocked */
      int p = k;
      double best = ABSF(A[k + (size_t)k * lda]);
      for (int i = k + 1; i < n; ++i) {
        const double v = ABSF(A[i + (size_t)k * lda]);
        if (v > best) { best = v; p = i; }
      }
      ipiv[k] = p + 1;
      if (best == 0.0) return k + 1;
      if (p != k)
        for (int j = 0; j < n; ++j) {
          const T t = A[k + (size_t)j * lda];
          A[k + (size_t)j * lda] = A[p + (size_t)j * lda];
          A[p + (size_t)j * lda] = t;
        }
      const T inv = (T)1.0 / A[k + (size_t)k * lda];
      for (int i = k + 1; i < n; ++i) A[i + (size_t)k * lda] *= inv;
      for (int j = k + 1; j < k0 + kb; ++j) {
        const T u = A[k + (size_t)j * lda];
        for (int i = k + 1; i < n; ++i) A[i + (size_t)j * lda] -= A[i + (size_t)k * lda] * u;
      }
    }
    const int r0 = k0 + kb;
    if (r0 >= n) break;
    /* U12 = L11^-1 * A12 */
#pragma omp parallel for schedule(static)
    for (int j = r0; j < n; ++j)
      for (int k = k0; k < r0; ++k) {
        const T u = A[k + (size_t)j * lda];
        for (int i = k + 1; i < r0; ++i) A[i + (size_t)j * lda] -= A[i + (size_t)k * lda] * u;
      }
    /* A22 -= L21 * U12 */
    const double keep = g_flops;
    NAME(gemm)(NULL, n - r0, n - r0, kb, (T)-1.0, A + r0 + (size_t)k0 * lda, lda, A + k0 + (size_t)r0 * lda, lda, (T)1.0, A + r0 + (size_t)r0 * lda, lda);
    g_flops = keep; /* counted once, above */
  }
  return 0;
}

static void NAME(getrs)(const hsc_blas* bl, int n, int nrhs, const T* LU, int lda, const int* ipiv, T* B, int ldb) {
  g_flops += FLOPMUL * 2.0 * (double)n * (double)n * (double)nrhs;
  if (n <= 0 || nrhs <= 0) return;
  if (bl && bl->BLASF(getrs)) {
    char nn = 'N';
    int info = 0;
    bl->BLASF(getrs)(&nn, &n, &nrhs, (T*)LU, &lda, (int*)ipiv, B, &ldb, &info);
    return;
  }
#pragma omp parallel for schedule(static)
  for (int j = 0; j < nrhs; ++j) {
    T* b = B + (size_t)j * ldb;
    for (int k = 0; k < n; ++k) {
      const int p = ipiv[k] - 1;
      if (p != k) { const T t = b[k]; b[k] = b[p]; b[p] = t; }
    }
    for (int k = 0; k < n; ++k) {
      const T v = b[k];
      for (int i = k + 1; i < n; ++i) b[i] -= LU[i + (size_t)k * lda] * v;
    }
    for (int k = n - 1; k >= 0; --k) {
      b[k] /= LU[k + (size_t)k * lda];
      const T v = b[k];
      for (int i = 0; i < k; ++i) b[i] -= LU[i + (size_t)k * lda] * v;
    }
  }
}

/* ---- small helpers ------------------------------------------------------------------------------------------------------------ */
static T* NAME(newmat)(int m, int n) {
  const size_t el = (size_t)(m > 0 ? m : 1) * (size_t)(n > 0 ? n : 1);
  T* p = (T*)malloc(el * sizeof(T));
  if (!p) { fprintf(stderr, "hs_oracle_c: out of memory (%d x %d)\n", m, n); abort(); }
  g_bytes += (double)el * sizeof(T);
  return p;
}
static T* NAME(copymat)(const T* A, int lda, int m, int n) {
  T* C = NAME(newmat)(m, n);
  for (int j = 0; j < n; ++j) memcpy(C + (size_t)j * m, A + (size_t)j * lda, sizeof(T) * (size_t)m);
  return C;
}
/* X = A \ B as Julia evaluates it for a square Matrix A: LU of a copy of A, then the two triangular solves (a NEW LU at every call:
 * the reference never keeps a factorization object -- blockmatrix.jl:119,137-143,160-171 all spell `A.A11 \ ...`) */
static T* NAME(ldiv_new)(const hsc_blas* bl, const T* A, int lda, int n, const T* B, int ldb, int nrhs) {
  T* X = NAME(copymat)(B, ldb, n, nrhs);
  if (n == 0 || nrhs == 0) return X;
  T* LU = NAME(copymat)(A, lda, n, n);
  int* ipiv = (int*)malloc(sizeof(int) * (size_t)n);
  if (NAME(getrf)(bl, n, LU, n, ipiv) != 0) g_singular += 1;
  NAME(getrs)(bl, n, nrhs, LU, n, ipiv, X, n);
  free(ipiv);
  free(LU);
  return X;
}
/* X = B / A (B is m x n, A n x n) = (A^T \ B^T)^T, as Julia's `/` does it */
static T* NAME(rdiv_new)(const hsc_blas* bl, const T* B, int ldb, int m, const T* A, int lda, int n) {
  T* X = NAME(newmat)(m, n);
  if (n == 0 || m == 0) return X;
  T* At = NAME(newmat)(n, n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) At[i + (size_t)j * n] = A[j + (size_t)i * lda];
  T* Bt = NAME(newmat)(n, m);
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < n; ++i) Bt[i + (size_t)j * n] = B[j + (size_t)i * ldb];
  int* ipiv = (int*)malloc(sizeof(int) * (size_t)n);
  if (NAME(getrf)(bl, n, At, n, ipiv) != 0) g_singular += 1;
  NAME(getrs)(bl, n, m, At, n, ipiv, Bt, n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) X[i + (size_t)j * m] = Bt[j + (size_t)i * n];
  free(ipiv);
  free(At);
  free(Bt);
  return X;
}

/* Matrix(view(A, Ix, J)) of the CSC matrix (factorization.jl:33-40,118-121): `pos` is a scratch map of n entries, all -1 on entry and exit */
static T* NAME(gather)(const NAME(Csc)* A, const int64_t* Ix, int nIx, const int64_t* J, int nJ, int64_t* pos) {
  T* out = NAME(newmat)(nIx, nJ);
  memset(out, 0, sizeof(T) * (size_t)(nIx > 0 ? nIx : 1) * (size_t)(nJ > 0 ? nJ : 1));
  for (int i = 0; i < nIx; ++i) pos[Ix[i]] = i;
  for (int j = 0; j < nJ; ++j)
    for (int64_t q = A->colptr[J[j]]; q < A->colptr[J[j] + 1]; ++q) {
      const int64_t r = pos[A->rowidx[q]];
      if (r >= 0) out[r + (size_t)j * nIx] = A->vals[q];
    }
  for (int i = 0; i < nIx; ++i) pos[Ix[i]] = -1;
  return out;
}

/* ---- BlockMatrix / BlockFactorization (src/blockmatrix.jl) -------------------------------------------------------------------- */
typedef struct {
  T *A11, *A12, *A21, *A22; /* (m1 x n1), (m1 x n2), (m2 x n1), (m2 x n2), tight leading dimensions */
  int m1, m2, n1, n2;
} NAME(Blk);

static void NAME(blk_free)(NAME(Blk)* B) {
  free(B->A11); free(B->A12); free(B->A21); free(B->A22);
  B->A11 = B->A12 = B->A21 = B->A22 = NULL;
}
/* Matrix(B) (blockmatrix.jl:67-75) */
static T* NAME(blk_dense)(const NAME(Blk)* B) {
  const int m = B->m1 + B->m2, n = B->n1 + B->n2;
  T* D = NAME(newmat)(m, n);
  for (int j = 0; j < B->n1; ++j) {
    memcpy(D + (size_t)j * m, B->A11 + (size_t)j * B->m1, sizeof(T) * (size_t)B->m1);
    memcpy(D + B->m1 + (size_t)j * m, B->A21 + (size_t)j * B->m2, sizeof(T) * (size_t)B->m2);
  }
  for (int j = 0; j < B->n2; ++j) {
    memcpy(D + (size_t)(B->n1 + j) * m, B->A12 + (size_t)j * B->m1, sizeof(T) * (size_t)B->m1);
    memcpy(D + B->m1 + (size_t)(B->n1 + j) * m, B->A22 + (size_t)j * B->m2, sizeof(T) * (size_t)B->m2);
  }
  return D;
}
/* blockfactor (blockmatrix.jl:115-120): S22 = A22 - A21 * (A11 \ A12); the blocks A11, A12, A21 are kept AS THEY ARE */
static void NAME(blockfactor)(const hsc_blas* bl, NAME(Blk)* A) {
  T* X = NAME(ldiv_new)(bl, A->A11, A->m1, A->m1, A->A12, A->m1, A->n2);
  NAME(gemm)(bl, A->m2, A->n2, A->n1, (T)-1.0, A->A21, A->m2, X, A->m1, (T)1.0, A->A22, A->m2);
  free(X);
}
/* y = F \ b for a vector block (blockldiv!, blockmatrix.jl:134-144): rows [0, n1) and [n1, n1+n2) of B (ld ldb), in place */
static void NAME(blockldiv_inplace)(const hsc_blas* bl, const NAME(Blk)* F, T* B, int ldb, int nrhs) {
  const int n1 = F->m1, n2 = F->m2;
  T* Y1 = NAME(ldiv_new)(bl, F->A11, n1, n1, B, ldb, nrhs);                                  /* :138 */
  NAME(gemm)(bl, n2, nrhs, n1, (T)-1.0, F->A21, n2, Y1, n1, (T)1.0, B + n1, ldb);             /* :139 */
  T* Y2 = NAME(ldiv_new)(bl, F->A22, n2, n2, B + n1, ldb, nrhs);                              /* :140 */
  T* W = NAME(newmat)(n1, nrhs);
  NAME(gemm)(bl, n1, nrhs, n2, (T)1.0, F->A12, n1, Y2, n2, (T)0.0, W, n1);
  T* V = NAME(ldiv_new)(bl, F->A11, n1, n1, W, n1, nrhs);                                     /* :141 */
  for (int j = 0; j < nrhs; ++j) {
    for (int i = 0; i < n1; ++i) B[i + (size_t)j * ldb] = Y1[i + (size_t)j * n1] - V[i + (size_t)j * n1];
    for (int i = 0; i < n2; ++i) B[n1 + i + (size_t)j * ldb] = Y2[i + (size_t)j * n2];
  }
  free(Y1); free(Y2); free(W); free(V);
}
/* R = F \ B for a BlockMatrix B (blockmatrix.jl:159-172): eight solves against A11 / A22, each with its own LU */
static NAME(Blk) NAME(blockldiv)(const hsc_blas* bl, const NAME(Blk)* F, const NAME(Blk)* B) {
  const int n1 = F->m1, n2 = F->m2, p1 = B->n1, p2 = B->n2;
  NAME(Blk) R = {NULL, NULL, NULL, NULL, n1, n2, p1, p2};
  T* B11 = NAME(ldiv_new)(bl, F->A11, n1, n1, B->A11, n1, p1);                                /* :161 */
  T* t21 = NAME(copymat)(B->A21, n2, n2, p1);
  NAME(gemm)(bl, n2, p1, n1, (T)-1.0, F->A21, n2, B11, n1, (T)1.0, t21, n2);                   /* :162 */
  R.A21 = NAME(ldiv_new)(bl, F->A22, n2, n2, t21, n2, p1);                                    /* :163 */
  free(t21);
  T* w = NAME(newmat)(n1, p1);
  NAME(gemm)(bl, n1, p1, n2, (T)1.0, F->A12, n1, R.A21, n2, (T)0.0, w, n1);
  T* v = NAME(ldiv_new)(bl, F->A11, n1, n1, w, n1, p1);                                       /* :164 */
  for (size_t e = 0; e < (size_t)n1 * p1; ++e) B11[e] -= v[e];
  R.A11 = B11;
  free(w); free(v);
  T* B12 = NAME(ldiv_new)(bl, F->A11, n1, n1, B->A12, n1, p2);                                /* :166 */
  T* t22 = NAME(copymat)(B->A22, n2, n2, p2);
  NAME(gemm)(bl, n2, p2, n1, (T)-1.0, F->A21, n2, B12, n1, (T)1.0, t22, n2);                   /* :167 */
  R.A22 = NAME(ldiv_new)(bl, F->A22, n2, n2, t22, n2, p2);                                    /* :168 */
  free(t22);
  w = NAME(newmat)(n1, p2);
  NAME(gemm)(bl, n1, p2, n2, (T)1.0, F->A12, n1, R.A22, n2, (T)0.0, w, n1);
  v = NAME(ldiv_new)(bl, F->A11, n1, n1, w, n1, p2);                                          /* :169 */
  for (size_t e = 0; e < (size_t)n1 * p2; ++e) B12[e] -= v[e];
  R.A12 = B12;
  free(w); free(v);
  return R;
}
/* L = B / F for a BlockMatrix B (blockmatrix.jl:174-187) */
static NAME(Blk) NAME(blockrdiv)(const hsc_blas* bl, const NAME(Blk)* B, const NAME(Blk)* F) {
  const int n1 = F->m1, n2 = F->m2, q1 = B->m1, q2 = B->m2;
  NAME(Blk) L = {NULL, NULL, NULL, NULL, q1, q2, n1, n2};
  T* B11 = NAME(rdiv_new)(bl, B->A11, q1, q1, F->A11, n1, n1);                                /* :176 */
  T* t12 = NAME(copymat)(B->A12, q1, q1, n2);
  NAME(gemm)(bl, q1, n2, n1, (T)-1.0, B11, q1, F->A12, n1, (T)1.0, t12, q1);                   /* :177 */
  L.A12 = NAME(rdiv_new)(bl, t12, q1, q1, F->A22, n2, n2);                                    /* :178 */
  free(t12);
  T* w = NAME(newmat)(q1, n1);
  NAME(gemm)(bl, q1, n1, n2, (T)1.0, L.A12, q1, F->A21, n2, (T)0.0, w, q1);
  T* v = NAME(rdiv_new)(bl, w, q1, q1, F->A11, n1, n1);                                       /* :179 */
  for (size_t e = 0; e < (size_t)q1 * n1; ++e) B11[e] -= v[e];
  L.A11 = B11;
  free(w); free(v);
  T* B21 = NAME(rdiv_new)(bl, B->A21, q2, q2, F->A11, n1, n1);                                /* :181 */
  T* t22 = NAME(copymat)(B->A22, q2, q2, n2);
  NAME(gemm)(bl, q2, n2, n1, (T)-1.0, B21, q2, F->A12, n1, (T)1.0, t22, q2);                   /* :182 */
  L.A22 = NAME(rdiv_new)(bl, t22, q2, q2, F->A22, n2, n2);                                    /* :183 */
  free(t22);
  w = NAME(newmat)(q2, n1);
  NAME(gemm)(bl, q2, n1, n2, (T)1.0, L.A22, q2, F->A21, n2, (T)0.0, w, q2);
  v = NAME(rdiv_new)(bl, w, q2, q2, F->A11, n1, n1);                                          /* :184 */
  for (size_t e = 0; e < (size_t)q2 * n1; ++e) B21[e] -= v[e];
  L.A21 = B21;
  free(w); free(v);
  return L;
}

/* ---- FactorNode (src/factornode.jl:7-39) ---------------------------------------------------------------------------------------- */
typedef struct {
  int ni, nb, left, right;
  T* D;           /* leaf: Matrix(A[int, int]) (factorization.jl:33), NOT factored: `F.D \ rhs` factors it at every solve (factornode.jl:96) */
  NAME(Blk) DB;   /* branch: the BlockFactorization (blockmatrix.jl:106-108) */
  T *L, *R, *S;   /* nb x ni, ni x nb, nb x nb (S permuted: the parent's interior part first, factorization.jl:41,74) */
} NAME(FNode);

/* S[perm, perm] with perm = [int_loc; bnd_loc] (factorization.jl:41,74) */
static T* NAME(permuted)(const T* S, int nb, const hsc_tree* t, int i) {
  const int64_t *li = t->li_idx + t->li_ptr[i], *lb = t->lb_idx + t->lb_ptr[i];
  const int nli = (int)(t->li_ptr[i + 1] - t->li_ptr[i]), nlb = (int)(t->lb_ptr[i + 1] - t->lb_ptr[i]);
  if (nli + nlb != nb) { /* the root: nb = 0 and empty lists; any other mismatch is an input error */
    T* C = NAME(copymat)(S, nb, nb, nb);
    return C;
  }
  int64_t* perm = (int64_t*)malloc(sizeof(int64_t) * (size_t)(nb > 0 ? nb : 1));
  for (int k = 0; k < nli; ++k) perm[k] = li[k];
  for (int k = 0; k < nlb; ++k) perm[nli + k] = lb[k];
  T* P = NAME(newmat)(nb, nb);
  for (int j = 0; j < nb; ++j)
    for (int r = 0; r < nb; ++r) P[r + (size_t)j * nb] = S[perm[r] + (size_t)perm[j] * nb];
  free(perm);
  return P;
}

/* _factor_leaf (factorization.jl:30-42) */
static void NAME(factor_leaf)(const hsc_blas* bl, const NAME(Csc)* A, const hsc_tree* t, int i, NAME(FNode)* F, int64_t* pos) {
  const int64_t *Ix = t->int_idx + t->int_ptr[i], *B = t->bnd_idx + t->bnd_ptr[i];
  const int ni = F[i].ni, nb = F[i].nb;
  T* D = NAME(gather)(A, Ix, ni, Ix, ni, pos);                 /* :33 */
  T* Abi = NAME(gather)(A, B, nb, Ix, ni, pos);               /* :34 */
  F[i].L = NAME(rdiv_new)(bl, Abi, nb, nb, D, ni, ni);       /* :36  L = Abi / D */
  T* Aib = NAME(gather)(A, Ix, ni, B, nb, pos);
  F[i].R = NAME(ldiv_new)(bl, D, ni, ni, Aib, ni, nb);       /* :37  R = D \ Aib */
  T* S = NAME(gather)(A, B, nb, B, nb, pos);
  NAME(gemm)(bl, nb, nb, ni, (T)-1.0, Abi, nb, F[i].R, ni, (T)1.0, S, nb); /* :40  S = Abb - Abi*R */
  F[i].S = NAME(permuted)(S, nb, t, i);                      /* :41 */
  F[i].D = D;
  free(S); free(Abi); free(Aib);
}

/* a block of a child's (permuted) Schur complement as a new matrix */
static T* NAME(subblock)(const T* S, int ld, int r0, int m, int c0, int n) { return NAME(copymat)(S + r0 + (size_t)c0 * ld, ld, m, n); }

/* _factor_branch (factorization.jl:62-75) with _assemble_blocks (:115-123) */
static void NAME(factor_branch)(const hsc_blas* bl, const NAME(Csc)* A, const hsc_tree* t, int i, NAME(FNode)* F, int64_t* pos) {
  const int l = F[i].left, r = F[i].right;
  /* int1 = left.bnd[left_loc.int] ... (:64-67): global ids of the children's boundary DOFs, split by what they are for THIS node */
  const int ni1 = (int)(t->li_ptr[l + 1] - t->li_ptr[l]), nb1 = (int)(t->lb_ptr[l + 1] - t->lb_ptr[l]);
  const int ni2 = (int)(t->li_ptr[r + 1] - t->li_ptr[r]), nb2 = (int)(t->lb_ptr[r + 1] - t->lb_ptr[r]);
  int64_t* ids = (int64_t*)malloc(sizeof(int64_t) * (size_t)(ni1 + nb1 + ni2 + nb2 + 1));
  int64_t *int1 = ids, *bnd1 = int1 + ni1, *int2 = bnd1 + nb1, *bnd2 = int2 + ni2;
  const int64_t *lbnd = t->bnd_idx + t->bnd_ptr[l], *rbnd = t->bnd_idx + t->bnd_ptr[r];
  for (int k = 0; k < ni1; ++k) int1[k] = lbnd[t->li_idx[t->li_ptr[l] + k]];
  for (int k = 0; k < nb1; ++k) bnd1[k] = lbnd[t->lb_idx[t->lb_ptr[l] + k]];
  for (int k = 0; k < ni2; ++k) int2[k] = rbnd[t->li_idx[t->li_ptr[r] + k]];
  for (int k = 0; k < nb2; ++k) bnd2[k] = rbnd[t->lb_idx[t->lb_ptr[r] + k]];
  const T *S1 = F[l].S, *S2 = F[r].S;
  const int ld1 = F[l].nb, ld2 = F[r].nb;
  /* :118-121 */
  NAME(Blk) Aii = {NAME(subblock)(S1, ld1, 0, ni1, 0, ni1), NAME(gather)(A, int1, ni1, int2, ni2, pos), NAME(gather)(A, int2, ni2, int1, ni1, pos),
                   NAME(subblock)(S2, ld2, 0, ni2, 0, ni2), ni1, ni2, ni1, ni2};
  NAME(Blk) Aib = {NAME(subblock)(S1, ld1, 0, ni1, ni1, nb1), NAME(gather)(A, int1, ni1, bnd2, nb2, pos), NAME(gather)(A, int2, ni2, bnd1, nb1, pos),
                   NAME(subblock)(S2, ld2, 0, ni2, ni2, nb2), ni1, ni2, nb1, nb2};
  NAME(Blk) Abi = {NAME(subblock)(S1, ld1, ni1, nb1, 0, ni1), NAME(gather)(A, bnd1, nb1, int2, ni2, pos), NAME(gather)(A, bnd2, nb2, int1, ni1, pos),
                   NAME(subblock)(S2, ld2, ni2, nb2, 0, ni2), nb1, nb2, ni1, ni2};
  NAME(Blk) Abb = {NAME(subblock)(S1, ld1, ni1, nb1, ni1, nb1), NAME(gather)(A, bnd1, nb1, bnd2, nb2, pos), NAME(gather)(A, bnd2, nb2, bnd1, nb1, pos),
                   NAME(subblock)(S2, ld2, ni2, nb2, ni2, nb2), nb1, nb2, nb1, nb2};
  free(ids);
  NAME(blockfactor)(bl, &Aii);                                /* :69  D = blockfactor(Aii) */
  NAME(Blk) Lb = NAME(blockrdiv)(bl, &Abi, &Aii);             /* :70  L = blockrdiv(Abi, D) */
  NAME(Blk) Rb = NAME(blockldiv)(bl, &Aii, &Aib);             /* :71  R = blockldiv(D, Aib) */
  /* :72  S = Abb - Abi*R (BlockMatrix product, blockmatrix.jl:94-98, then the subtraction) */
  NAME(gemm)(bl, nb1, nb1, ni1, (T)-1.0, Abi.A11, nb1, Rb.A11, ni1, (T)1.0, Abb.A11, nb1);
  NAME(gemm)(bl, nb1, nb1, ni2, (T)-1.0, Abi.A12, nb1, Rb.A21, ni2, (T)1.0, Abb.A11, nb1);
  NAME(gemm)(bl, nb1, nb2, ni1, (T)-1.0, Abi.A11, nb1, Rb.A12, ni1, (T)1.0, Abb.A12, nb1);
  NAME(gemm)(bl, nb1, nb2, ni2, (T)-1.0, Abi.A12, nb1, Rb.A22, ni2, (T)1.0, Abb.A12, nb1);
  NAME(gemm)(bl, nb2, nb1, ni1, (T)-1.0, Abi.A21, nb2, Rb.A11, ni1, (T)1.0, Abb.A21, nb2);
  NAME(gemm)(bl, nb2, nb1, ni2, (T)-1.0, Abi.A22, nb2, Rb.A21, ni2, (T)1.0, Abb.A21, nb2);
  NAME(gemm)(bl, nb2, nb2, ni1, (T)-1.0, Abi.A21, nb2, Rb.A12, ni1, (T)1.0, Abb.A22, nb2);
  NAME(gemm)(bl, nb2, nb2, ni2, (T)-1.0, Abi.A22, nb2, Rb.A22, ni2, (T)1.0, Abb.A22, nb2);
  T* S = NAME(blk_dense)(&Abb);
  F[i].S = NAME(permuted)(S, nb1 + nb2, t, i);                /* :73-74 */
  free(S);
  F[i].L = NAME(blk_dense)(&Lb);
  F[i].R = NAME(blk_dense)(&Rb);
  F[i].DB = Aii;
  NAME(blk_free)(&Lb); NAME(blk_free)(&Rb); NAME(blk_free)(&Aib); NAME(blk_free)(&Abi); NAME(blk_free)(&Abb);
}

/* ---- ldiv! (src/factornode.jl:62-99) -------------------------------------------------------------------------------------------- */
static void NAME(apply)(const hsc_blas* bl, const T* M, int m, int k, const int64_t* rows_out, const int64_t* rows_in, T* X, int64_t n, int nrhs) {
  /* X[rows_out, :] -= M * X[rows_in, :] (factornode.jl:81,84) */
  if (m == 0 || k == 0) return;
  T* xin = NAME(newmat)(k, nrhs);
  T* y = NAME(newmat)(m, nrhs);
  for (int j = 0; j < nrhs; ++j)
    for (int q = 0; q < k; ++q) xin[q + (size_t)j * k] = X[rows_in[q] + (size_t)j * n];
  NAME(gemm)(bl, m, nrhs, k, (T)1.0, M, m, xin, k, (T)0.0, y, m);
  for (int j = 0; j < nrhs; ++j)
    for (int q = 0; q < m; ++q) X[rows_out[q] + (size_t)j * n] -= y[q + (size_t)j * m];
  free(xin); free(y);
}
static void NAME(lsolve)(const hsc_blas* bl, const hsc_tree* t, const NAME(FNode)* F, int i, T* X, int64_t n, int nrhs) { /* :77-82 */
  if (F[i].left >= 0) NAME(lsolve)(bl, t, F, F[i].left, X, n, nrhs);
  if (F[i].right >= 0) NAME(lsolve)(bl, t, F, F[i].right, X, n, nrhs);
  NAME(apply)(bl, F[i].L, F[i].nb, F[i].ni, t->bnd_idx + t->bnd_ptr[i], t->int_idx + t->int_ptr[i], X, n, nrhs);
}
static void NAME(rsolve)(const hsc_blas* bl, const hsc_tree* t, const NAME(FNode)* F, int i, T* X, int64_t n, int nrhs) { /* :83-88 */
  NAME(apply)(bl, F[i].R, F[i].ni, F[i].nb, t->int_idx + t->int_ptr[i], t->bnd_idx + t->bnd_ptr[i], X, n, nrhs);
  if (F[i].left >= 0) NAME(rsolve)(bl, t, F, F[i].left, X, n, nrhs);
  if (F[i].right >= 0) NAME(rsolve)(bl, t, F, F[i].right, X, n, nrhs);
}
static void NAME(dsolve)(const hsc_blas* bl, const hsc_tree* t, const NAME(FNode)* F, int i, T* X, int64_t n, int nrhs) { /* :89-99 */
  if (F[i].left >= 0) NAME(dsolve)(bl, t, F, F[i].left, X, n, nrhs);
  if (F[i].right >= 0) NAME(dsolve)(bl, t, F, F[i].right, X, n, nrhs);
  const int ni = F[i].ni;
  if (ni == 0) return;
  const int64_t* Ix = t->int_idx + t->int_ptr[i];
  T* x = NAME(newmat)(ni, nrhs);
  for (int j = 0; j < nrhs; ++j)
    for (int q = 0; q < ni; ++q) x[q + (size_t)j * ni] = X[Ix[q] + (size_t)j * n];
  if (F[i].left >= 0) {
    NAME(blockldiv_inplace)(bl, &F[i].DB, x, ni, nrhs);
  } else {
    T* y = NAME(ldiv_new)(bl, F[i].D, ni, ni, x, ni, nrhs);
    free(x);
    x = y;
  }
  for (int j = 0; j < nrhs; ++j)
    for (int q = 0; q < ni; ++q) X[Ix[q] + (size_t)j * n] = x[q + (size_t)j * ni];
  free(x);
}

/* factor(A, nd, nd_loc; swlevel = 0) + ldiv!(F, B) (factorization.jl:5-27, factornode.jl:62-74).
 *   x (n x nrhs, ld n): B on entry, the solution on exit;  snorm[i] = |S_i|_F of every node (parity with the NumPy restatement)
 *   stats[0..7]: factor seconds, ldiv seconds, executed real flops (factor), executed real flops (ldiv), getrf calls (factor), getrf calls (ldiv),
 *                bytes allocated over the run, singular pivots met */
int NAME(hsc_factor_solve)(int64_t n, const int64_t* colptr, const int64_t* rowidx, const T* vals, const hsc_tree* t, const hsc_blas* bl, T* x, int nrhs,
                           double* snorm, double* stats) {
  NAME(Csc) A = {n, colptr, rowidx, vals};
  const int nn = t->nnodes;
  NAME(FNode)* F = (NAME(FNode)*)calloc((size_t)nn, sizeof(NAME(FNode)));
  int64_t* pos = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t k = 0; k < n; ++k) pos[k] = -1;
  g_flops = 0.0; g_getrf = 0; g_bytes = 0.0; g_singular = 0;
  const double t0 = hsc_now();
  for (int i = 0; i < nn; ++i) { /* post-order: children before parents, the order the recursion of _factor (:14-27) visits them in */
    F[i].left = t->left[i];
    F[i].right = t->right[i];
    F[i].ni = (int)(t->int_ptr[i + 1] - t->int_ptr[i]);
    F[i].nb = (int)(t->bnd_ptr[i + 1] - t->bnd_ptr[i]);
    if ((F[i].left < 0) != (F[i].right < 0)) { /* factorization.jl:25 */
      fprintf(stderr, "hs_oracle_c: expected nested dissection to be a binary tree, found a node with only one child\n");
      free(F); free(pos);
      return -1;
    }
    if (F[i].left < 0) NAME(factor_leaf)(bl, &A, t, i, F, pos);
    else NAME(factor_branch)(bl, &A, t, i, F, pos);
    if (snorm) {
      double acc = 0.0;
      for (size_t e = 0; e < (size_t)F[i].nb * F[i].nb; ++e) acc += ABSF(F[i].S[e]) * ABSF(F[i].S[e]);
      snorm[i] = sqrt(acc);
    }
  }
  const double t1 = hsc_now();
  const double ff = g_flops;
  const long gf = g_getrf;
  const int root = nn - 1;
  NAME(lsolve)(bl, t, F, root, x, n, nrhs);  /* :69 */
  NAME(dsolve)(bl, t, F, root, x, n, nrhs);  /* :70 */
  if (F[root].nb > 0) {                      /* :72 */
    const int nb = F[root].nb;
    const int64_t* B = t->bnd_idx + t->bnd_ptr[root];
    T* xb = NAME(newmat)(nb, nrhs);
    for (int j = 0; j < nrhs; ++j)
      for (int q = 0; q < nb; ++q) xb[q + (size_t)j * nb] = x[B[q] + (size_t)j * n];
    T* y = NAME(ldiv_new)(bl, F[root].S, nb, nb, xb, nb, nrhs);
    for (int j = 0; j < nrhs; ++j)
      for (int q = 0; q < nb; ++q) x[B[q] + (size_t)j * n] = y[q + (size_t)j * nb];
    free(xb); free(y);
  }
  NAME(rsolve)(bl, t, F, root, x, n, nrhs);  /* :73 */
  const double t2 = hsc_now();
  if (stats) {
    stats[0] = t1 - t0; stats[1] = t2 - t1; stats[2] = ff; stats[3] = g_flops - ff;
    stats[4] = (double)gf; stats[5] = (double)(g_getrf - gf); stats[6] = g_bytes; stats[7] = (double)g_singular;
  }
  for (int i = 0; i < nn; ++i) {
    free(F[i].D); free(F[i].L); free(F[i].R); free(F[i].S);
    if (F[i].left >= 0) NAME(blk_free)(&F[i].DB);
  }
  free(F); free(pos);
  return 0;
}
