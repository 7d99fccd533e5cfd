// hs_testhooks.hip -- kernel-level entry points used only by tests/ (declared in include/hs_kernels.h).
// They drive exactly the kernels hs_factor_* uses, on caller-supplied dense data, so every kernel
// can be checked against the oracle in isolation.
#include <cstring>
#include <chrono>
#include <vector>

#include "../../include/hs_kernels.h"
#include "hs_sched.h"
#include "hs_lowrank.h"

#define CK(call)                                                                                   \
  do {                                                                                             \
    hipError_t e__ = (call);                                                                       \
    if (e__ != hipSuccess) {                                                                       \
      hs_set_error(HS_ERR_DEVICE, 0, "%s failed: %s", #call, hipGetErrorString(e__));              \
      return HS_ERR_DEVICE;                                                                        \
    }                                                                                              \
  } while (0)

template <class T>
static int gemm_hook(int64_t M, int64_t N, int64_t K, const T* A, int64_t lda, const T* B, int64_t ldb, T* C, int64_t ldc, int minus,
                     int repeat, double* ms_out) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    hs_set_error(HS_ERR_DEVICE, 0, "no HIP device available");
    return HS_ERR_DEVICE;
  }
  T *dA = nullptr, *dB = nullptr, *dC = nullptr;
  GemmProb<T>* dp = nullptr;
  CK(hipMalloc((void**)&dA, sizeof(T) * (size_t)lda * K + 16));
  CK(hipMalloc((void**)&dB, sizeof(T) * (size_t)ldb * N + 16));
  CK(hipMalloc((void**)&dC, sizeof(T) * (size_t)ldc * N + 16));
  CK(hipMalloc((void**)&dp, sizeof(GemmProb<T>)));
  CK(hipMemcpy(dA, A, sizeof(T) * (size_t)lda * K, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B, sizeof(T) * (size_t)ldb * N, hipMemcpyHostToDevice));
  CK(hipMemcpy(dC, C, sizeof(T) * (size_t)ldc * N, hipMemcpyHostToDevice));
  GemmProb<T> p{dA, dB, dC, (int)M, (int)N, (int)K, (int)lda, (int)ldb, (int)ldc};
  CK(hipMemcpy(dp, &p, sizeof p, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  launch_gemm_probs<T>(dp, 1, (int)M, (int)N, minus, 0);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(C, dC, sizeof(T) * (size_t)ldc * N, hipMemcpyDeviceToHost));
  if (repeat > 0) {
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < repeat; ++r) launch_gemm_probs<T>(dp, 1, (int)M, (int)N, minus, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms_out) *ms_out = ms / repeat;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  (void)hipFree(dp);
  return HS_OK;
}

extern "C" int hsk_gemm_d(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                          int64_t ldc, int minus, int repeat, double* ms_out) {
  return gemm_hook<double>(M, N, K, A, lda, B, ldb, C, ldc, minus, repeat, ms_out);
}
extern "C" int hsk_gemm_z(int64_t M, int64_t N, int64_t K, const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                          int64_t ldc, int minus, int repeat, double* ms_out) {
  return gemm_hook<cplx>(M, N, K, (const cplx*)A, lda, (const cplx*)B, ldb, (cplx*)C, ldc, minus, repeat, ms_out);
}

// Factor `count` identical-shape dense fronts F[k] ((ni+nb)^2, column-major, front order [int;bnd]) in one batch.
template <class T>
static int front_hook(int64_t count, int64_t ni, int64_t nb, const T* F, T* outLF, T* outUR, T* outSB, int64_t* out_rperm, int64_t* info,
                      double* ms_out) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    hs_set_error(HS_ERR_DEVICE, 0, "no HIP device available");
    return HS_ERR_DEVICE;
  }
  const int m = (int)(ni + nb);
  const int ldl = (m + 1) / 2 * 2, ldu = ((int)ni + 1) / 2 * 2 > 0 ? ((int)ni + 1) / 2 * 2 : 2, lds = ((int)nb + 1) / 2 * 2 > 0 ? ((int)nb + 1) / 2 * 2 : 2;
  const int nblk = ((int)ni + HS_PB - 1) / HS_PB;
  const size_t eLF = (size_t)ldl * ni + 32, eUR = (size_t)ldu * nb + 32, eSB = (size_t)lds * nb + 32, eInv = (size_t)2 * nblk * 1024 + 32;
  const int ncand = (((int)ni + HS_CHUNK - 1) / HS_CHUNK + 1) * HS_PB;
  const size_t eInt = (size_t)2 * ni + 2 * ncand + HS_PB + 1;
  T* dbuf = nullptr;
  int* dint = nullptr;
  NodeDesc<T>* dn = nullptr;
  const size_t per = eLF + eUR + eSB + eInv;
  CK(hipMalloc((void**)&dbuf, sizeof(T) * per * count));
  CK(hipMalloc((void**)&dint, sizeof(int) * eInt * count));
  CK(hipMalloc((void**)&dn, sizeof(NodeDesc<T>) * count));
  CK(hipMemset(dbuf, 0, sizeof(T) * per * count));
  CK(hipMemset(dint, 0, sizeof(int) * eInt * count));
  std::vector<NodeDesc<T>> hn(count);
  for (int64_t k = 0; k < count; ++k) {
    NodeDesc<T>& d = hn[k];
    memset(&d, 0, sizeof d);
    d.LF = dbuf + per * k;
    d.UR = d.LF + eLF;
    d.SB = d.UR + eUR;
    d.invL = d.SB + eSB;
    d.invU = d.invL + (size_t)nblk * 1024;
    d.ipiv = dint + eInt * k;
    d.rperm = d.ipiv + ni;
    d.cand0 = d.rperm + ni;
    d.cand1 = d.cand0 + ncand;
    d.pivlist = d.cand1 + ncand;
    d.info = d.pivlist + HS_PB;
    d.growth = d.info;  // HS_HOOK_OPTIMISTIC=1: the growth flag of optimistic pivoting comes back through `info` as -1
    d.ni = (int)ni; d.nb = (int)nb; d.m = m;
    d.ldl = ldl; d.ldu = ldu; d.lds = lds;
    d.ni1 = (int)ni; d.nb1 = (int)nb; d.isleaf = 1; d.node = (int)k;
    d.finalize();
    const T* Fk = F + (size_t)m * m * k;
    if (ni > 0) CK(hipMemcpy2D(d.LF, sizeof(T) * ldl, Fk, sizeof(T) * m, sizeof(T) * m, ni, hipMemcpyHostToDevice));
    if (ni > 0 && nb > 0) CK(hipMemcpy2D(d.UR, sizeof(T) * ldu, Fk + (size_t)m * ni, sizeof(T) * m, sizeof(T) * ni, nb, hipMemcpyHostToDevice));
    if (nb > 0) CK(hipMemcpy2D(d.SB, sizeof(T) * lds, Fk + (size_t)m * ni + ni, sizeof(T) * m, sizeof(T) * nb, nb, hipMemcpyHostToDevice));
  }
  CK(hipMemcpy(dn, hn.data(), sizeof(NodeDesc<T>) * count, hipMemcpyHostToDevice));
  Profiler prof;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // same stream set-up as hs_analyze: a main stream and a high-priority side stream for look-ahead panels
  hipStream_t s_main = nullptr, s_side = nullptr, s_la = nullptr, s_sidem = nullptr;
  CK(hipStreamCreate(&s_main));
  hs_create_lookahead_streams(&s_la, &s_sidem, &s_side);
  CK(hipEventRecord(e0, s_main));
  launch_init_fronts<T>(dn, (int)count, (int)ni, s_main);
  Sched<T> sch{dn, (int)count, (int)ni, (int)nb, m, s_main, &prof, nullptr, nullptr, s_side, 0, s_la, s_sidem};
  const auto h0 = std::chrono::steady_clock::now();
  const bool hook_opt = getenv("HS_HOOK_OPTIMISTIC") != nullptr;
  sch.optimistic = hook_opt;
  sch.factor_fronts();
  CK(hipEventRecord(e1, s_main));
  if (getenv("HS_HOOK_VERBOSE"))
    fprintf(stderr, "[hook] host enqueue time %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count());
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  if (ms_out) *ms_out = ms;
  CK(hipDeviceSynchronize());
  std::vector<int> ip(ni);
  for (int64_t k = 0; k < count; ++k) {
    NodeDesc<T>& d = hn[k];
    if (ni > 0 && outLF) CK(hipMemcpy2D(outLF + (size_t)m * ni * k, sizeof(T) * m, d.LF, sizeof(T) * ldl, sizeof(T) * m, ni, hipMemcpyDeviceToHost));
    if (ni > 0 && nb > 0 && outUR) CK(hipMemcpy2D(outUR + (size_t)ni * nb * k, sizeof(T) * ni, d.UR, sizeof(T) * ldu, sizeof(T) * ni, nb, hipMemcpyDeviceToHost));
    if (nb > 0 && outSB) CK(hipMemcpy2D(outSB + (size_t)nb * nb * k, sizeof(T) * nb, d.SB, sizeof(T) * lds, sizeof(T) * nb, nb, hipMemcpyDeviceToHost));
    if (ni > 0) CK(hipMemcpy(ip.data(), d.rperm, sizeof(int) * ni, hipMemcpyDeviceToHost));
    if (out_rperm)
      for (int64_t i = 0; i < ni; ++i) out_rperm[ni * k + i] = ip[i];
    int inf = 0;
    CK(hipMemcpy(&inf, d.info, sizeof(int), hipMemcpyDeviceToHost));
    if (info) info[k] = (hook_opt && inf != 0) ? -1 : inf;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(s_main);
  if (s_side) (void)hipStreamDestroy(s_side);
  if (s_la) (void)hipStreamDestroy(s_la);
  if (s_sidem) (void)hipStreamDestroy(s_sidem);
  (void)hipFree(dbuf);
  (void)hipFree(dint);
  (void)hipFree(dn);
  return HS_OK;
}

extern "C" int hsk_front_factor_d(int64_t count, int64_t ni, int64_t nb, const double* F, double* outLF, double* outUR, double* outSB,
                                  int64_t* out_rperm, int64_t* info, double* ms_out) {
  return front_hook<double>(count, ni, nb, F, outLF, outUR, outSB, out_rperm, info, ms_out);
}
extern "C" int hsk_front_factor_z(int64_t count, int64_t ni, int64_t nb, const double* F, double* outLF, double* outUR, double* outSB,
                                  int64_t* out_rperm, int64_t* info, double* ms_out) {
  return front_hook<cplx>(count, ni, nb, (const cplx*)F, (cplx*)outLF, (cplx*)outUR, (cplx*)outSB, out_rperm, info, ms_out);
}

// Low-rank compression of a dense rows x cols block (column-major host array): returns the rank r and
// the dense factors C (rows x r) and Z (r x cols), X ~= C * Z, built from the device representation.
template <class T>
static int lowrank_hook(int64_t rows, int64_t cols, const T* X, double atol, double rtol, int64_t kinit, int64_t seed, int64_t* r_out, T* Cout,
                        T* Zout, int64_t cap) {
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    hs_set_error(HS_ERR_DEVICE, 0, "no HIP device available");
    return HS_ERR_DEVICE;
  }
  T* dX = nullptr;
  const int ldx = ((int)rows + 1) / 2 * 2;
  CK(hipMalloc((void**)&dX, sizeof(T) * ((size_t)ldx * cols + 32)));
  CK(hipMemcpy2D(dX, sizeof(T) * ldx, X, sizeof(T) * rows, sizeof(T) * rows, cols, hipMemcpyHostToDevice));
  // the primitive the compressed fronts use (hs_compress.h): interpolative form, rank and interpolation from the orthogonalisation of
  // the sketch rows in tournament-pivot order (lowrank_id_batch); HS_LR_QR=0: the LU-based factors of hs_lowrank.hip
  static const bool lr_qr = !(getenv("HS_LR_QR") && getenv("HS_LR_QR")[0] == '0');
  LowRank<T> lr;
  int st;
  if (lr_qr) {
    LowRankJob<T> job{dX, ldx, (int)rows, (int)cols, (int)kinit, (uint64_t)seed, &lr, 0};
    st = lowrank_id_batch<T>(&job, 1, atol, rtol, 0);
  } else {
    st = lowrank_compress<T>(dX, ldx, (int)rows, (int)cols, atol, rtol, (int)kinit, (uint64_t)seed, 0, &lr);
  }
  if (st != 0) return st;
  *r_out = lr.r;
  if (lr.r > cap) {
    hs_set_error(HS_ERR_ARGUMENT, lr.r, "rank %d exceeds the output capacity %lld", lr.r, (long long)cap);
    return HS_ERR_ARGUMENT;
  }
  if (lr.Cd) {
    if (lr.r > 0) CK(hipMemcpy2D(Cout, sizeof(T) * rows, lr.Cd, sizeof(T) * lr.ldc, sizeof(T) * rows, lr.r, hipMemcpyDeviceToHost));
  } else {
    std::vector<T> hL((size_t)lr.ldp * lr.k);
    std::vector<int> rp(rows);
    CK(hipMemcpy(hL.data(), lr.Lp, sizeof(T) * hL.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(rp.data(), lr.rperm, sizeof(int) * rows, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < rows; ++i)
      for (int64_t j = 0; j < lr.r; ++j) {
        T v = (i == j) ? Scal<T>::one() : (i > j ? hL[(size_t)i + (size_t)j * lr.ldp] : Scal<T>::zero());
        Cout[(size_t)rp[i] + (size_t)j * rows] = v;
      }
  }
  if (lr.r > 0) CK(hipMemcpy2D(Zout, sizeof(T) * lr.r, lr.Z, sizeof(T) * lr.ldz, sizeof(T) * lr.r, cols, hipMemcpyDeviceToHost));
  lowrank_free(lr);
  (void)hipFree(dX);
  return HS_OK;
}
extern "C" int hsk_lowrank_d(int64_t rows, int64_t cols, const double* X, double atol, double rtol, int64_t kinit, int64_t seed, int64_t* r_out,
                             double* Cout, double* Zout, int64_t cap) {
  return lowrank_hook<double>(rows, cols, X, atol, rtol, kinit, seed, r_out, Cout, Zout, cap);
}
extern "C" int hsk_lowrank_z(int64_t rows, int64_t cols, const double* X, double atol, double rtol, int64_t kinit, int64_t seed, int64_t* r_out,
                             double* Cout, double* Zout, int64_t cap) {
  return lowrank_hook<cplx>(rows, cols, (const cplx*)X, atol, rtol, kinit, seed, r_out, (cplx*)Cout, (cplx*)Zout, cap);
}
