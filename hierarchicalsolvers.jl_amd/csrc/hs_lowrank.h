// hs_lowrank.h -- device low-rank factor X ~= P' * trap(Lp[:, :r]) * Z (see hs_lowrank.hip)
#pragma once
#include <cstdint>
#include "hs_common.h"

template <class T>
struct LowRank {
  T* Lp = nullptr;      // rows x k packed L\U of the pivoted sketch (ld = ldp); C = P' * unit-lower-trapezoid(Lp[:, :r])
  int* rperm = nullptr; // rows: pivoted row i of Lp is original row rperm[i]
  T* Z = nullptr;       // r x cols (ld = ldz)
  int rows = 0, cols = 0, k = 0, r = 0, ldp = 0, ldz = 0;
};

template <class T>
int lowrank_compress(T* X, int ldx, int rows, int cols, double atol, double rtol, int kinit, uint64_t seed, hipStream_t s, LowRank<T>* out);
template <class T>
void lowrank_free(LowRank<T>& lr);
