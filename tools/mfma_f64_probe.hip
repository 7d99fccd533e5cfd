// mfma_f64_probe.hip -- issue interval and sustained clock of v_mfma_f64_16x16x4_f64 on gfx950.
// The guides give no FP64 MFMA rate (MI355X_MICROARCH.md "Matrix cores" has no f64 row); this probe
// fixes the roofline denominator: cycles per MFMA per SIMD (s_memtime) and the clock the chip holds
// (s_memtime / s_memrealtime * 100 MHz), for 1/2/4 waves per SIMD and 4/8/16 independent accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC, int CHAIN>
__global__ __launch_bounds__(256) void probe(double* out, unsigned long long* stamps, int iters, double seed) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  double a = seed * (1.0 + (threadIdx.x % 17) * 0.013), b = seed * (0.7 - (threadIdx.x % 13) * 0.021);
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
#pragma unroll
      for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);  // CHAIN back-to-back dependent
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) {
    int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = r1 - r0;
  }
}

template <int NACC, int CHAIN>
void run(int wps, int iters, double seed) {
  iters /= CHAIN;
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int blocks = prop.multiProcessorCount * wps;
  double* out; unsigned long long* st;
  hipMalloc(&out, sizeof(double) * blocks * 256); hipMalloc(&st, sizeof(unsigned long long) * blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL((probe<NACC, CHAIN>), dim3(blocks), dim3(256), 0, 0, out, st, iters, seed);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NACC, CHAIN>), dim3(blocks), dim3(256), 0, 0, out, st, iters, seed);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), st, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (int w = 0; w < blocks * 4; ++w) { cyc.push_back((double)h[2 * w] / ((double)iters * NACC * CHAIN)); clk.push_back((double)h[2 * w] / (double)h[2 * w + 1] * 100.0); }
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  double flops = (double)blocks * 4 * iters * NACC * CHAIN * 2048.0;
  printf("nacc=%2d chain=%d waves/simd=%d seed=%g: %7.2f TFLOP/s  cycles/MFMA/wave median %.1f  => per-SIMD issue interval %.1f cyc  clock median %.0f MHz\n", NACC, CHAIN, wps, seed,
         flops / (ms * 1e-3) / 1e12, cyc[cyc.size() / 2], cyc[cyc.size() / 2] / wps, clk[clk.size() / 2]);
  hipFree(out); hipFree(st);
}
int main() {
  int it = 20000;
  for (double seed : {1.0}) {
    run<8, 1>(1, it, seed); run<8, 2>(1, it, seed); run<8, 4>(1, it, seed); run<8, 8>(1, it, seed); run<1, 8>(1, it * 8, seed); run<2, 16>(1, it * 4, seed);
    run<8, 1>(2, it, seed); run<8, 2>(2, it, seed); run<8, 4>(2, it, seed); run<8, 8>(2, it, seed); run<1, 8>(2, it * 8, seed); run<16, 4>(2, it / 2, seed);
  }
  return 0;
}
