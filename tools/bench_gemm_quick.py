"""Three shapes of the FP64 MFMA GEMM (kernel experiments): python tools/bench_gemm_quick.py"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsamd
hs = hsamd.load(); L = hs._lib.lib()
def ptr(a): return a.ctypes.data_as(C.POINTER(C.c_double))
def run(M, N, K, cplx=False, rep=5):
    rng = np.random.default_rng(0)
    mk = (lambda s: np.asfortranarray(rng.standard_normal(s) + (1j * rng.standard_normal(s) if cplx else 0)))
    A, B, Cm = mk((M, K)), mk((K, N)), mk((M, N))
    ref = Cm[:64, :64] - A[:64] @ B[:, :64]
    ms = C.c_double(0)
    fn = L.hsk_gemm_z if cplx else L.hsk_gemm_d
    hs._lib.check(fn(M, N, K, ptr(A), M, ptr(B), K, ptr(Cm), M, 1, 0, C.byref(ms)))
    err = np.abs(Cm[:64, :64] - ref).max() / np.abs(ref).max()
    hs._lib.check(fn(M, N, K, ptr(A), M, ptr(B), K, ptr(Cm), M, 1, rep, C.byref(ms)))
    fl = 2.0 * M * N * K * (4 if cplx else 1)
    print(f"{'z' if cplx else 'd'} M={M:6d} N={N:6d} K={K:6d}  {ms.value:9.3f} ms  {fl / ms.value / 1e9:8.2f} TFLOP/s  err {err:.1e}", flush=True)
for s in [(8192, 8192, 8192), (16384, 16384, 1024), (16384, 16384, 256), (30000, 1024, 1024)]:
    run(*s)
run(4096, 4096, 4096, cplx=True)
