"""GEMM kernel micro-benchmark (not a test): TFLOP/s of hsk_gemm_{d,z} over shapes."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsamd
hs = hsamd.load(); L = hs._lib.lib()
def ptr(a): return a.ctypes.data_as(C.POINTER(C.c_double))
def run(M,N,K,cplx=False,rep=5):
    rng = np.random.default_rng(0)
    mk = (lambda s: np.asfortranarray(rng.standard_normal(s) + (1j*rng.standard_normal(s) if cplx else 0)))
    A, B, Cm = mk((M,K)), mk((K,N)), mk((M,N))
    ms = C.c_double(0)
    fn = L.hsk_gemm_z if cplx else L.hsk_gemm_d
    hs._lib.check(fn(M,N,K,ptr(A),M,ptr(B),K,ptr(Cm),M,1,rep,C.byref(ms)))
    fl = 2.0*M*N*K*(4 if cplx else 1)
    print(f"{'z' if cplx else 'd'} M={M:6d} N={N:6d} K={K:6d}  {ms.value:9.3f} ms  {fl/ms.value/1e9:8.2f} TFLOP/s", flush=True)
shapes = [(8192,8192,8192),(4096,4096,4096),(16384,16384,32),(16384,16384,64),(16384,16384,128),(16384,16384,256),(16384,16384,512),(16384,16384,1024),(16384,16384,4096),
          (4096,4096,32),(4096,4096,128),(4096,4096,512),(2048,2048,2048),(1024,1024,1024),(16384,32,16384),(32,16384,32),(128,16384,128)]
for s in shapes: run(*s)
for s in [(4096,4096,4096),(8192,8192,512),(8192,8192,64)]: run(*s, cplx=True)
