"""One launch of the MFMA GEMM kernel through the test hook (used by tools/pmc_gemm.sh, tools/pmc_traffic.sh): M N K [repeat]."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hsamd
hs = hsamd.load(); L = hs._lib.lib()
M,N,K = [int(a) for a in sys.argv[1:4]]; rep = int(sys.argv[4]) if len(sys.argv)>4 else 3
rng = np.random.default_rng(0)
A = np.asfortranarray(rng.standard_normal((M,K))); B = np.asfortranarray(rng.standard_normal((K,N))); Cm = np.asfortranarray(rng.standard_normal((M,N)))
p = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
ms = C.c_double(0)
hs._lib.check(L.hsk_gemm_d(M,N,K,p(A),M,p(B),K,p(Cm),M,1,rep,C.byref(ms)))
print(M,N,K, ms.value, "ms", 2.0*M*N*K/ms.value/1e9, "TF/s")
