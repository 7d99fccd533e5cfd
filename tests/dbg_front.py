import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hsamd
hs = hsamd.load()
import test_kernels_gpu as t
ni, nb, cplx = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3]))
print("front_check", ni, nb, cplx, flush=True)
print("worst", t.front_check(hs, 1, ni, nb, cplx, seed=1))
