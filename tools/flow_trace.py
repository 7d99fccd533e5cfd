"""Timing experiment (library built with -DHS_FLOW_TRACE): device timestamps of the chain phases of the root's forward sweep."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import hsamd
hs = hsamd.load()
A, b, nd = hs.problems.make_problem(sys.argv[1] if len(sys.argv) > 1 else "poisson3d_128", rhs="randn")
nd, nd_loc = hs.symfact(nd); perm = hs.postorder(nd)
A = A[perm - 1][:, perm - 1].tocsc(); nd = hs.permuted(nd, hs.invperm(perm)); b = b[perm - 1]
F = hs.factor(A, nd, nd_loc, swlevel=0)
for _ in range(3):
    x = hs.ldiv(F, b)
torch.cuda.synchronize()
L = hs._lib.lib()
n = 8 * 512
buf = (C.c_ulonglong * n)()
L.hsk_flow_trace.argtypes = [C.c_void_p, C.c_int]
assert L.hsk_flow_trace(buf, n) == 0
T = np.array(buf[:], dtype=np.int64).reshape(-1, 8)[:, :4] * 10e-3  # us (100 MHz)
T = T[T[:, 3] > 0]
print("sub-blocks traced:", len(T))
# per 256-block: q = 3 is the last of its block; step = time between y of consecutive blocks
y = T[3::4, 3]
print("y_j published -> y_{j+1} published (us), median %.2f mean %.2f" % (np.median(np.diff(y)), np.mean(np.diff(y))))
for name, a, bq in (("last round done -> w published", 0, 1), ("w published -> w of the block in", 1, 2), ("w in -> y published", 2, 3)):
    d = T[:, bq] - T[:, a]
    print("%-34s q=0 %.2f  q=1 %.2f  q=2 %.2f  q=3 %.2f (median us)" % (name, *[np.median(d[q::4]) for q in range(4)]))
# from y_{j} published (by q=3 of block j) to 'last round done' of the sub-blocks of block j+1
for q in range(4):
    d = T[4 + q::4, 0][: len(y) - 1] - y[: len(T[4 + q::4, 0])][: len(T[4 + q::4, 0][: len(y) - 1])]
    print("y_j published -> last round of block j+1, q=%d done: median %.2f us" % (q, np.median(d)))
