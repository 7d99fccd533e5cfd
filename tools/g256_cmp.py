import sys, numpy as np
a = np.load(sys.argv[1]); b = np.load(sys.argv[2])
for k in a.files:
    A, B = a[k], b[k]
    if A.shape != B.shape: print(k, "shape", A.shape, B.shape); continue
    d = np.abs(A.astype(complex) - B.astype(complex))
    if d.max() > 1e-9 * max(1.0, np.abs(A).max()):
        print(k, A.shape, "max diff", d.max())
        if A.ndim == 2:
            nbr, nbc = (A.shape[0] + 31) // 32, (A.shape[1] + 31) // 32
            bad = [(i, j) for i in range(nbr) for j in range(nbc) if d[32*i:32*i+32, 32*j:32*j+32].max() > 1e-9]
            print("  bad 32-blocks (row, col):", bad[:40], "..." if len(bad) > 40 else "")
        break
else:
    print("all equal")
