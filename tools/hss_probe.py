"""Times hs_hss_compress / hs_hss_factor / hs_hss_ldiv on a synthetic kernel block (device HSS module, include/hs_hss.h)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import hsamd

hs = hsamd.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
leaf = int(sys.argv[2]) if len(sys.argv) > 2 else 256
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
kest = int(sys.argv[4]) if len(sys.argv) > 4 else 64
rng = np.random.default_rng(0)
x = np.sort(rng.random(n))
d = np.abs(x[:, None] - x[None, :])
A = 1.0 / (1.0 + 40.0 * d) + n * 0.05 * np.eye(n)
for rep in range(2):
    t0 = time.perf_counter()
    H = hs.hss.compress(A, leafsize=leaf, atol=tol, rtol=tol, kest=kest)
    t1 = time.perf_counter()
    b = np.ones((n, 4))
    xs = H.ldiv(b)
    t2 = time.perf_counter()
    xs = H.ldiv(b)
    t3 = time.perf_counter()
    print(f"n={n} leaf={leaf} tol={tol:g}: rank {H.rank} samples {H.samples} device compress {H.times()['compress_s']*1e3:.1f} ms "
          f"factor {H.times()['factor_s']*1e3:.1f} ms | host: compress+upload {1e3*(t1-t0):.0f} ms, first ldiv (4 rhs, incl. factor) {1e3*(t2-t1):.0f} ms, "
          f"second ldiv {1e3*(t3-t2):.1f} ms, residual {np.linalg.norm(A @ xs - b) / np.linalg.norm(b):.1e}")
