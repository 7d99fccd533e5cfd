"""Module-level reproduction of the sub-1e-10 erratic error: compress ONE matrix with hs.hss.compress at
tolerances 1e-6 .. 1e-15 and print rank, |full(H) - K| / |K| and the solve error.  Usage (GPU box): python tools/qr_floor_probe.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hsamd

hs = hsamd.load()

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(0)
pts = np.sort(rng.random(n))
Kr = 1.0 / (1.0 + 40.0 * np.abs(pts[:, None] - pts[None, :])) + 0.05 * n * np.eye(n)
# complex: 2-D points on a sheet (a separator), Helmholtz-like kernel
m = int(round(np.sqrt(n)))
gx, gy = np.meshgrid(np.arange(m), np.arange(m), indexing="ij")
P2 = np.stack([gx.ravel(), gy.ravel()], 1).astype(float)
# recursive-bisection order keeps clusters compact
def rb(idx, out):
    if len(idx) <= 16:
        out.extend(idx)
        return
    c = P2[idx]
    ax = int(np.argmax(c.max(0) - c.min(0)))
    o = np.argsort(c[:, ax], kind="stable")
    h = len(idx) // 2
    rb([idx[i] for i in o[:h]], out)
    rb([idx[i] for i in o[h:]], out)
order = []
rb(list(range(m * m)), order)
P2 = P2[order]
R = np.sqrt(((P2[:, None, :] - P2[None, :, :]) ** 2).sum(-1))
Kz = np.exp(1j * 0.7 * R) / (1.0 + R) + (4.0 + 1.0j) * np.eye(m * m)
tols = [float(x) for x in os.environ.get("PROBE_TOLS", "1e-6,1e-8,1e-9,1e-10,1e-11,1e-12,1e-13,1e-14,1e-15").split(",")]
kinds = os.environ.get("PROBE_KINDS", "real-1d,cplx-2d").split(",")
for name, K in (("real-1d", Kr), ("cplx-2d", Kz)):
    if name not in kinds:
        continue
    b = rng.standard_normal(K.shape[0]) + (1j * rng.standard_normal(K.shape[0]) if np.iscomplexobj(K) else 0)
    xs = np.linalg.solve(K, b)
    for tol in tols:
        H = hs.hss.compress(K, leafsize=64, atol=tol * 1e-3, rtol=tol, kest=64)
        eh = np.linalg.norm(H.full() - K) / np.linalg.norm(K)
        es = np.linalg.norm(H.ldiv(b) - xs) / np.linalg.norm(xs)
        print(f"{name} n={K.shape[0]} tol={tol:g}: hssrank {H.rank} samples {H.samples}  |full-K|/|K| {eh:.2e}  solve err {es:.2e}", flush=True)
