"""Diagnostic: factor ONE random front through the test hook (run under rocprofv3 --kernel-trace)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import hsamd

hs = hsamd.load()
from hierarchicalsolvers_jl_amd import _lib

ni, nb = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
m = ni + nb
rng = np.random.default_rng(0)
F = np.asfortranarray(rng.standard_normal((m, m)))
if os.environ.get("SPD", "1") == "1":  # diagonally dominant like the fronts of the Poisson workloads (no swaps)
    F *= 1.0 / np.sqrt(m)
    F[np.arange(m), np.arange(m)] += 4.0
else:
    F[np.arange(m), np.arange(m)] += 4.0
L = _lib.lib()
outLF = np.zeros((m, ni), order="F")
outUR = np.zeros((ni, max(nb, 1)), order="F")
outSB = np.zeros((max(nb, 1), max(nb, 1)), order="F")
rp = np.zeros(ni, dtype=np.int64)
info = np.zeros(1, dtype=np.int64)
ms = C.c_double(0)
p = lambda a: a.ctypes.data_as(_lib.p_f64)
for _ in range(reps):
    _lib.check(L.hsk_front_factor_d(1, ni, nb, p(F), p(outLF), p(outUR), p(outSB), rp.ctypes.data_as(_lib.p_i64), info.ctypes.data_as(_lib.p_i64), C.byref(ms)))
    fl = 2 / 3 * ni**3 + 2 * ni * ni * nb + 2 * ni * nb * nb
    print(f"front ni={ni} nb={nb}: {ms.value:.2f} ms  {fl / ms.value / 1e9:.2f} TFLOP/s", flush=True)
