"""Debug aid: per-node LU blocks of an exact factorization, dumped for comparison between HS_GROUP_FUSED=0 and 1."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import hsamd
hs = hsamd.load()
from helpers import prepare
name = sys.argv[1]
P = prepare(hs, name, rhs="randn")
F = hs.factor(P["A"], P["nd"], P["nd_loc"], swlevel=0)
x = hs.ldiv(F, P["b"])
print("residual", np.linalg.norm(P["A"] @ x - P["b"]) / np.linalg.norm(P["b"]))
out = {}
for k in range(F.nnodes):
    ni, nb, lv = F.node_info(k)
    if ni >= 256:
        b = F.node_blocks(k)
        out[f"LU{k}"] = b["LU"]; out[f"rp{k}"] = b["rperm"]; out[f"Lbi{k}"] = b["Lbi"]
np.savez(sys.argv[2], **out)
