"""Summarise a rocprofv3 kernel-trace CSV: busy time per kernel, union busy time, time with no GEMM running."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
rows.sort()
if len(sys.argv) > 2:
    frac = float(sys.argv[2])
    lo, hi = rows[0][0], max(e for _, e, _ in rows)
    cut = lo + (hi - lo) * (1 - frac)
    rows = [r for r in rows if r[0] >= cut]
per = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    per[n][0] += e - s
    per[n][1] += 1
span = max(e for _, e, _ in rows) - rows[0][0]
# union
ev = []
for s, e, n in rows:
    g = "gemm" in n
    ev.append((s, 1, g))
    ev.append((e, -1, g))
ev.sort()
busy = gemm_busy = 0
na = ng = 0
last = ev[0][0]
for t, d, g in ev:
    if na > 0:
        busy += t - last
    if ng > 0:
        gemm_busy += t - last
    last = t
    na += d
    if g:
        ng += d
print(f"span {span/1e6:.2f} ms  any-kernel busy {busy/1e6:.2f} ms  gemm running {gemm_busy/1e6:.2f} ms  no-gemm {(span-gemm_busy)/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms")
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{t/1e6:10.2f} ms {c:7d} x  avg {t/c/1e3:9.1f} us  {n}")
