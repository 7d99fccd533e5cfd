"""Look-ahead timeline of a traced front factorization: per outer step, the big GEMM vs the side-stream panel work."""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"]))
rows.sort()
last = [i for i, r in enumerate(rows) if r[2].startswith("init_fronts")][-1]
rows = rows[last:]
t0 = rows[0][0]
span = max(r[1] for r in rows) - t0
print(f"span {span/1e6:.1f} ms, {len(rows)} launches")
perq = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for s, e, n, q in rows:
    perq[q][n][0] += (e - s) / 1e6
    perq[q][n][1] += 1
for q, d in perq.items():
    tot = sum(v[0] for v in d.values())
    print(f"queue {q}: busy {tot:.1f} ms")
    for n, (t, c) in sorted(d.items(), key=lambda kv: -kv[1][0])[:6]:
        print(f"    {t:9.2f} ms {c:6d} x avg {t/c*1e3:8.1f} us  {n[:40]}")
big = [(s, e, q) for s, e, n, q in rows if "gemm" in n and e - s > 1e6]
print("big GEMMs (start ms, dur ms, queue):")
for s, e, q in big[:80]:
    print(f"   {(s-t0)/1e6:8.2f} +{(e-s)/1e6:7.2f}  q{q}")
