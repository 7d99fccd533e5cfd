"""Merged busy intervals per queue inside a time window of the last traced front factorization."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:22], r["Queue_Id"]))
rows.sort()
last = [i for i, r in enumerate(rows) if r[2].startswith("init_fronts")][-1]
rows = rows[last:]
t0 = rows[0][0]
lo, hi = float(sys.argv[2]) * 1e6 + t0, float(sys.argv[3]) * 1e6 + t0
gap = float(sys.argv[4]) * 1e3 if len(sys.argv) > 4 else 30e3
for q in sorted(set(r[3] for r in rows)):
    cur = None
    print("queue", q)
    for s, e, n, qq in rows:
        if qq != q or e < lo or s > hi:
            continue
        if cur and s - cur[1] < gap:
            cur[1] = max(cur[1], e)
            cur[2] += 1
            cur[3] += e - s
        else:
            if cur:
                print(f"   {(cur[0]-t0)/1e6:9.3f} .. {(cur[1]-t0)/1e6:9.3f} ms  ({(cur[1]-cur[0])/1e6:7.3f} ms, {cur[2]:4d} kernels, busy {cur[3]/1e6:7.3f})  first={cur[4]}")
            cur = [s, e, 1, e - s, n]
    if cur:
        print(f"   {(cur[0]-t0)/1e6:9.3f} .. {(cur[1]-t0)/1e6:9.3f} ms  ({(cur[1]-cur[0])/1e6:7.3f} ms, {cur[2]:4d} kernels, busy {cur[3]/1e6:7.3f})  first={cur[4]}")
