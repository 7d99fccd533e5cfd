"""Export the per-kernel summary of a rocprofv3 run that wrote a rocpd SQLite database (ROCm 7.2 default output).

    python tools/rocpd_top_kernels.py gpurun_out/prof/xxx_results.db out.csv
"""
import csv
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
    for r in rows:
        w.writerow([r[0], r[1], round(r[2], 1), round(r[3], 2), round(r[4], 3)])
print(f"{len(rows)} kernels -> {sys.argv[2]}")
