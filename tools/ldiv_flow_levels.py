"""Per-level durations of the sweep kernels of the LAST solve in a tools/ldiv_profile.sh trace (gpurun_out/ldivprof)."""
import csv, glob
import os
kt = sorted(glob.glob("gpurun_out/ldivprof/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(kt))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for key in ("flow_sweep_kernel<double, false>", "flow_sweep_kernel<double, true>", "flow_sweep_kernel<cplx, false>", "flow_sweep_kernel<cplx, true>", "fwd_wide", "bwd_wide", "int_update_partial"):
    fl = [r for r in rows if key in r["Kernel_Name"]]
    if not fl:
        continue
    n = len(fl) // 4
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in fl[-n:]]
    g = [int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) for r in fl[-n:]]
    if n > 24:  # launch-per-step sweeps: one line per level (= per batch size, grid.y)
        import collections
        lv = collections.OrderedDict()
        for r, a in zip(fl[-n:], d):
            key2 = int(r["Grid_Size_Y"])
            lv.setdefault(key2, [0, 0.0])
            lv[key2][0] += 1
            lv[key2][1] += a
        print("%s: %d launches per solve, %.2f ms; per level (fronts: steps, us): %s" % (key, n, sum(d) * 1e-3, " ".join("%d: %d, %.0f" % (k2, v[0], v[1]) for k2, v in lv.items())))
        continue
    print("%s: %d launches per solve, %.2f ms; (us/workgroups): %s" % (key, n, sum(d) * 1e-3, " ".join("%.0f/%d" % (a, b) for a, b in zip(d, g)) if n <= 24 else "..."))
