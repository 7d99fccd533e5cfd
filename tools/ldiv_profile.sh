#!/bin/bash
# usage (GPU box): tools/ldiv_profile.sh [WORKLOAD] -- kernel trace of the solves; prints per-kernel time per solve and the idle gaps of the LAST solve
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/ldivprof; mkdir -p $R/gpurun_out/ldivprof; cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ldivprof -- python3 tools/ldiv_profile.py ${1:-poisson3d_128} 4 > gpurun_out/ldivprof/run.log 2>&1
tail -1 gpurun_out/ldivprof/run.log
python3 - <<'PY'
import csv,glob,collections
kt=glob.glob("gpurun_out/ldivprof/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(kt))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
# the last solve starts at the last fwd_gather group
starts=[i for i,n in enumerate(names) if "fwd_gather" in n and (i==0 or "fwd_gather" not in names[i-1])]
i0=starts[-1]
sel=rows[i0:]
t0=int(sel[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in sel)
agg=collections.defaultdict(lambda:[0,0.0])
busy=0.0; prev_end=t0; gaps=0.0
for r in sel:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    k=r["Kernel_Name"].split("<")[0].replace("void ","")
    agg[k][0]+=1; agg[k][1]+=(e-s)*1e-3
    if s>prev_end: gaps+=(s-prev_end)*1e-3
    prev_end=max(prev_end,e)
print("last solve: %d launches, wall %.2f ms, idle gaps between kernels %.2f ms"%(len(sel),(t1-t0)*1e-6,gaps*1e-3))
for k,(n,us) in sorted(agg.items(),key=lambda kv:-kv[1][1]): print("  %-32s %5d launches %9.1f us  avg %7.1f us"%(k,n,us,us/n))
# the forward launches of the last solve in order: duration of each (shows the per-level chains)
fw=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-3 for r in sel if "fwd_wide" in r["Kernel_Name"]]
gx=[int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]) if "Grid_Size_X" in r else 0 for r in sel if "fwd_wide" in r["Kernel_Name"]]
print("fwd_wide durations (us) in launch order, with workgroups in x:")
print(" ".join("%.0f/%d"%(a,b) for a,b in zip(fw,gx)))
PY
