#!/bin/bash
# usage (GPU box): tools/factor_trace.sh [WORKLOAD] -- kernel trace of ONE numeric factorization (exact path): per batch size (= tree level: grid.y of the
# grouped kernels) the wall time, the summed time of every kernel and the average launch; the csv stays in gpurun_out/ftrace_levels.txt
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/ftrace; mkdir -p $R/gpurun_out/ftrace; cd $R
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ftrace -- python3 bench.py --workload ${1:-poisson3d_128} --no-cpu-baseline --no-oneshot --metric-workload "" --steps 1 --warmup 0 --no-profile > gpurun_out/ftrace/run.log 2>&1 || { tail -5 gpurun_out/ftrace/run.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/ftrace_levels.txt
import csv,glob,collections
kt=glob.glob("gpurun_out/ftrace/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(kt))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
def short(n): return n.split("<")[0].replace("void ","").split("(")[0]
# one factorization = from the first init_fronts to the first fwd_gather
names=[short(r["Kernel_Name"]) for r in rows]
i0=next(i for i,n in enumerate(names) if n=="init_fronts_kernel")
i1=next(i for i,n in enumerate(names) if n=="fwd_gather_kernel" and i>i0)
sel=rows[i0:i1]
lev=collections.OrderedDict()
cur=None
for r in sel:
    n=short(r["Kernel_Name"])
    gy=int(r.get("Grid_Size_Y",1))//max(int(r.get("Workgroup_Size_Y",1)),1)
    if n=="init_fronts_kernel":
        cur=int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]) if False else None
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if n=="mark_kernel": cur=gy; lev.setdefault(cur,{"t0":s,"t1":e,"k":collections.defaultdict(lambda:[0,0.0])})
    if cur is None: continue
    L=lev[cur]; L["t1"]=max(L["t1"],e); L["k"][n][0]+=1; L["k"][n][1]+=(e-s)*1e-3
for b,L in lev.items():
    print("fronts %4d: wall %8.2f ms"%(b,(L["t1"]-L["t0"])*1e-6))
    for k,(n,us) in sorted(L["k"].items(),key=lambda kv:-kv[1][1])[:8]: print("      %-28s %5d launches %10.1f us  avg %8.1f us"%(k,n,us,us/n))
PY
rm -rf gpurun_out/ftrace
