#!/bin/bash
# usage (GPU box): tools/mf_trace.sh WORKLOAD SWLEVEL TOL [MF] -- kernel trace of ONE matrix-free factorization; top kernels by total time, launches, GPU-busy share
W=${1:-poisson3d_128}; SW=${2:-4}; TOL=${3:-1e-4}; MF=${4:-1}
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/mftrace; mkdir -p $R/gpurun_out/mftrace; cd $R
timeout -k 10 800 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mftrace -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' --swlevel $SW --tol $TOL --mf $MF > gpurun_out/mftrace/run.log 2>&1
python3 - <<'PY'
import csv,glob,collections
kt=glob.glob("gpurun_out/mftrace/**/*kernel_trace.csv",recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in csv.DictReader(open(kt))]
rows.sort()
agg=collections.defaultdict(lambda:[0,0.0])
for s,e,k in rows:
    k=k.split("(")[0].replace("void ","")[:70]
    agg[k][0]+=1; agg[k][1]+=(e-s)*1e-6
tot=sum(v[1] for v in agg.values())
# union of busy intervals
busy=0; cur_s,cur_e=rows[0][0],rows[0][1]
for s,e,_ in rows[1:]:
    if s>cur_e: busy+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("launches %d, sum of kernel times %.1f ms, GPU busy (union) %.1f ms, span %.1f ms"%(len(rows),tot,busy*1e-6,(rows[-1][1]-rows[0][0])*1e-6))
for k,(n,ms) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:40]: print("  %-70s %7d  %9.1f ms  avg %8.1f us"%(k,n,ms,1e3*ms/n))
PY
