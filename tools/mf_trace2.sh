#!/bin/bash
# usage (GPU box): [ENV=..] tools/mf_trace2.sh TAG [MF] -- kernel trace of a warmed-up matrix-free factorization (the SECOND of two); for the kernels of the
# HSS module (everything but the front kernels): launches, sum of kernel times, union of their busy intervals, and the same per stream
TAG=${1:-x}; MF=${2:-1}
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/mftrace_$TAG; mkdir -p $R/gpurun_out/mftrace_$TAG; cd $R
timeout -k 10 800 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/mftrace_$TAG -- python3 bench.py --workload poisson3d_128 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' --swlevel 4 --tol 1e-4 --mf $MF > gpurun_out/mftrace_$TAG/run.log 2>&1
python3 - $TAG <<'PY'
import csv,glob,collections,sys
tag=sys.argv[1]
kt=glob.glob("gpurun_out/mftrace_%s/**/*kernel_trace.csv"%tag,recursive=True)[0]
rows=[(int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"],r.get("Queue_Id","?")) for r in csv.DictReader(open(kt))]
rows.sort()
# second factorization: after the second init_fronts burst... use the midpoint split by the largest gap in gemm_op activity: take rows in the last 55% of time span that follow the last 'perm_gather' kernel
pg=[i for i,r in enumerate(rows) if "perm_gather" in r[2]]
start=pg[-1] if pg else 0
rows=rows[start:]
front=("gemm_op_kernel","panel_pivot","panel_l21","trsm_inv","laswp","inv256","scatter_kernel","gather_kernel","mark_kernel","init_fronts","fwd_wide","bwd_wide","int_update","fwd_gather","bwd_scatter","mfma_f64","perm_gather")
def union(iv):
    iv=sorted(iv); b=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: b+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    return b+ce-cs
hss=[(s,e) for s,e,k,q in rows if not any(f in k for f in front)]
allk=[(s,e) for s,e,k,q in rows]
print("[%s] second factorization: %d launches, span %.1f ms; HSS-module kernels: %d launches, sum %.1f ms, union %.1f ms; all kernels union %.1f ms"%(tag,len(rows),(rows[-1][1]-rows[0][0])*1e-6,len(hss),sum(e-s for s,e in hss)*1e-6,union(hss)*1e-6,union(allk)*1e-6))
agg=collections.defaultdict(lambda:[0,0.0])
for s,e,k,q in rows:
    if any(f in k for f in front): continue
    k=k.split("(")[0].replace("void ","")[:50]
    agg[k][0]+=1; agg[k][1]+=(e-s)*1e-6
for k,(n,ms) in sorted(agg.items(),key=lambda kv:-kv[1][1])[:12]: print("   %-50s %7d %9.1f ms avg %7.1f us"%(k,n,ms,1e3*ms/n))
PY
rm -rf gpurun_out/mftrace_$TAG/*/
