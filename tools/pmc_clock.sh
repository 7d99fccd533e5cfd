#!/bin/bash
# usage: tools/pmc_clock.sh M N K -- effective shader clock and true MFMA-pipe occupancy of one GEMM launch:
# clock = GRBM_GUI_ACTIVE / 8 / kernel time (MI355X_MICROARCH.md, "DVFS give-back"); busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/pmcclk; mkdir -p $R/gpurun_out/pmcclk; cd $R
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcclk/a -- python3 tools/one_gemm.py $1 $2 $3 4 > gpurun_out/pmcclk/a.log 2>&1
grep TF gpurun_out/pmcclk/a.log
python3 - <<'PY'
import csv,glob,collections
cc=glob.glob("gpurun_out/pmcclk/a/**/*counter_collection.csv",recursive=True)[0]
kt=glob.glob("gpurun_out/pmcclk/a/**/*kernel_trace.csv",recursive=True)[0]
dur={}
for r in csv.DictReader(open(kt)):
    if "gemm_probs" in r["Kernel_Name"]: dur[r["Dispatch_Id"]]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))*1e-9
agg=collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "gemm_probs" in r["Kernel_Name"]: agg[r["Dispatch_Id"]][r["Counter_Name"]]=float(r["Counter_Value"])
for d,c in agg.items():
    t=dur.get(d)
    if not t: continue
    cyc=c["GRBM_GUI_ACTIVE"]/8
    print("dispatch %s: %.3f ms  clock %.3f GHz  MFMA busy %.1f %% of SIMD-cycles at that clock  (peak at that clock %.1f TF/s)"%(d,t*1e3,cyc/t/1e9,100*c["SQ_VALU_MFMA_BUSY_CYCLES"]/(1024*cyc),cyc/t*1024*32/1e12))
PY
