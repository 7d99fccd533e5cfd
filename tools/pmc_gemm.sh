#!/bin/bash
# usage: tools/pmc_gemm.sh M N K  -- SQ counters of one real GEMM launch (two passes), printed per counter
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/pmc; mkdir -p $R/gpurun_out/pmc; cd $R
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc/a -- python3 tools/one_gemm.py $1 $2 $3 2 > gpurun_out/pmc/a.log 2>&1
grep TF gpurun_out/pmc/a.log
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d gpurun_out/pmc/b -- python3 tools/one_gemm.py $1 $2 $3 2 > gpurun_out/pmc/b.log 2>&1
for f in $(find gpurun_out/pmc -name "*counter_collection.csv"); do python3 - $f <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(list)
for r in rows:
    if "gemm_probs" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()): print("  %-34s n=%d last=%.4g"%(k,len(v),v[-1]))
PY
done
