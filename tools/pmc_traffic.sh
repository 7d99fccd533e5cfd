#!/bin/bash
# usage: tools/pmc_traffic.sh M N K -- HBM traffic of one real GEMM launch: FETCH_SIZE and WRITE_SIZE in SEPARATE passes
# (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units: KiB.
# gfx950: FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced read -> doubled below.
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf $R/gpurun_out/pmct; mkdir -p $R/gpurun_out/pmct; cd $R
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct/f -- python3 tools/one_gemm.py $1 $2 $3 2 > gpurun_out/pmct/f.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmct/w -- python3 tools/one_gemm.py $1 $2 $3 2 > gpurun_out/pmct/w.log 2>&1
grep TF gpurun_out/pmct/f.log
python3 - $1 $2 $3 <<'PY'
import csv,sys,glob
M,N,K=[int(a) for a in sys.argv[1:4]]
def last(pat,name):
    f=glob.glob(pat)[0]; v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "gemm_probs" in r["Kernel_Name"] and r["Counter_Name"]==name]
    return v[-1]
fe=last("gpurun_out/pmct/f/*/*counter_collection.csv","FETCH_SIZE"); wr=last("gpurun_out/pmct/w/*/*counter_collection.csv","WRITE_SIZE")
alg=(M*K+K*N+2*M*N)*8
print("GEMM %dx%dx%d: FETCH_SIZE %.0f KiB (x2 gfx950 correction -> %.3f GB), WRITE_SIZE %.0f KiB (%.3f GB); algorithmic A+B+2C = %.3f GB; tile-level L2 traffic (A,B re-read per 128x128 tile) = %.3f GB"%(M,N,K,fe,2*fe*1024/1e9,wr,wr*1024/1e9,alg/1e9,((M/128)*(N/128)*(128+128)*K*8+2*M*N*8)/1e9))
PY
