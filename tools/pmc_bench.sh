#!/bin/bash
# usage (on the GPU box): tools/pmc_bench.sh [workload] -- HBM traffic of the dominant kernel over ONE bench step.
# FETCH_SIZE and WRITE_SIZE are collected in SEPARATE rocprofv3 passes with --kernel-trace only (TCC has 4 counter
# slots: FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units of both: KiB.
# gfx950 correction: FETCH_SIZE reports exactly half of the bytes of wide coalesced reads -> doubled (same guide).
W=${1:-poisson3d_128}
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmcb; mkdir -p $R/gpurun_out/pmcb
cd $R
HS_PROGRESS=1 timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gemm_op_kernel" --kernel-trace --output-format csv -d gpurun_out/pmcb/f -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' > gpurun_out/pmcb/f.log 2>&1 || { tail -5 gpurun_out/pmcb/f.log; exit 1; }
HS_PROGRESS=1 timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "gemm_op_kernel" --kernel-trace --output-format csv -d gpurun_out/pmcb/w -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' > gpurun_out/pmcb/w.log 2>&1 || { tail -5 gpurun_out/pmcb/w.log; exit 1; }
python3 - $W <<'PY'
import csv, glob, json, sys
def collect(pat, name):
    f = glob.glob(pat)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "gemm_op_kernel" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
fe, nf = collect("gpurun_out/pmcb/f/*/*counter_collection.csv", "FETCH_SIZE")
wr, nw = collect("gpurun_out/pmcb/w/*/*counter_collection.csv", "WRITE_SIZE")
out = dict(workload=sys.argv[1], kernel="gemm_op_kernel", launches=nf, launches_write_pass=nw,
           fetch_bytes_total=2.0 * fe * 1024, write_bytes_total=wr * 1024,
           traffic_bytes_per_launch=(2.0 * fe * 1024 + wr * 1024) / max(nf, 1),
           note="one numeric factorization; FETCH_SIZE (KiB) doubled per the gfx950 correction of MI355X_MICROARCH.md; separate --pmc passes with --kernel-trace only")
json.dump(out, open("gpurun_out/pmcb/pmc_traffic.json", "w"), indent=1)
json.dump(out, open("gpurun_out/r03_%s_gemm_pmc_traffic.json" % sys.argv[1], "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/pmcb/f gpurun_out/pmcb/w
