#!/bin/bash
# usage (GPU box): tools/final_profiles.sh -- the judged artefacts of a round: PMC traffic of the GEMM kernel, the kernel table of the default bench,
# the default bench line.  Outputs under gpurun_out/ (copied into profiles/ by hand).
R=$GRAFT_REPO_ROOT
tools/pmc_bench.sh poisson3d_128 > gpurun_out/final_pmc.log 2>&1 || { tail -5 gpurun_out/final_pmc.log; exit 1; }
tail -1 gpurun_out/final_pmc.log | cut -c1-400
cp gpurun_out/r03_poisson3d_128_gemm_pmc_traffic.json profiles/r03_poisson3d_128_gemm_pmc_traffic.json  # the bench line below reports it (same head, same launch count)
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_final; mkdir -p $R/gpurun_out/prof_final; cd $R
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_final -- python3 bench.py --no-cpu-baseline --no-oneshot --metric-workload "" --steps 3 --warmup 1 > gpurun_out/prof_final/bench.log 2>&1 || { tail -5 gpurun_out/prof_final/bench.log; exit 1; }
DB=$(find gpurun_out/prof_final -name "*results.db" | head -1)
python3 tools/rocpd_top_kernels.py $DB gpurun_out/final_kernel_stats.csv
rm -rf gpurun_out/prof_final
head -8 gpurun_out/final_kernel_stats.csv | cut -c1-160
timeout -k 10 900 python bench.py --steps 3 --warmup 1 > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err || { tail -5 gpurun_out/final_bench_default.err; exit 1; }
tail -c 1500 gpurun_out/final_bench_default.json
