#!/bin/bash
# usage (GPU box): tools/mf_verbose.sh WORKLOAD SWLEVEL TOL -- per-step wall times and ranks of the matrix-free compressed branch
W=${1:-poisson3d_64}; SW=${2:-4}; TOL=${3:-1e-2}
HS_VERBOSE_COMPRESS=1 HS_HSS_VERBOSE=${HS_HSS_VERBOSE:-} timeout -k 10 500 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' --swlevel $SW --tol $TOL --mf ${MF:-1} > gpurun_out/mfv_${W}_${TOL}.log 2>&1
grep -E "^\[hs" gpurun_out/mfv_${W}_${TOL}.log | tail -${LINES_OUT:-80}
