#!/bin/bash
# usage (GPU box): tools/cmp_verbose.sh WORKLOAD SWLEVEL TOL [extra bench flags] -- per-step wall times of the compressed levels
W=${1:-poisson3d_128}; SW=${2:-4}; TOL=${3:-1e-2}; shift 3
HS_VERBOSE_COMPRESS=1 timeout -k 10 500 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' --swlevel $SW --tol $TOL "$@" > gpurun_out/cmpv.log 2>&1
grep -E "^\[hs compress\]|^\{" gpurun_out/cmpv.log | tail -${LINES_OUT:-40} | cut -c1-400
