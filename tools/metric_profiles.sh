#!/bin/bash
# usage (GPU box): tools/metric_profiles.sh [ROUND] -- the metric's own workload (BASELINE.json: 3-D Helmholtz, fixed HSS tol): kernel table
# (rocprofv3 --kernel-trace --stats), level trace, and the bench line with its own roofline.  Outputs under gpurun_out/ (copied into profiles/ by hand).
R=$GRAFT_REPO_ROOT; RT=${1:-r03}; W=${W:-helmholtz3d_112}; SW=${SW:-4}; TOL=${TOL:-1e-4}
ARGS="--workload $W --swlevel $SW --tol $TOL --no-cpu-baseline --no-oneshot --metric-workload '' ${EXTRA:-}"
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_metric; mkdir -p $R/gpurun_out/prof_metric; cd $R
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_metric -- python3 bench.py --workload $W --swlevel $SW --tol $TOL --no-cpu-baseline --no-oneshot --metric-workload "" --no-profile --steps 2 --warmup 1 ${EXTRA:-} > gpurun_out/prof_metric/bench.log 2>&1 || { tail -5 gpurun_out/prof_metric/bench.log; exit 1; }
DB=$(find gpurun_out/prof_metric -name "*results.db" | head -1)
python3 tools/rocpd_top_kernels.py $DB gpurun_out/${RT}_${W}_tol${TOL}_kernel_stats.csv
grep '^{' gpurun_out/prof_metric/bench.log > gpurun_out/${RT}_${W}_tol${TOL}_bench_under_rocprof.json
rm -rf gpurun_out/prof_metric
head -14 gpurun_out/${RT}_${W}_tol${TOL}_kernel_stats.csv | cut -c1-170
HS_VERBOSE_LEVELS=1 timeout -k 10 600 python bench.py --workload $W --swlevel $SW --tol $TOL --no-cpu-baseline --no-oneshot --metric-workload "" --steps 1 --warmup 1 ${EXTRA:-} > gpurun_out/${RT}_${W}_tol${TOL}_bench.json 2> gpurun_out/${RT}_${W}_tol${TOL}_level_trace.txt || { tail -5 gpurun_out/${RT}_${W}_tol${TOL}_level_trace.txt; exit 1; }
grep "^\[hs\] level" gpurun_out/${RT}_${W}_tol${TOL}_level_trace.txt | tail -12
tail -c 2500 gpurun_out/${RT}_${W}_tol${TOL}_bench.json
