#!/bin/bash
# usage (GPU box): [ENV=..] tools/mf_one.sh WORKLOAD SWLEVEL TOL [extra flags] -- one matrix-free compressed bench run, one summary line
W=${1:-poisson3d_128}; SW=${2:-4}; TOL=${3:-1e-4}; shift 3
timeout -k 10 600 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' --swlevel $SW --tol $TOL --mf ${MF:-1} "$@" > gpurun_out/mf_one.log 2>&1 || { tail -20 gpurun_out/mf_one.log; exit 1; }
python - "$W $SW $TOL $* HS_HSS_LEAF=${HS_HSS_LEAF:-}" <<'PY'
import json,sys
j=json.loads([x for x in open("gpurun_out/mf_one.log") if x.startswith("{")][-1])
print("%s: value %.3f s factor %.3f s residual %.2e maxrank %d bytes %.1f GiB solve %.1f ms"%(sys.argv[1],j["value"],j["factor_s"],j["residual"],j["maxrank"],j["bytes_factors_GiB"],j["solve"]["seconds"]*1e3))
PY
