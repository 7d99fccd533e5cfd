#!/bin/bash
# usage (GPU box): tools/mf_runs.sh WORKLOAD SWLEVEL TOL -- the compressed paths side by side: dense S between fronts / matrix-free HSS hand-over
W=${1:-poisson3d_64}; SW=${2:-4}; TOL=${3:-1e-2}
for mode in "" "--mf"; do
  HS_VERBOSE_COMPRESS=${HS_VERBOSE_COMPRESS:-} timeout -k 10 500 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' --swlevel $SW --tol $TOL $mode > gpurun_out/mf_${W}_${TOL}_${mode#--}.log 2>&1 || { tail -20 gpurun_out/mf_${W}_${TOL}_${mode#--}.log; exit 1; }
  python - <<PY
import json
l=[x for x in open("gpurun_out/mf_${W}_${TOL}_${mode#--}.log") if x.startswith("{")][-1]
j=json.loads(l)
print("$W tol $TOL mode '${mode}': value %.3f s factor %.3f s residual %.2e maxrank %d bytes %.1f GiB solve %.1f ms"%(j["value"],j["factor_s"],j["residual"],j["maxrank"],j["bytes_factors_GiB"],j["solve"]["seconds"]*1e3))
PY
done
