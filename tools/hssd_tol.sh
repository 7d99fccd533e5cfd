#!/bin/bash
# usage (GPU box): tools/hssd_tol.sh WORKLOAD -- residual of the plain ldiv! against the tolerance with D of the root kept as HSS (hs_options.hss_d)
W=${1:-poisson3d_128}; HM=${2:-32768}
for tol in 1e-2 1e-4 1e-6; do
  for hm in 0 $HM; do
    timeout -k 10 500 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' --swlevel 4 --tol $tol --hss-min $hm > gpurun_out/hssd.log 2>&1 || { tail -5 gpurun_out/hssd.log; exit 1; }
    python - <<PY
import json
j=json.loads([x for x in open("gpurun_out/hssd.log") if x.startswith("{")][-1])
print("$W tol $tol hss_min $hm: value %.3f s residual %.2e maxrank %d solve %.1f ms"%(j["value"],j["residual"],j["maxrank"],j["solve"]["seconds"]*1e3))
PY
  done
done
