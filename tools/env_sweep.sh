#!/bin/bash
# usage (GPU box): tools/env_sweep.sh "VAR=a VAR=b ..." -- the default bench (exact Poisson 128^3, 2 steps) under each environment setting
for kv in "$@"; do
  env $kv timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-oneshot --metric-workload '' > gpurun_out/sweep.log 2>&1 || { tail -5 gpurun_out/sweep.log; exit 1; }
  python - "$kv" <<'PY'
import json,sys
j=json.loads([x for x in open("gpurun_out/sweep.log") if x.startswith("{")][-1])
print("%-28s value %.3f s factor %.3f s residual %.1e"%(sys.argv[1],j["value"],j["factor_s"],j["residual"]))
PY
done
