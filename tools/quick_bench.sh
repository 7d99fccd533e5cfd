#!/bin/bash
# usage (GPU box): [ENV=..] tools/quick_bench.sh [bench flags] -- one exact-factorization bench run without the side measurements, one summary line
timeout -k 10 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-oneshot --metric-workload '' "$@" > gpurun_out/quick_bench.log 2>&1 || { tail -20 gpurun_out/quick_bench.log; exit 1; }
python - "$*" <<'PY'
import json,sys
j=json.loads([x for x in open("gpurun_out/quick_bench.log") if x.startswith("{")][-1])
r=j.get("roofline") or {}
print("[%s] value %.3f s factor %.3f s solve %.1f ms (%.0f GB/s) residual %.2e gemm %.1f TF/s frac %.3f"%(sys.argv[1],j["value"],j["factor_s"],j["solve"]["seconds"]*1e3,j["solve"]["achieved_GBps"],j["residual"],r.get("achieved",0),r.get("frac",0)))
PY
