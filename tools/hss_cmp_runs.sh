for cfg in "1e-2 0" "1e-2 32768" "1e-2 16384" "1e-4 0" "1e-4 32768" "1e-4 16384"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --workload poisson3d_128 --steps 2 --warmup 1 --no-cpu-baseline --no-profile --swlevel 4 --tol $1 --hss-min $2 > gpurun_out/cmp_$1_$2.log 2>&1 || exit 1
  python - <<PY
import json
l=[x for x in open("gpurun_out/cmp_$1_$2.log") if x.startswith("{")][-1]
j=json.loads(l)
print("tol $1 hss_min $2: value %.3f s factor %.3f s residual %.2e maxrank %d solve %.1f ms"%(j["value"],j["factor_s"],j["residual"],j["maxrank"],j["solve"]["seconds"]*1e3))
PY
done
