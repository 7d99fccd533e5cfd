#!/bin/bash
# usage (GPU box): tools/qr_norm_try2.sh -- module-level tolerance sweep and the matrix-free test file, default order against HS_QR_ORDER=norm
mkdir -p gpurun_out
for ord in lu norm; do
  echo "== order $ord"
  HS_QR_ORDER=$ord PROBE_TOLS=1e-2,1e-3,1e-4,1e-6,1e-8,1e-10,1e-12,1e-14 timeout -k 10 300 python tools/qr_floor_probe.py 2048 2>&1 | tee gpurun_out/qr_probe_$ord.txt || exit 1
done
HS_QR_ORDER=norm timeout -k 10 900 python -m pytest tests/test_mf_gpu.py tests/test_compressed_gpu.py tests/test_hss_gpu.py tests/test_lowrank_gpu.py -m gpu -q > gpurun_out/qr_norm_tests.txt 2>&1
echo "tests under HS_QR_ORDER=norm: exit $?"; grep -E "^E  |passed|failed|FAILED" gpurun_out/qr_norm_tests.txt | head -40
