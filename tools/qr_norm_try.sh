#!/bin/bash
# usage (GPU box): tools/qr_norm_try.sh -- the HS_QR_ORDER=norm switch (windows of qr_refine chosen by downdated residual norms, no pivoted LU of
# a sketch) against the default order: the compression test files under the switch, then ranks / errors / times of three bench flows both ways
mkdir -p gpurun_out
HS_QR_ORDER=norm timeout -k 10 900 python -m pytest tests/test_hss_gpu.py tests/test_lowrank_gpu.py tests/test_mf_gpu.py tests/test_compressed_gpu.py -m gpu -x -q > gpurun_out/qr_norm_tests.txt 2>&1
echo "tests under HS_QR_ORDER=norm: exit $?"; tail -5 gpurun_out/qr_norm_tests.txt
for ord in lu norm; do
  echo "== order $ord"
  HS_QR_ORDER=$ord MF=1 tools/mf_one.sh poisson3d_128 4 1e-4 || exit 1
  HS_QR_ORDER=$ord MF=0 tools/mf_one.sh poisson3d_128 4 1e-4 || exit 1
  HS_QR_ORDER=$ord MF=0 tools/mf_one.sh helmholtz3d_112 4 1e-4 || exit 1
done
