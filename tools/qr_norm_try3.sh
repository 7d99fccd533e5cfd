#!/bin/bash
# usage (GPU box): tools/qr_norm_try3.sh -- HS_QR_ORDER=norm at HS_QR_THETA in $THETAS: the module-level tolerance sweep, then one bench flow
mkdir -p gpurun_out
for th in ${THETAS:-0.5 1.0}; do
  echo "== theta $th"
  HS_QR_THETA=$th HS_QR_ORDER=norm HS_QR_TIMING=1 PROBE_KINDS=cplx-2d PROBE_TOLS=1e-2,1e-4,1e-8,1e-10,1e-12,1e-14 timeout -k 10 300 python tools/qr_floor_probe.py 2048 > gpurun_out/qr_probe_norm_$th.txt 2>&1 || exit 1
  grep -E "^cplx|^real" gpurun_out/qr_probe_norm_$th.txt
  grep "windows" gpurun_out/qr_probe_norm_$th.txt | tail -2
  HS_QR_THETA=$th HS_QR_ORDER=norm MF=0 tools/mf_one.sh poisson3d_128 4 1e-4 || exit 1
done
