#!/bin/bash
# usage (GPU box): tools/qr_sweep.sh -- the module probe (complex case) under several HS_CHOL_COND / HS_NOISE_REL settings
for cfg in "1e-3 1e-13" "1e-2 1e-13" "1e-1 1e-13" "1e-3 1e-12" "1e-5 1e-13"; do
  set -- $cfg
  echo "== HS_CHOL_COND=$1 HS_NOISE_REL=$2"
  HS_CHOL_COND=$1 HS_NOISE_REL=$2 PROBE_KINDS=${PROBE_KINDS:-cplx-2d} PROBE_TOLS=${PROBE_TOLS:-1e-10,1e-11,1e-12,1e-13} timeout -k 10 200 python tools/qr_floor_probe.py 2048 2>&1 | grep -v amdgpu.ids
done
echo "== verbose 1e-13 default"
HS_HSS_VERBOSE=1 PROBE_KINDS=cplx-2d PROBE_TOLS=1e-13 timeout -k 10 200 python tools/qr_floor_probe.py 2048 2>&1 | grep -E "rank max|T_ij|hssrank" | tail -24
