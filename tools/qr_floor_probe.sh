#!/bin/bash
# usage (GPU box): tools/qr_floor_probe.sh -- module-level tolerance sweep + the verbose matrix-free block flow at 1e-11 (Helmholtz 24^3)
mkdir -p gpurun_out
timeout -k 10 400 python tools/qr_floor_probe.py 2048 > gpurun_out/qr_floor_module.txt 2>&1 || { tail -20 gpurun_out/qr_floor_module.txt; exit 1; }
cat gpurun_out/qr_floor_module.txt
PROBE_TOLS=${PROBE_TOLS:-1e-10,1e-11,1e-12,1e-13} HS_MF_THREADS=1 HS_HSS_VERBOSE=1 timeout -k 10 400 python tools/mf_exactness_probe.py 24x24x24 helmholtz block > gpurun_out/qr_floor_mf.txt 2> gpurun_out/qr_floor_mf.err || { tail -20 gpurun_out/qr_floor_mf.err; exit 1; }
cat gpurun_out/qr_floor_mf.txt
grep -E "max \|T_ij\|" gpurun_out/qr_floor_mf.err | sort -t= -k2 -g -r | head -5
