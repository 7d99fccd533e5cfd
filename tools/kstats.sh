#!/bin/bash
# usage (GPU box): [ENV=..] tools/kstats.sh TAG WORKLOAD [bench flags] -- rocprofv3 kernel table of one bench run (1 warm-up + 1 step) -> gpurun_out/kstats_TAG.csv
R=$GRAFT_REPO_ROOT; TAG=$1; W=$2; shift 2
cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_$TAG; mkdir -p $R/gpurun_out/prof_$TAG; cd $R
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -- python3 bench.py --workload $W --no-cpu-baseline --no-oneshot --metric-workload "" --no-profile --steps 1 --warmup 1 "$@" > gpurun_out/prof_$TAG/bench.log 2>&1 || { tail -5 gpurun_out/prof_$TAG/bench.log; exit 1; }
DB=$(find gpurun_out/prof_$TAG -name "*results.db" | head -1)
python3 tools/rocpd_top_kernels.py $DB gpurun_out/kstats_$TAG.csv
grep '^{' gpurun_out/prof_$TAG/bench.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('value %.3f s'%j['value'])"
rm -rf gpurun_out/prof_$TAG
head -${LINES_OUT:-12} gpurun_out/kstats_$TAG.csv | cut -c1-150
