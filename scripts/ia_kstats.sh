#!/bin/bash
# per-kernel time of the InitialAligner measurement (dev tool, run on the GPU box): ia_kstats.sh <tag> [ia_bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/iaks_$TAG
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/iaks_$TAG -- python3 scripts/ia_bench.py --cpu-reads 0 "$@" > gpurun_out/iaks_$TAG.json 2> gpurun_out/iaks_$TAG.err
cp gpurun_out/iaks_$TAG/*/*kernel_stats.csv gpurun_out/iaks_${TAG}_kernel_stats.csv
cp gpurun_out/iaks_$TAG/*/*kernel_trace.csv gpurun_out/iaks_${TAG}_kernel_trace.csv
head -8 gpurun_out/iaks_${TAG}_kernel_stats.csv | cut -c1-160
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/iaks_${TAG}_kernel_trace.csv')))
t0=min(int(r['Start_Timestamp']) for r in rows)
for r in rows[-14:]:
    print(r['Kernel_Name'][:24], 'start %.1f ms  dur %.1f ms' % ((int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6))
PY
rm -rf gpurun_out/iaks_$TAG
cat gpurun_out/iaks_$TAG.json
