#!/bin/bash
# per-kernel time of one round (dev tool): kstats.sh <workload> [bench args]
WL=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline "$@" > gpurun_out/ks.json 2> gpurun_out/ks.err
head -6 gpurun_out/ks/*/*kernel_stats.csv | cut -c1-110
rm -rf gpurun_out/ks/*/*kernel_trace.csv
python -c "
import json; d=json.load(open('gpurun_out/ks.json')); print('ms/step %.0f value %.3e score %d' % (d['ms_per_step'], d['value'], d['config']['score_after']))"
