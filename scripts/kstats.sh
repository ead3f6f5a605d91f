#!/bin/bash
# per-kernel time of the bench command (dev tool, run on the GPU box): kstats.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks_$TAG
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$TAG -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/ks_$TAG.json 2> gpurun_out/ks_$TAG.err
cp gpurun_out/ks_$TAG/*/*kernel_stats.csv gpurun_out/ks_${TAG}_kernel_stats.csv
head -8 gpurun_out/ks_${TAG}_kernel_stats.csv | cut -c1-120
rm -rf gpurun_out/ks_$TAG
python3 -c "
import json; d=json.load(open('gpurun_out/ks_$TAG.json')); print('ms/step %.0f value %.3e frac %.5f avg_launch_ms %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms']))"
