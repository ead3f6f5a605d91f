#!/bin/bash
# PMC passes for the fill kernel (dev tool). usage: pmc.sh <workload> <fill> "<counters>" <tag>
WL=$1; FILL=$2; CNT=$3; TAG=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$TAG
timeout -k 10 400 rocprofv3 --pmc $CNT --output-format csv -d gpurun_out/pmc_$TAG -- python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --fill $FILL --window 1 > gpurun_out/pmc_$TAG.json 2> gpurun_out/pmc_$TAG.err
python - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_$TAG/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in agg:
    if "fill" in k or "trace" in k or "commit" in k:
        print(k, {c: "%.4g" % v for c, v in agg[k].items()}, "dispatches", max(n[(k, c)] for c in agg[k]))
PY
