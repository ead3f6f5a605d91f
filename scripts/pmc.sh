#!/bin/bash
# PMC passes for the fill kernel (dev tool). usage: pmc.sh <workload> "<counters>" <tag>
WL=$1; CNT=$2; TAG=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_$TAG
timeout -k 10 500 rocprofv3 --pmc $CNT --output-format csv -d gpurun_out/pmc_$TAG -- python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --window 1 > gpurun_out/pmc_$TAG.json 2> gpurun_out/pmc_$TAG.err
python - <<PY
import csv, glob, collections, json
f = glob.glob("gpurun_out/pmc_$TAG/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
out = {}
for k in agg:
    out[k] = {c: v for c, v in agg[k].items()}
    out[k]["dispatches"] = max(n[(k, c)] for c in agg[k])
    print(k, {c: "%.4g" % v for c, v in agg[k].items()}, "dispatches", out[k]["dispatches"])
json.dump(out, open("gpurun_out/pmc_$TAG.summary.json", "w"), indent=1)
PY
