#!/bin/bash
# per-kernel time and SQ counters of the MaxCorrelation measurement (dev tool, run on the GPU box): mc_kstats.sh <tag> [mc_bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/mcks_$TAG
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mcks_$TAG -- python3 scripts/mc_bench.py --cpu-columns 0 "$@" > gpurun_out/mcks_$TAG.json 2> gpurun_out/mcks_$TAG.err
grep -E "k_mc_|Name" gpurun_out/mcks_$TAG/*/*kernel_stats.csv > gpurun_out/mcks_${TAG}_kernel_stats.csv
cut -c1-200 gpurun_out/mcks_${TAG}_kernel_stats.csv
rm -rf gpurun_out/mcks_$TAG
cat gpurun_out/mcks_$TAG.json
rm -rf gpurun_out/mcpmc_$TAG
timeout -k 10 900 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/mcpmc_$TAG -- python3 scripts/mc_bench.py --cpu-columns 0 "$@" > /dev/null 2> gpurun_out/mcpmc_$TAG.err
python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob("gpurun_out/mcpmc_$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "k_mc_" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
out = {k: dict(v, dispatches=max(n[(k, c)] for c in v)) for k, v in agg.items()}
json.dump({"command": "python3 scripts/mc_bench.py --cpu-columns 0 $@ (two pmc_maxcorrs calls)", "kernels": out}, open("gpurun_out/mcpmc_${TAG}_sq.json", "w"), indent=1)
for k in out: print(k, {c: "%.4g" % v for c, v in out[k].items()})
PY
rm -rf gpurun_out/mcpmc_$TAG
