#!/bin/bash
# PMC passes over the InitialAligner measurement (run on the GPU box): SQ issue counters and HBM write/fetch traffic per kernel.
#   usage: ia_pmc.sh <tag> [ia_bench args]   -> gpurun_out/iapmc_<tag>_{sq,write,fetch}.json
TAG=$1; shift
ARGS="$@"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run_pass() {   # name, counters
  rm -rf gpurun_out/iapmc_$1
  timeout -k 10 600 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/iapmc_$1 -- python3 scripts/ia_bench.py --cpu-reads 0 --repeats 1 $ARGS > gpurun_out/iapmc_$1.json 2> gpurun_out/iapmc_$1.err || { echo "pass $1 failed"; tail -3 gpurun_out/iapmc_$1.err; return 1; }
  python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob("gpurun_out/iapmc_$1/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
out = {}
for k in agg:
    out[k] = dict(agg[k]); out[k]["dispatches"] = max(n[(k, c)] for c in agg[k])
bench = json.load(open("gpurun_out/iapmc_$1.json"))
json.dump({"command": "python3 scripts/ia_bench.py --cpu-reads 0 --repeats 1 $ARGS (under rocprofv3 --pmc $2; two pia_align calls)", "kernels": out,
           "bench": {k: bench[k] for k in ("reads", "cells", "template", "last_align_ms")}}, open("gpurun_out/iapmc_${TAG}_$1.json", "w"), indent=1)
for k in out:
    print("$1", k, {c: "%.4g" % v for c, v in out[k].items()})
PY
  rm -rf gpurun_out/iapmc_$1
}
run_pass sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" && run_pass write WRITE_SIZE && run_pass fetch FETCH_SIZE
