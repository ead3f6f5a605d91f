#!/bin/bash
# PMC passes over the bench command (run on the GPU box): HBM traffic (FETCH_SIZE and WRITE_SIZE need separate passes,
# MI355X_MICROARCH.md "rocprofv3 PMC slots") and SQ issue counters, summarised per kernel.
#   usage: pmc_bench.sh <tag> [bench args]      -> gpurun_out/pmc_<tag>_{fetch,write,sq}.json, gpurun_out/bench_traffic_<tag>.json
TAG=$1; shift
ARGS="$@"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run_pass() {   # name, counters
  rm -rf gpurun_out/pmcb_$1
  timeout -k 10 900 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/pmcb_$1 -- python3 bench.py --no-cpu-baseline $ARGS > gpurun_out/pmcb_$1.json 2> gpurun_out/pmcb_$1.err || { echo "pass $1 failed"; tail -3 gpurun_out/pmcb_$1.err; return 1; }
  python3 - <<PY
import csv, glob, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in glob.glob("gpurun_out/pmcb_$1/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
out = {}
for k in agg:
    out[k] = dict(agg[k]); out[k]["dispatches"] = max(n[(k, c)] for c in agg[k])
bench = json.load(open("gpurun_out/pmcb_$1.json"))
json.dump({"command": "python3 bench.py --no-cpu-baseline $ARGS (under rocprofv3 --pmc $2)", "kernels": out,
           "bench": {"ms_per_step": bench["ms_per_step"], "fill_launches": bench["roofline"]["launches"], "cells_computed": bench["roofline"]["cells_computed"]}},
          open("gpurun_out/pmc_${TAG}_$1.json", "w"), indent=1)
for k in sorted(out, key=lambda k: -sum(v for c, v in out[k].items() if c != "dispatches"))[:5]:
    print("$1", k, {c: "%.4g" % v for c, v in out[k].items()})
PY
  rm -rf gpurun_out/pmcb_$1
}
run_pass fetch FETCH_SIZE && run_pass write WRITE_SIZE && run_pass sq "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" && python3 - <<PY
import json
f = json.load(open("gpurun_out/pmc_${TAG}_fetch.json")); w = json.load(open("gpurun_out/pmc_${TAG}_write.json"))
key = [k for k in f["kernels"] if "k_fill_v3" in k][0]
fk, wk = f["kernels"][key], w["kernels"][key]
# both counters are in KB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM): doubled
fetch_b = 2.0 * fk["FETCH_SIZE"] * 1024.0 / fk["dispatches"]
write_b = wk["WRITE_SIZE"] * 1024.0 / wk["dispatches"]
out = {"hbm_bytes_per_launch": fetch_b + write_b, "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
       "kernel": key, "dispatches": fk["dispatches"], "command": "bench.py $ARGS",
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over the bench command; KB -> bytes; FETCH_SIZE doubled "
                 "(gfx950 tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md); per launch = sum over the kernel's dispatches / dispatches",
       "cells_per_launch": f["bench"]["cells_computed"] / max(1, f["bench"]["fill_launches"])}
out["hbm_bytes_per_cell"] = out["hbm_bytes_per_launch"] / out["cells_per_launch"]
json.dump(out, open("gpurun_out/bench_traffic_${TAG}.json", "w"), indent=1)
print(out)
PY
