#!/bin/bash
# compare fill kernels / windows on one workload (dev tool): args = workload, then "fill threads window" triples
WL=${1:-tree_medium}; shift
mkdir -p gpurun_out
for cfg in "$@"; do
  set -- $cfg; fill=$1; th=$2; win=$3; wu=${4:-0}; wv=${5:-9}
  tag=${fill}_${th}_${win}_${wu}_${wv}
  timeout -k 10 400 python bench.py --workload $WL --steps 1 --warmup $wu --no-cpu-baseline --fill $fill --threads $th --window $win --waves $wv > gpurun_out/var_$tag.json 2>gpurun_out/var_$tag.err || echo FAIL $tag
  python - <<PY
import json; d=json.load(open("gpurun_out/var_$tag.json")); c=d["config"]; r=d["roofline"]
print("fill $fill thr $th win $win warm $wu waves $wv", "value %.3e"%d["value"], "ms/step %.0f"%d["ms_per_step"], "fill avg ms %.3f"%r["avg_launch_ms"], "launches", r["launches"], "recomp", c["rows_recomputed"], "changed", c["rows_changed"], "rej", c["reject_reason"], "frac %.4f"%r["frac"], c["score_after"])
PY
  grep debug: gpurun_out/var_$tag.err
done
