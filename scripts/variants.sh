#!/bin/bash
# compare fill-kernel geometries / windows on one workload (dev tool)
WL=${1:-tree_medium}
mkdir -p gpurun_out
for cfg in "256 1" "256 8" "256 32" "256 128" "512 32" "128 32" "64 64"; do
  set -- $cfg; th=$1; win=$2
  timeout -k 10 200 python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline --threads $th --window $win > gpurun_out/var_${th}_$win.json 2>gpurun_out/var_${th}_$win.err || echo FAIL $th $win
  python - <<PY
import json; d=json.load(open("gpurun_out/var_${th}_$win.json")); c=d["config"]; r=d["roofline"]
print("thr $th win $win", "value %.3e"%d["value"], "ms/step %.0f"%d["ms_per_step"], "fill avg ms %.3f"%r["avg_launch_ms"], "launches", r["launches"], "recomp", c["rows_recomputed"], "clk", c["shader_clock_mhz_last_fill"], "frac %.4f"%r["frac"], c["score_after"])
PY
done
