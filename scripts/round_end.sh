#!/bin/bash
# measurements committed under profiles/ (run on the GPU box): PMC traffic passes, kernel stats, the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash scripts/pmc.sh tree_medium FETCH_SIZE fetch > gpurun_out/pmc_fetch.log 2>&1
bash scripts/pmc.sh tree_medium WRITE_SIZE write > gpurun_out/pmc_write.log 2>&1
rm -rf gpurun_out/prof_full
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_full -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/prof_full_bench.json 2> gpurun_out/prof_full.err
cp gpurun_out/prof_full/*/*kernel_stats.csv gpurun_out/prof_full_kernel_stats.csv
rm -rf gpurun_out/prof_full/*/*kernel_trace.csv
timeout -k 10 500 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -3 gpurun_out/pmc_fetch.log; tail -3 gpurun_out/pmc_write.log; cat gpurun_out/prof_full_kernel_stats.csv | head -8; cut -c1-400 gpurun_out/bench_default.json
