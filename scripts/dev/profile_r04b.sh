# dev (GPU box): the profiles of the round's last build (the sections run apart)
set -x
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_driver_cmd.json 2> gpurun_out/r04_bench_driver_cmd.err || exit 1
bash scripts/kstats.sh r04 --steps 20 --warmup 5 || exit 1
bash scripts/pmc_bench.sh r04 --steps 20 --warmup 5 || exit 1
python3 scripts/converge.py tree_default 12 > gpurun_out/r04_converge.log 2>&1 || exit 1
python3 scripts/pipeline_converge.py tree_default /tmp/pc > gpurun_out/r04_pipeline.json 2> gpurun_out/r04_pipeline.err || exit 1
tail -1 gpurun_out/r04_converge.log | cut -c1-200
