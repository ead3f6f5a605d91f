# dev (GPU box): the bench with the round-3 library and with the current one, on the SAME box (devices differ by several percent)
for lib in repeatresolver_amd/csrc/libpwr_r3.so repeatresolver_amd/csrc/libpwr.so; do
  python3 scripts/dev/with_lib.py $lib bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$lib', 'ms/step %.1f value %.3e launch_ms %.4f batches %d cpb %.3f' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['config']['batches'], d['config']['commits_per_batch']))" | tee -a gpurun_out/r4_ab.log
done
