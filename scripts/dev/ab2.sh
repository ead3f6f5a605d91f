# dev (GPU box): the bench with two builds of the library on the SAME box, alternating:  ab2.sh <libA.so> <libB.so> [bench args]
A=$1; B=$2; shift 2
for lib in $A $B $A $B; do
  python3 scripts/dev/with_lib.py $lib bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab.err; continue; }
  python3 -c "
import json; d=json.load(open('gpurun_out/ab.json')); print('$lib', 'ms/step %.1f value %.3e launch_ms %.4f batches %d cpb %.3f' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['config']['batches'], d['config']['commits_per_batch']))" | tee -a gpurun_out/r4_ab2.log
done
