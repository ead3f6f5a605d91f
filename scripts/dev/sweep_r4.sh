# dev (GPU box): bench over option sets given as lines on stdin-free list below
run() {
  o="$1"
  args=""; for kv in $o; do case $kv in waves=*) args="$args --waves ${kv#waves=}";; window=*) args="$args --window ${kv#window=}";; *) args="$args --opt $kv";; esac; done
  python3 bench.py --steps ${STEPS:-8} --warmup ${WARM:-3} --no-cpu-baseline $args > gpurun_out/r4_sw.json 2> gpurun_out/r4_sw.err || { echo "$o FAILED"; tail -3 gpurun_out/r4_sw.err; return; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r4_sw.json')); print('$o', 'ms/step %.1f value %.3e launch_ms %.4f useful %.3f cpb %.3f fails %d stalls %d warm %d' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['config']['useful_frac'], d['config']['commits_per_batch'], d['config']['seg_fails'], d['config']['stalls'], d['config']['options']['warm_now']), 'jumped', d['config'].get('rows_jumped'), 'evrate', d['config']['options'].get('evrate_x100'), 'hard', [d['config']['options'].get(k) for k in ('hard_marked','hard_fills','hard_refail')])" | tee -a gpurun_out/r4_sweep.log
}
while read -r line; do [ -n "$line" ] && run "$line"; done < scripts/dev/sweep_r4.txt
