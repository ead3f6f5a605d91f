# dev (GPU box): ONE Window.py section of the benchmark MSA with the GPU to itself, and all six side by side, for a few plans
one() {
  python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline --sections 6 "$@" > gpurun_out/sec.json 2> gpurun_out/sec.err || { echo "$* FAILED"; tail -3 gpurun_out/sec.err; return; }
  python3 -c "
import json; d=json.load(open('gpurun_out/sec.json')); print('$*', 'ms/step %.1f value %.3e launch_ms %.4f batches %d cpb %.3f useful %.3f' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['config']['batches'], d['config']['commits_per_batch'], d['config']['useful_frac']))" | tee -a gpurun_out/r4_sections_sweep.log
}
one --only-section 2 --window 1
one --only-section 2 --window 2
one --window 1
one --window 2
one --window 1 --opt seg_rows=96
