# dev (GPU box): where the HOST's time goes per batch (enqueue of 8 launches / waiting for the header), whole MSA and one section
one() {
  python3 bench.py --steps 8 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/host.json 2> gpurun_out/host.err || { echo "$* FAILED"; tail -3 gpurun_out/host.err; return; }
  python3 -c "
import json; d=json.load(open('gpurun_out/host.json')); o=d['config']['options']; b=d['config']['batches']
print('$*', 'ms/step %.1f value %.3e launch_ms %.4f batches %d us/batch %.1f' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], b, d['ms_per_step']*d['steps']*1e3/b), 'host enqueue_us', o.get('host_enqueue_us'), 'wait_us', o.get('host_wait_us'))" | tee -a gpurun_out/r4_host.log
}
one
one --sections 6 --only-section 2
one --sections 6 --only-section 2 --window 1
one --sections 6
