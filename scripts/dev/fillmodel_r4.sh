# dev (GPU box): what the fill launch's duration is made of -- one section alone, one job per batch, plans varied
one() {
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --sections 6 --only-section 2 --window 1 "$@" > gpurun_out/fm.json 2> gpurun_out/fm.err || { echo "$* FAILED"; tail -3 gpurun_out/fm.err; return; }
  python3 -c "
import json; d=json.load(open('gpurun_out/fm.json')); c=d['config']
print('$*', 'launch_ms %.4f ms/step %.1f segs/job %.1f fails %d jobs %d useful %.3f warm %s' % (d['roofline']['avg_launch_ms'], d['ms_per_step'], c['segs']/max(1,c['seg_jobs']), c['seg_fails'], c['seg_jobs'], c['useful_frac'], c['options'].get('warm_now')))" | tee -a gpurun_out/r4_fillmodel.log
}
one --opt warm_adapt=0 --opt warm_pct=190
one --opt warm_adapt=0 --opt warm_pct=190 --waves 9 --opt wave_cols=4
one --opt warm_adapt=0 --opt warm_pct=190 --waves 9
one --opt warm_adapt=0 --opt warm_pct=190 --waves 8
one --opt warm_adapt=0 --opt warm_pct=190 --waves 17
one --opt warm_adapt=0 --opt warm_pct=190 --waves 4
one --opt seg_rows=0
one --opt seg_rows=0 --waves 9 --opt wave_cols=4
one --opt seg_rows=0 --waves 9
one --opt seg_rows=0 --waves 17
