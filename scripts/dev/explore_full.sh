#!/bin/bash
# dev (GPU box): short bench runs at full scale over a list of option sets; one json line each into gpurun_out/explore.jsonl
out=gpurun_out/explore.jsonl
: > $out
while read -r opts; do
  [ -z "$opts" ] && continue
  echo "== $opts" >&2
  timeout -k 10 300 python3 bench.py --steps ${STEPS:-4} --warmup ${WARM:-1} --no-cpu-baseline $opts 2> gpurun_out/explore.err | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']; r = d['roofline']
    print(json.dumps({'opts': '$opts', 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'avg_launch_ms': r['avg_launch_ms'], 'batches': c['batches'], 'cpb': c['commits_per_batch'], 'useful': c['useful_frac'], 'seg_fails': c['seg_fails'], 'segs': c['segs'], 'stalls': c['stalls'], 'score_after': c['score_after']}))
" | tee -a $out || { tail -5 gpurun_out/explore.err; exit 1; }
done
