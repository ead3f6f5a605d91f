#!/bin/bash
# idle time between the kernels of consecutive batches, from a rocprofv3 kernel trace (dev tool)
WL=${1:-tree_medium}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gp -- python bench.py --workload $WL --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/gp.json 2> gpurun_out/gp.err
python - <<PY
import csv, glob
rows = []
for fn in glob.glob("gpurun_out/gp/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:24]))
rows.sort()
gap = {}
busy = 0
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    key = n0 + " -> " + n1
    g = gap.setdefault(key, [0, 0]); g[0] += max(0, s1 - e0); g[1] += 1
for s, e, n in rows: busy += e - s
print("kernels busy %.3f s, span %.3f s" % (busy / 1e9, (rows[-1][1] - rows[0][0]) / 1e9))
for k, (t, c) in sorted(gap.items(), key=lambda kv: -kv[1][0])[:8]:
    print("%-60s total %.3f s  n %d  avg %.1f us" % (k, t / 1e9, c, t / c / 1e3))
PY
rm -rf gpurun_out/gp
