#!/bin/bash
# dev (GPU box): distribution of the fill kernel's launch durations over the bench command (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/fh
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/fh -- python3 bench.py --no-cpu-baseline --steps 8 --warmup 3 "$@" > gpurun_out/fh.json 2> gpurun_out/fh.err
python3 - <<'PY'
import csv, glob
import numpy as np
rows = []
for fn in glob.glob("gpurun_out/fh/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(fn)):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for key in ("k_fill_v3", "k_trace_blk", "k_gather_c", "k_commit_finish"):
    d = np.array([(e - s) / 1e3 for n, s, e in rows if key in n])
    if not len(d): continue
    tot = d.sum()
    print(key, "launches", len(d), "mean %.1f median %.1f p90 %.1f p99 %.1f max %.0f us" % (d.mean(), np.median(d), np.percentile(d, 90), np.percentile(d, 99), d.max()))
    for lim in (300, 500, 1000):
        m = d > lim
        print("   > %4d us: %5d launches (%.2f %%), %.1f %% of the kernel's time" % (lim, m.sum(), 100.0 * m.mean(), 100.0 * d[m].sum() / tot))
PY
rm -rf gpurun_out/fh
