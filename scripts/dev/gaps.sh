#!/bin/bash
# dev (GPU box): idle time between the kernels of the bench's batches, from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaps
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > gpurun_out/gaps.json 2> gpurun_out/gaps.err
python3 - <<'PY'
import csv, glob, collections
fn = glob.glob("gpurun_out/gaps/*/*kernel_trace.csv")[0]
rows = []
for r in csv.DictReader(open(fn)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:28]))
rows.sort()
# keep the realigner's kernels only, after the first k_commit_chain
names = ("k_gather_a", "k_gather_c", "void k_fill_v3", "k_seg_check", "k_trace_blk", "k_commit_chain")
rows = [r for r in rows if r[2].startswith(names)]
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gap[(a[2], b[2])].append(b[0] - a[1])
for r in rows: dur[r[2]].append(r[1] - r[0])
tot_gap = sum(sum(v) for v in gap.values()); tot_dur = sum(sum(v) for v in dur.values())
nb = len(dur["k_commit_chain"])
print("batches", nb, "kernel time per batch %.1f us, gaps per batch %.1f us" % (tot_dur / nb / 1e3, tot_gap / nb / 1e3))
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("  %-28s -> %-28s n=%6d mean %.2f us" % (k[0], k[1], len(v), sum(v) / len(v) / 1e3))
PY
rm -rf gpurun_out/gaps
