#!/bin/bash
# run a few realignments with the stamp-instrumented library (dev tool)
cp repeatresolver_amd/csrc/libpwr.so /tmp/libpwr_orig.so
cp repeatresolver_amd/csrc/libpwr_stamps.so repeatresolver_amd/csrc/libpwr.so
python - <<PY 2>&1 | tail -12
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
rows=[bytes(r) for r in dg.make_msa("tree_medium")]
g=PWReAligner(rows, bandwidth=1000, window=1, fill=int("${2:-3}"), waves=int("${1:-9}"))
g.trim_ends(); g.total_score()
for k in range(2): g.realign_row(k)
print(g.debug_fill_clock())
g.close()
PY
cp /tmp/libpwr_orig.so repeatresolver_amd/csrc/libpwr.so
