"""dev (GPU box): which rows fail their segment check?  window 1, per row: L, segments, failures.  usage: segfail_rows.py n [key=value...]"""
import sys
sys.path.insert(0, ".")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
n = int(sys.argv[1]); opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[2:]}
rows = [bytes(r) for r in dg.make_msa("tree_default")]
g = PWReAligner(rows, bandwidth=1000, window=1, **opts)
g.trim_ends(); g.total_score()
prev = 0; bad = []
for k in range(n):
    g.realign_row(k)
    st = g.stats()
    if st["seg_fails"] != prev:
        bad.append((k, g.debug_last_job()["L"], st["seg_fails"] - prev)); prev = st["seg_fails"]
print(opts, "rows", n, "fails", prev, "in rows", bad[:40])
g.close()
