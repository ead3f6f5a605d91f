"""dev (GPU box): what a DP row costs -- single realignments (window 1) of a workload, product library.
Per realignment: the fill launch's duration (k_fill_v3 + k_seg_check, HIP events) over the row's length = the EFFECTIVE
time per DP row of the chain (the segments of a fill run side by side), in ns and cycles; with seg_rows=0 the same in one
piece = what a DP row costs the wave pipeline itself.   usage: rowcost.py [workload] [n] [key=value ...]"""
import os, sys
sys.path.insert(0, ".")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
args = [a for a in sys.argv[1:] if "=" not in a]
opts = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
wl = args[0] if args else "tree_default"
n = int(args[1]) if len(args) > 1 else 24
rows = [bytes(r) for r in dg.make_msa(wl)]
print(wl, len(rows), "rows x", len(rows[0]), opts)
for label, o in (("segments (default)", {}), ("one piece (seg_rows=0)", {"seg_rows": 0})):
    g = PWReAligner(rows, bandwidth=1000, window=1, profile=True, **{**opts, **o})
    g.trim_ends(); g.total_score()
    g.realign_row(0); g.reset_stats()
    tot_L = 0
    for k in range(1, n + 1):
        g.realign_row(k)
        tot_L += g.debug_last_job()["L"]
    st = g.stats()
    ns = 1e6 * st["fill_ms"] / tot_L
    print(f"{label}: {n} realignments, {tot_L} DP rows, fill {st['fill_ms']:.2f} ms in {st['fill_launches_timed']} launches = {ns:.1f} ns per DP row "
          f"= {ns * 2.4:.0f} cycles at 2.4 GHz; segments per fill {st['segs'] / max(1, st['seg_jobs']):.1f}, check failures {st['seg_fails']}")
    g.close()
