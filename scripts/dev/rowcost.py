"""dev (GPU box): ns per DP row of single realignments (window 1) on a workload, product library.  usage: rowcost.py [workload] [n]"""
import os, sys
sys.path.insert(0, ".")
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
wl = sys.argv[1] if len(sys.argv) > 1 else "tree_default"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = [bytes(r) for r in dg.make_msa(wl)]
print(wl, len(rows), "rows x", len(rows[0]))
for win in (1, 8):
    g = PWReAligner(rows, bandwidth=1000, window=win, profile=True)
    g.trim_ends(); g.total_score()
    if win == 1:
        tot_us = tot_L = 0
        for k in range(n):
            g.realign_row(k)
            mhz, us = g.debug_fill_clock()
            L = g.debug_last_job()["L"]
            tot_us += us; tot_L += L
            if k < 8:
                print(f"  row {k}: L={L} fill {us:.0f} us = {1e3*us/max(L,1):.1f} ns/DP row = {mhz*us/max(L,1):.0f} cycles/row at {mhz:.0f} MHz")
        print(f"window 1: {1e3*tot_us/tot_L:.1f} ns per DP row over {n} rows")
    else:
        g.realign_rows(0, 400)
        st = g.stats()
        import numpy as np
        print(f"window {win}: 400 rows, {st['fill_launches']} launches, avg {st['fill_ms']/st['fill_launches_timed']:.3f} ms; committed {st['rows_committed']} recomputed {st['rows_recomputed']}")
    g.close()
