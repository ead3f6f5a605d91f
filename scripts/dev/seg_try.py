"""dev (GPU box): the segmented fill -- parity against the oracle with small segments forced, and what it buys."""
import os, sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import Oracle, golden_input, split_rows
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner

oracle = Oracle(); lib = oracle.lib

def parity(rows, bw, nrows, **opts):
    g = PWReAligner(rows, bandwidth=bw, **opts)
    g.trim_ends()
    h = oracle.create(rows, bw); lib.pwo_trim(h)
    bad = 0
    for k in range(min(nrows, len(rows))):
        assert lib.pwo_realign_row(h, k) == 0
        g.realign_row(k)
        L = lib.pwo_dbg_L(h)
        if L == 0: continue
        d = g.debug_last_job()
        exp_new = [(lib.pwo_dbg_newcol(h)[x] << 1) | lib.pwo_dbg_newins(h)[x] for x in range(L)]
        if d["entry"] != lib.pwo_dbg_entry(h) or d["newcol"] != exp_new:
            bad += 1
            nd = sum(1 for a, b in zip(d["newcol"], exp_new) if a != b)
            first = next((i for i, (a, b) in enumerate(zip(d["newcol"], exp_new)) if a != b), -1)
            print("  MISMATCH row", k, "L", L, "entry", d["entry"], lib.pwo_dbg_entry(h), "newcol diffs", nd, "first at", first)
    st = g.stats()
    print("  rows", min(nrows, len(rows)), "bad", bad, "seg_jobs", st["seg_jobs"], "segs", st["segs"], "seg_fails", st["seg_fails"], "committed", st["rows_committed"], "recomputed", st["rows_recomputed"])
    lib.pwo_destroy(h); g.close()
    return bad

what = sys.argv[1] if len(sys.argv) > 1 else "all"
if what in ("all", "toy"):
    for name, bw in (("toy_b_b1000", 1000), ("toy_a_b1000", 1000), ("toy_a_b50", 50), ("lowcov_b300", 300), ("deep_b200", 200)):
        rows = split_rows(golden_input(name))
        for sr, wp in ((128, 200), (128, 20)):
            print(name, "seg_rows", sr, "warm_pct", wp, flush=True)
            parity(rows, bw, 60, seg_rows=sr, seg_max=16, warm_pct=wp)
if what in ("all", "medium"):
    rows = [bytes(r) for r in dg.make_msa("tree_medium")]
    print("tree_medium", len(rows), "x", len(rows[0]), flush=True)
    for sr in (1024, 512):
        print(" parity seg_rows", sr, flush=True)
        parity(rows, 1000, 40, seg_rows=sr)
    for sr, sm in ((0, 1), (2048, 16), (1024, 16), (512, 32), (256, 32)):
        g = PWReAligner(rows, bandwidth=1000, window=1, seg_rows=sr, seg_max=sm)
        g.trim_ends(); g.total_score()
        tot_us = tot_L = 0
        t0 = time.time()
        for k in range(24):
            g.realign_row(k)
            mhz, us = g.debug_fill_clock()
            tot_L += g.debug_last_job()["L"]
        g.reset_stats()
        t0 = time.time()
        g.realign_rows(24, 200)
        dt = time.time() - t0
        st = g.stats()
        print(f" seg_rows {sr} seg_max {sm}: window 1, 200 rows in {dt:.3f} s = {1e3*dt/200:.3f} ms/row; seg_jobs {st['seg_jobs']} segs {st['segs']} fails {st['seg_fails']}", flush=True)
        g.close()
    for sr, sm, win in ((0, 1, 8), (1024, 16, 8), (512, 32, 8), (512, 32, 4)):
        g = PWReAligner(rows, bandwidth=1000, window=win, seg_rows=sr, seg_max=sm, profile=True)
        g.trim_ends(); g.total_score()
        g.realign_rows(0, 100); g.reset_stats()
        t0 = time.time()
        g.realign_rows(100, 600)
        dt = time.time() - t0
        st = g.stats()
        print(f" seg_rows {sr} seg_max {sm} window {win}: 600 rows in {dt:.3f} s; batches {st['batches']} fill avg {st['fill_ms']/max(1,st['fill_launches_timed']):.3f} ms; committed {st['rows_committed']} recomputed {st['rows_recomputed']} fails {st['seg_fails']} cells/s {st['cells_reference']/dt:.3e}", flush=True)
        g.close()
