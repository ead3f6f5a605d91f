"""dev (GPU box): the bandwidth-1500 stress case, details of the first differing realignment."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import Oracle
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
oracle = Oracle(); lib = oracle.lib
cfg = dg.SimConfig(kind='Tree', copies=5, coverage=14, difference=0.005, repeat_len=478, flank=676, length_scale=0.06, min_aligned=102, seed=30351)
rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
for bw in (1500, 1400, 1300, 1600, 1280, 1536):
    g = PWReAligner(rows, bandwidth=bw, window=1, seg_rows=0); g.trim_ends()
    g3 = PWReAligner(rows, bandwidth=bw, window=1, fill=3); g3.trim_ends()
    h = oracle.create(rows, bw); lib.pwo_trim(h)
    res = "ok"
    for rnd in range(3):
        for k in range(len(rows)):
            lib.pwo_realign_row(h, k); g.realign_row(k); g3.realign_row(k)
            L = lib.pwo_dbg_L(h)
            if L == 0: continue
            d = g.debug_last_job(); d3 = g3.debug_last_job()
            exp_new = [(lib.pwo_dbg_newcol(h)[x] << 1) | lib.pwo_dbg_newins(h)[x] for x in range(L)]
            way = [lib.pwo_dbg_way(h)[x] for x in range(L)]
            assert d3["newcol"] == exp_new
            if d["newcol"] != exp_new:
                res = f"DIFF round {rnd} row {k} L {L} W {d['W']} entry {d['entry']}"
                print("bw", bw, res)
                for x in range(L):
                    if d["newcol"][x] != exp_new[x] or x < 4:
                        print("   base", x, "way", way[x], "anf", max(0, way[x] - bw // 2), "got", d["newcol"][x] >> 1, d["newcol"][x] & 1, "exp", exp_new[x] >> 1, exp_new[x] & 1)
                break
        if res != "ok": break
    print("bw", bw, "->", res, flush=True)
    lib.pwo_destroy(h); g.close(); g3.close()
