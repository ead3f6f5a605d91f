"""dev (GPU box): one stress case again, with variations of its options; where does it first differ from the oracle?"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import Oracle
from repeatresolver_amd import datagen as dg
from repeatresolver_amd.realigner import PWReAligner
oracle = Oracle(); lib = oracle.lib
cfg = dg.SimConfig(kind='Tree', copies=5, coverage=14, difference=0.005, repeat_len=478, flank=676, length_scale=0.06, min_aligned=102, seed=30351)
rows = [bytes(r) for r in dg.build_msa(dg.simulate(cfg))]
print("rows", len(rows), "x", len(rows[0]), "max L", max(sum(c in b"acgt" for c in r) for r in rows), flush=True)
base = {'window': 1, 'seg_rows': 128, 'seg_max': 64, 'warm_pct': 20, 'ptrace': 2, 'waves': 9, 'onewg': 0, 'seg_align': 16, 'slack': 8192}
bw = 1500
def run(label, **over):
    o = dict(base); o.update(over)
    b = o.pop("bw", bw)
    g = PWReAligner(rows, bandwidth=b, **o); g.trim_ends()
    h = oracle.create(rows, b); lib.pwo_trim(h)
    res = "ok"
    for rnd in range(3):
        for k in range(len(rows)):
            lib.pwo_realign_row(h, k); g.realign_row(k)
            L = lib.pwo_dbg_L(h)
            if L == 0: continue
            d = g.debug_last_job()
            exp_new = [(lib.pwo_dbg_newcol(h)[x] << 1) | lib.pwo_dbg_newins(h)[x] for x in range(L)]
            way = [lib.pwo_dbg_way(h)[x] for x in range(L)]
            if d["way"] != way or d["entry"] != lib.pwo_dbg_entry(h) or d["newcol"] != exp_new:
                nd = sum(1 for a, b2 in zip(d["newcol"], exp_new) if a != b2)
                res = f"DIFF round {rnd} row {k} L {L}: way_ok {d['way'] == way} entry {d['entry']} vs {lib.pwo_dbg_entry(h)} newcol diffs {nd} first {next((i for i,(a,b2) in enumerate(zip(d['newcol'], exp_new)) if a != b2), -1)}"
                break
        if res != "ok": break
    st = g.stats()
    print(label, o, "->", res, "| seg_jobs", st["seg_jobs"], "fails", st["seg_fails"], flush=True)
    lib.pwo_destroy(h); g.close()
run("as found")
run("warm 190", warm_pct=190)
run("one piece", seg_rows=0)
run("ptrace 1", ptrace=1)
run("ptrace 0", ptrace=0)
run("bw 1000", bw=1000)
run("bw 1000 w5", bw=1000, waves=5)
