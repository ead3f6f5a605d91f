"""Dev aid: how much of a row does one realignment change, round by round?  Per round of the CPU oracle: rows that move at
all, bases whose column changes (or that open a column), and the share of a row's 160-base stretches that hold such a base
-- what a commit invalidates of the speculative fills behind it.   usage: change_density.py [workload] [rounds]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Oracle
from repeatresolver_amd import datagen as dg
name = sys.argv[1] if len(sys.argv) > 1 else "tree_medium"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
step = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # look at every step-th row
o = Oracle(); lib = o.lib
rows = [bytes(r) for r in dg.make_msa(name)]
h = o.create(rows, 1000); del rows
lib.pwo_trim(h); lib.pwo_compact(h)
T = lib.pwo_rows(h)
t0 = time.time()
for r in range(rounds):
    moved = 0; nrows = 0; fb = []; fs = []; runs = []
    for k in range(T):
        lib.pwo_realign_row(h, k)
        if k % step: continue
        L = lib.pwo_dbg_L(h)
        if L < 1000: continue
        way = np.ctypeslib.as_array(lib.pwo_dbg_way(h), (L,))
        nc = np.ctypeslib.as_array(lib.pwo_dbg_newcol(h), (L,))
        ins = np.ctypeslib.as_array(lib.pwo_dbg_newins(h), (L,)).astype(bool)
        # compare in shift-invariant form: a base "changes" if its column relative to the row's first base changes or it opens a column
        ch = ins | ((nc - nc[0]) != (way - way[0]))
        ch2 = ins | (nc != way)
        nrows += 1; moved += bool(ch2.any())
        fb.append(ch2.mean())
        seg = np.add.reduceat(ch2.astype(int), np.arange(0, L, 160)) > 0
        fs.append(seg.mean())
        # longest run of consecutive touched stretches
        best = cur = 0
        for v in seg:
            cur = cur + 1 if v else 0; best = max(best, cur)
        runs.append(best / max(1, len(seg)))
    print("round %d (%.0f s): rows looked at %d, moved %.0f %%; bases that change: mean %.1f %% median %.1f %%; 160-base stretches with a change: mean %.0f %% median %.0f %%; longest run of such stretches / stretches: mean %.0f %%" %
          (r + 1, time.time() - t0, nrows, 100.0 * moved / max(1, nrows), 100 * np.mean(fb), 100 * np.median(fb), 100 * np.mean(fs), 100 * np.median(fs), 100 * np.mean(runs)), flush=True)
