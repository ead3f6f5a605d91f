"""Dev aid (not product): how local are the changes a realignment makes?  Runs the CPU oracle over a workload and
prints, per round, how many bases move / columns open per realignment and how they cluster."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Oracle
from repeatresolver_amd import datagen as dg

name = sys.argv[1] if len(sys.argv) > 1 else "tree_medium"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 7
maxrows = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
bw = 1000
rows = [bytes(r) for r in dg.make_msa(name)]
o = Oracle(); lib = o.lib
h = o.create(rows, bw); lib.pwo_trim(h)
T = len(rows)
for rnd in range(rounds):
    t0 = time.time()
    nchg = []; nins = []; hull = []; span = []; clusters = []; unchanged = 0; ndel = []
    for k in range(min(T, maxrows)):
        Wb = None
        lib.pwo_realign_row(h, k)
        L = lib.pwo_dbg_L(h)
        if L == 0: continue
        way = np.ctypeslib.as_array(lib.pwo_dbg_way(h), (L,)).copy()
        nc = np.ctypeslib.as_array(lib.pwo_dbg_newcol(h), (L,)).copy()
        ni = np.ctypeslib.as_array(lib.pwo_dbg_newins(h), (L,)).copy()
        ch = (way != nc) | (ni != 0)
        n = int(ch.sum())
        if n == 0: unchanged += 1
        nchg.append(n); nins.append(int(ni.sum()))
        span.append(int(way[-1] - way[0] + 1))
        if n:
            idx = np.nonzero(ch)[0]
            cols = np.minimum(way[idx], nc[idx])
            hull.append(int(max(way[idx].max(), nc[idx].max()) - cols.min() + 1))
            cs = np.sort(cols)
            clusters.append(1 + int((np.diff(cs) > 1000).sum()))
    tot = len(nchg)
    print(f"round {rnd+1}: rows {tot} unchanged {unchanged} ({100*unchanged/tot:.1f}%) moved bases/row mean {np.mean(nchg):.1f} median {np.median(nchg):.0f} p90 {np.percentile(nchg,90):.0f} "
          f"ins cols/row mean {np.mean(nins):.2f} hull/span mean {np.mean(hull)/np.mean(span):.3f} clusters mean {np.mean(clusters):.2f} median {np.median(clusters):.0f} span {np.mean(span):.0f} "
          f"score {lib.pwo_total_score(h)} W {lib.pwo_width(h)} [{time.time()-t0:.0f}s]", flush=True)
