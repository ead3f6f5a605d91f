"""Dev aid (not product): does the banded fill of PW:1493-1513 FORGET where it started?

The fill is a min-plus recurrence; started at DP row x0 from an arbitrary row vector (here: the free start of
PW:265, all zeros) instead of the true scores of row x0 - 1, its row vectors become PARALLEL to the true ones
(equal up to one additive constant over the whole band) after some rows -- from there on every comparison the
traceback record is made of has the same outcome.  This script measures after how many rows that happens, with
the CPU oracle supplying Way[], the bases and the tallies of real realignments.  (Rank convergence of tropical
DP, Maleki/Musuvathi/Mytkowicz, PPoPP 2014; here the question is how long it takes for THIS band and cost.)

    python scripts/dev/rank_convergence.py tree_medium 40 8      # workload, rows to look at, starts per row
    python scripts/dev/rank_convergence.py /path/to/file.msa 6 8 # a text MSA (full scale: the input of gen_fullscale)
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import Oracle  # noqa: E402
from repeatresolver_amd import datagen as dg  # noqa: E402

INF = np.int64(1) << 60
B, H = 1000, 500
MODES = [int(m) for m in os.environ.get('MODES', '-1,0,1000,1008').split(',')]


class Fill:
    def __init__(self, way, seq, tal, W):
        self.way, self.seq, self.W = way, seq, W
        self.S = tal.astype(np.int64)                     # [W][6]
        self.G = np.cumsum(self.S[:, 4])
        up = np.maximum(self.S[:, 5], np.concatenate(([0], self.S[:-1, 5])))
        up[0] = INF
        up[W - 1] = INF
        self.up = up
        self.L = len(way)

    def out_prev(self, prev, ys, last):
        """Out(x-1, ys) of PW:249-303; prev = (a_p, Mp) or None for the free start (all zeros)."""
        if prev is None:
            return np.zeros(len(ys), dtype=np.int64)
        if isinstance(prev, tuple) and prev[0] == "src":
            # start vector: 0 inside [c - k, c + k], INF elsewhere
            _, c, k = prev
            r = np.full(len(ys), INF, dtype=np.int64)
            if k >= 3000:                                 # the k - 3000 columns left of c at the price of the gaps between them and c
                m_ = (ys >= c - (k - 3000)) & (ys <= c) & (ys >= 0)
                r[m_] = self.G[c] - self.G[ys[m_]]
            elif k >= 2000:                               # the k - 2000 columns left of c for nothing
                r[(ys >= c - (k - 2000)) & (ys <= c)] = 0
            elif k >= 1000:                               # half-open: free up to c + (k - 1000), unreachable right of it
                r[ys <= c + (k - 1000)] = 0
            else:
                r[(ys >= c - k) & (ys <= c + k)] = 0
            return r
        a_p, Mp = prev
        Bp = len(Mp)
        r = np.full(len(ys), INF, dtype=np.int64)
        inb = (ys >= a_p) & (ys < a_p + Bp)
        r[inb] = Mp[ys[inb] - a_p]
        ext = ys >= a_p + Bp
        if ext.any():
            base = Mp[Bp - 1]
            r[ext] = base if last else np.minimum(base + self.G[ys[ext]] - self.G[a_p + Bp - 1], INF)
        r[ys < 0] = INF
        return r

    def row(self, x, prev):
        a = max(0, self.way[x] - H)
        Bx = min(B, self.W - a)
        ys = np.arange(a, a + Bx)
        last = False                                       # (row x - 1 is never the last row)
        diag = self.out_prev(prev, ys - 1, last) + self.S[ys, self.seq[x]]
        upc = self.out_prev(prev, ys, last) + self.up[ys]
        t = np.minimum(np.minimum(diag, upc), INF)
        g = self.G[ys]
        M = np.minimum(g + np.minimum.accumulate(t - g), INF)
        return a, M


def parallel(Mt, Ms):
    ft, fs = Mt < INF // 2, Ms < INF // 2
    if not np.array_equal(ft, fs):
        return False
    if not ft.any():
        return True
    d = Ms[ft] - Mt[ft]
    return bool((d == d[0]).all())


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "tree_medium"
    nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    nstarts = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    o = Oracle()
    lib = o.lib
    if os.path.exists(name):
        lib.pwo_load.restype = ctypes.c_void_p
        lib.pwo_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        h = lib.pwo_load(name.encode(), B, None, 0)
    else:
        rows = [bytes(r) for r in dg.make_msa(name)]
        h = o.create(rows, B)
    lib.pwo_trim(h)
    lib.pwo_compact(h)
    rng = np.random.default_rng(1)
    lens_all = []
    bymode = {}
    T = lib.pwo_rows(h)
    for k in range(skip + nrows):
        lib.pwo_realign_row(h, k % T)                      # (skip >= T: the rows of a later round)
        if k < skip:
            continue
        L = lib.pwo_dbg_L(h)
        if L < 1500:
            continue
        W = lib.pwo_dbg_W_at_fill(h)
        way = np.ctypeslib.as_array(lib.pwo_dbg_way(h), (L,)).copy()
        seq = np.ctypeslib.as_array(lib.pwo_dbg_seq(h), (L,)).copy()
        tal = np.ctypeslib.as_array(lib.pwo_dbg_tallies(h), (W * 6,)).copy().reshape(W, 6)
        f = Fill(way, seq, tal, W)
        t0 = time.time()
        true = []
        prev = None
        for x in range(L):
            prev = f.row(x, prev)
            true.append(prev)
        # the restatement above against the oracle's own matrix, a few cells
        for x in (0, L // 3, L - 2):
            a, M = true[x]
            for j in (0, len(M) // 2, len(M) - 1):
                ref = lib.pwo_dbg_M(h, x, j)
                assert min(int(M[j]), int(INF)) == min(ref, int(INF)), (x, j, int(M[j]), ref)
        starts = sorted(int(s) for s in rng.integers(200, L - 1200, nstarts))
        res = []
        for x0 in starts:
          for mode in MODES:
            prev = None if mode < 0 else ("src", int(way[x0 - 1]), mode)
            conv = None
            for x in range(x0, min(L - 1, x0 + 6000)):
                prev = f.row(x, prev)
                if parallel(true[x][1], prev[1]):
                    # must STAY parallel (it does by construction once it is; checked for 20 rows anyway)
                    ok = True
                    p2 = prev
                    for x2 in range(x + 1, min(L - 1, x + 20)):
                        p2 = f.row(x2, p2)
                        ok = ok and parallel(true[x2][1], p2[1])
                    assert ok
                    conv = x - x0 + 1
                    break
            res.append(conv)
            bymode.setdefault(mode, []).append(conv if conv is not None else 10**6)
        lens_all += res
        print("row %d L=%d W=%d: rows until parallel from starts %s: %s   (%.1f s)" % (k, L, W, starts, res, time.time() - t0), flush=True)
    for mode, vals in bymode.items():
        v = np.array(vals)
        print("mode %d: n=%d median %d p90 %d p99 %d max %d" % (mode, len(v), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max()))
    v = np.array([r if r is not None else 10**6 for r in lens_all])
    if 0:
        print("all: n=%d median %d  p90 %d  p99 %d  max %d  not converged within 6000: %d" %
              (len(v), np.median(v), np.percentile(v, 90), np.percentile(v, 99), v.max(), int((v >= 10**6).sum())))


if __name__ == "__main__":
    main()
