"""Dev aid (not product): can a segment of the fill START from a row vector that was stored earlier?  (round-3 verdict, item 1)

k_seg_check makes any start vector safe; the question is how many rows a segment must warm up before its rows are PARALLEL
to the true ones when it starts from
  batch   the vector the SAME row had at the same DP row ONE COMMIT earlier (the refill of a stale speculative job: the row
          before it in the k loop has been committed in between), re-anchored at the column of the base before the segment;
  round   the vector the same row had at the same DP row in the round before;
  cell    (for comparison) the one-cell start the product uses.
The CPU oracle supplies Way[], the bases, the tallies and the true matrix of real realignments (oracle/pw_oracle.c,
pwo_fill_only: the fill of a row against the state as it is, nothing committed).

    python scripts/dev/stored_vectors.py tree_default batch 0 40        # workload, mode, first row, rows
    python scripts/dev/stored_vectors.py tree_default round 2000 24 1   # ... rounds to run before the first measured one
"""
import ctypes
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
from conftest import Oracle  # noqa: E402
from repeatresolver_amd import datagen as dg  # noqa: E402
from start_vectors import Fill, parallel, INF, B, H  # noqa: E402

SEG = 160          # own rows of a segment (the product's default)
EVERY = 5          # every n-th boundary is measured


def setup(lib):
    lib.pwo_fill_only.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.pwo_dbg_Mrow.restype = ctypes.POINTER(ctypes.c_uint64)
    lib.pwo_dbg_Mrow.argtypes = [ctypes.c_void_p, ctypes.c_int]


def snapshot(lib, h):
    L = lib.pwo_dbg_L(h)
    W = lib.pwo_dbg_W_at_fill(h)
    way = np.ctypeslib.as_array(lib.pwo_dbg_way(h), (L,)).copy()
    seq = np.ctypeslib.as_array(lib.pwo_dbg_seq(h), (L,)).copy()
    return L, W, way, seq


def mrow(lib, h, x, way, W):
    a = max(0, int(way[x]) - H)
    n = min(B, W - a)
    v = np.ctypeslib.as_array(lib.pwo_dbg_Mrow(h, x), (B,))[:n].astype(np.int64, copy=True)
    v[(v < 0) | (v >= (1 << 60))] = INF
    return a, v


def boundaries(L):
    return [x for x in range(SEG, L - 1200, SEG)][::EVERY]


def rows_until_parallel(f, lib, h, way, W, x0, prev, limit=1500):
    """rows from x0 on until the fill started from `prev` (the scores of row x0 - 1) is parallel to the true one"""
    if prev is not None and not (isinstance(prev, tuple) and prev[0] == "src"):
        at, mt = mrow(lib, h, x0 - 1, way, W)
        if at == prev[0] and len(mt) == len(prev[1]) and parallel(mt, prev[1]):
            return 0
    for x in range(x0, min(f.L - 1, x0 + limit)):
        prev = f.row(x, prev)
        at, mt = mrow(lib, h, x, way, W)
        if len(mt) == len(prev[1]) and parallel(mt, prev[1]):
            return x - x0 + 1
    return 10 ** 6


def measure(lib, h, stored, label, res):
    """stored: {x0: (way_old[x0-1], vector of row x0-1)}; the oracle's last fill is the truth"""
    L, W, way, seq = snapshot(lib, h)
    tal = np.ctypeslib.as_array(lib.pwo_dbg_tallies(h), (W * 6,)).copy().reshape(W, 6)
    f = Fill(way, seq, tal, W)
    for x0, (w_old, v_old) in stored.items():
        if x0 >= L - 1200:
            continue
        a_new = max(0, int(way[x0 - 1]) - H)
        if a_new == 0 or a_new + B > W or len(v_old) != B:
            continue                                       # (clamped bands: left out)
        res.setdefault(label, []).append(rows_until_parallel(f, lib, h, way, W, x0, (a_new, v_old)))
        res.setdefault("cell", []).append(rows_until_parallel(f, lib, h, way, W, x0, ("src", int(way[x0 - 1]), 0)))
        res.setdefault("moved", []).append(int(way[x0 - 1]) - int(w_old))


def store(lib, h):
    L, W, way, _ = snapshot(lib, h)
    out = {}
    for x0 in boundaries(L):
        a, v = mrow(lib, h, x0 - 1, way, W)
        out[x0] = (int(way[x0 - 1]), v)
    return out


def report(res):
    for k_, v in res.items():
        v = np.array(v)
        if k_ == "moved":
            print("column of the base before the boundary moved by: median |d| %d, unchanged %.0f %%" % (np.median(np.abs(v)), 100.0 * np.mean(v == 0)))
            continue
        ok = v[v < 10 ** 6]
        print("%-6s n=%d  already parallel %.0f %%  <=16 rows %.0f %%  <=64 rows %.0f %%  median %d  p90 %d  max %d  never (1500 rows) %d" %
              (k_, len(v), 100.0 * np.mean(v == 0), 100.0 * np.mean(v <= 16), 100.0 * np.mean(v <= 64), np.median(v), np.percentile(v, 90),
               ok.max() if len(ok) else -1, int((v >= 10 ** 6).sum())), flush=True)


def main():
    name, mode = sys.argv[1], sys.argv[2]
    k0, n = int(sys.argv[3]), int(sys.argv[4])
    pre_rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    o = Oracle()
    lib = o.lib
    setup(lib)
    if os.path.exists(name):
        lib.pwo_load.restype = ctypes.c_void_p
        lib.pwo_load.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        h = lib.pwo_load(name.encode(), B, None, 0)
    else:
        rows = [bytes(r) for r in dg.make_msa(name)]
        h = o.create(rows, B)
        del rows
    lib.pwo_trim(h)
    lib.pwo_compact(h)
    T = lib.pwo_rows(h)
    t0 = time.time()
    for r in range(pre_rounds):
        lib.pwo_realign_round(h)
        print("round %d done, %.0f s" % (r + 1, time.time() - t0), flush=True)
    res = {}
    if mode == "batch":
        for k in range(k0):
            lib.pwo_realign_row(h, k)
        for k in range(k0 + 1, k0 + 1 + n):
            if lib.pwo_row_length(h, k) < 3000:
                lib.pwo_realign_row(h, k - 1)
                continue
            lib.pwo_fill_only(h, k)                        # row k as a speculative job sees it: row k - 1 not yet committed
            st = store(lib, h)
            lib.pwo_realign_row(h, k - 1)
            lib.pwo_fill_only(h, k)                        # ... and as its refill sees it
            measure(lib, h, st, "batch", res)
            print("row %d, %.0f s" % (k, time.time() - t0), flush=True)
            if (k - k0) % 8 == 0:
                report(res)
    else:
        sample = [k for k in range(k0, k0 + n) if lib.pwo_row_length(h, k) >= 3000]
        stored = {}
        for k in range(T):
            if k in sample:
                lib.pwo_fill_only(h, k)
                stored[k] = store(lib, h)
            lib.pwo_realign_row(h, k)
        print("round with the stored vectors done, %.0f s" % (time.time() - t0), flush=True)
        for k in range(max(sample) + 1):
            if k in sample:
                lib.pwo_fill_only(h, k)
                measure(lib, h, stored[k], "round", res)
                print("row %d, %.0f s" % (k, time.time() - t0), flush=True)
            lib.pwo_realign_row(h, k)
    report(res)


if __name__ == "__main__":
    main()
