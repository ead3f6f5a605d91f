"""CPU: the two facts the segmented fill of k_fill_v3 rests on, shown on the ORACLE's own matrix -- no GPU
code is involved, this pins the reasoning, the GPU parity tests pin the kernel.

The banded fill of PW:1493-1513 is restated here row by row in numpy (min-plus scan form) and first checked cell for cell
against the oracle's matrix (oracle/pw_oracle.c, itself pinned to the compiled reference by tests/golden).  Then, for DP
rows x0 in the middle of a row's fill and both start vectors the kernel knows -- the free start of PW:265 (every column
0) and ONE cell (score 0 in the column of the base before row x0, everything else unreachable):

 (1) the rows filled from that start become PARALLEL to the true rows (the same cells unreachable, one and the same
     difference in all others) after a bounded number of rows -- and stay so;
 (2) from the first parallel row on, every bit of the traceback record (A = "the cell's score equals its left candidate",
     C = "diagonal <= up", PW:1375) is the true bit.

(1) is what k_seg_check tests per segment on the GPU, (2) is why passing it is enough."""
import numpy as np
import pytest

from conftest import golden_input, split_rows

INF = np.int64(1) << 60


class _Fill:
    def __init__(self, way, seq, tal, W, B):
        self.way, self.seq, self.W, self.B, self.H = way, seq, W, B, B // 2
        self.S = tal.astype(np.int64)                        # [W][6]: w_con of every column, the row itself taken out
        self.G = np.cumsum(self.S[:, 4])
        up = np.maximum(self.S[:, 5], np.concatenate(([0], self.S[:-1, 5])))
        up[0] = INF                                          # PW:1505: no new column before the first / after the last
        up[W - 1] = INF
        self.up = up

    def out_prev(self, prev, ys):
        """Out(x-1, ys), PW:249-303; prev = (anf, M) of the row above, None = free start, ("cell", c) = one cell."""
        if prev is None:
            return np.zeros(len(ys), dtype=np.int64)
        if prev[0] == "cell":
            r = np.full(len(ys), INF, dtype=np.int64)
            r[ys == prev[1]] = 0
            return r
        a_p, Mp = prev
        Bp = len(Mp)
        r = np.full(len(ys), INF, dtype=np.int64)
        inb = (ys >= a_p) & (ys < a_p + Bp)
        r[inb] = Mp[ys[inb] - a_p]
        ext = ys >= a_p + Bp                                 # the virtual extension past the band's end, PW:285-295
        if ext.any():
            r[ext] = np.minimum(Mp[Bp - 1] + self.G[ys[ext]] - self.G[a_p + Bp - 1], INF)
        r[ys < 0] = INF
        return r

    def row(self, x, prev):
        a = max(0, self.way[x] - self.H)
        Bx = min(self.B, self.W - a)
        ys = np.arange(a, a + Bx)
        diag = np.minimum(self.out_prev(prev, ys - 1) + self.S[ys, self.seq[x]], INF)
        upc = np.minimum(self.out_prev(prev, ys) + self.up[ys], INF)
        t = np.minimum(diag, upc)
        g = self.G[ys]
        run = np.minimum.accumulate(t - g)
        M = np.minimum(g + run, INF)
        left = np.concatenate(([INF], M[:-1] + self.S[ys[1:], 4]))
        bitA = (M >= INF) | (M == np.minimum(left, INF))     # the score equals the left candidate
        bitC = diag <= upc
        return (a, M), bitA, bitC


def _parallel(Mt, Ms):
    ft, fs = Mt < INF // 2, Ms < INF // 2
    if not np.array_equal(ft, fs):
        return False
    d = Ms[ft] - Mt[ft]
    return (not ft.any()) or bool((d == d[0]).all())


@pytest.mark.parametrize("name,bw", [("toy_b_b1000", 1000), ("deep_b200", 200)])
def test_a_fill_forgets_its_start_and_then_makes_the_true_record(name, bw, oracle):
    rows = split_rows(golden_input(name))
    lib = oracle.lib
    h = oracle.create(rows, bw)
    lib.pwo_trim(h)
    lib.pwo_compact(h)
    done = 0
    worst = {"free": 0, "cell": 0}
    for k in range(len(rows)):
        lib.pwo_realign_row(h, k)
        L = lib.pwo_dbg_L(h)
        if L < 6 * bw // 4 + 200 and L < 600:
            continue
        W = lib.pwo_dbg_W_at_fill(h)
        way = np.ctypeslib.as_array(lib.pwo_dbg_way(h), (L,)).copy()
        seq = np.ctypeslib.as_array(lib.pwo_dbg_seq(h), (L,)).copy()
        tal = np.ctypeslib.as_array(lib.pwo_dbg_tallies(h), (W * 6,)).copy().reshape(W, 6)
        f = _Fill(way, seq, tal, W, bw)
        true, prev = [], None
        for x in range(L - 1):                               # (the last DP row has its own rule, PW:1386: not needed here)
            prev, bA, bC = f.row(x, prev)
            true.append((prev, bA, bC))
        for x in (0, L // 3, L - 2):                         # the restatement against the oracle's matrix
            a, M = true[x][0]
            for j in (0, len(M) // 2, len(M) - 1):
                assert min(int(M[j]), int(INF)) == min(lib.pwo_dbg_M(h, x, j), int(INF)), (k, x, j)
        span = L - 1
        for x0 in sorted({span // 5, span // 2, max(1, span - span // 3)}):
            for kind in ("free", "cell"):
                prev = None if kind == "free" else ("cell", int(way[x0 - 1]))
                conv = None
                for x in range(x0, span):
                    prev, bA, bC = f.row(x, prev)
                    par = _parallel(true[x][0][1], prev[1])
                    if conv is None and par:
                        conv = x
                    elif conv is not None:
                        assert par, (k, x0, kind, x)                          # (1) ... and stays parallel
                        fin = true[x][0][1] < INF // 2
                        assert np.array_equal(bA[fin], true[x][1][fin]), (k, x0, kind, x)    # (2) the true record
                        assert np.array_equal(bC[fin], true[x][2][fin]), (k, x0, kind, x)
                if conv is not None:
                    worst[kind] = max(worst[kind], conv - x0 + 1)
                else:
                    assert span - x0 < 3 * bw, (k, x0, kind)                  # only a start too close to the row's end may not get there
        done += 1
        if done == 4:
            break
    lib.pwo_destroy(h)
    assert done > 0
    assert 0 < worst["cell"] and 0 < worst["free"]
