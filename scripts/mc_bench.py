#!/usr/bin/env python3
"""Measurement of the HIP MaxCorrelation on the benchmark MSA after one realignment round (blanks at the row
ends give the columns their coverage structure): one JSON line.
    python3 scripts/mc_bench.py [--workload tree_default] [--rounds 1] [--cpu-columns 1500]
`value` = pairs of variations evaluated (four bit-set intersections + one hypergeometric tail each, MC:421-434) per second
of pmc_maxcorrs, MSA text on the host.  CPU baseline: the oracle's restatement on a slice of --cpu-columns columns of the
same MSA (one core), in the same unit."""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="tree_default")
    ap.add_argument("--rounds", type=int, default=1)
    ap.add_argument("--mincov", type=int, default=30)
    ap.add_argument("--cpu-columns", type=int, default=1500)
    a = ap.parse_args()
    import numpy as np
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.max_correlation import last_timing, max_correlations
    from repeatresolver_amd.pipeline import initial_msa
    from repeatresolver_amd.realigner import PWReAligner
    t0 = time.time()
    rows, info = initial_msa(dg.CONFIGS[a.workload])
    g = PWReAligner(rows, bandwidth=1000)
    g.trim_ends()
    for _ in range(a.rounds):
        g.realign_round()
    rows = g.export_rows()
    g.close()
    prep_s = time.time() - t0
    T, W = len(rows), len(rows[0])
    walls = []
    for _ in range(2):
        t0 = time.time()
        mc = max_correlations(rows, a.mincov)
        walls.append(time.time() - t0)
    tm = last_timing()
    out = {"metric": "MaxCorrelation variation pairs/sec", "value": tm["pairs"] / (tm["total_ms"] * 1e-3), "unit": "pairs/s",
           "workload": f"{a.workload}: pipeline MSA after {a.rounds} realignment round(s), {T} rows x {W} columns, mincov {a.mincov}",
           "pairs": tm["pairs"], "timing_ms": {k: round(v, 1) for k, v in tm.items() if k != "pairs"}, "wall_s": round(min(walls), 2),
           "variations_with_signal": int((mc > 0).sum()), "max": float(mc.max()), "prepare_s": round(prep_s, 1)}
    if a.cpu_columns:
        lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmcoracle.so"))
        lib.mco_maxcorrs.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        c0 = W // 2
        sl = [r[c0:c0 + a.cpu_columns] for r in rows]
        exp = np.zeros(a.cpu_columns * 5)
        t0 = time.time()
        lib.mco_maxcorrs(T, a.cpu_columns, b"".join(sl), a.mincov, exp.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
        cs = time.time() - t0
        got = max_correlations(sl, a.mincov)
        sp = last_timing()["pairs"]
        assert np.allclose(got, exp, rtol=0, atol=1e-9)
        out["cpu_baseline"] = {"value": sp / cs, "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": f"columns [{c0}, {c0 + a.cpu_columns}) of the same MSA, {sp} pairs, {cs:.1f} s; values equal to the GPU's within 1e-9"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
