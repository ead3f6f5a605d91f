#!/usr/bin/env python3
"""Measurement of the HIP InitialAligner on the benchmark data set's reads: one JSON line.

    python3 scripts/ia_bench.py [--workload tree_default] [--reads N] [--repeats K] [--cpu-reads M]

The reads are the simulated data set's reads cut to their repeat part (what ReadCutter hands to InitialAligner), the
template is the repeat.  `value` = matrix cells of the reference (read length x template length per read, IA:300-328) per
second of pia_align wall time with reads on the host; `fill` = the same over the two kernel passes alone.  The CPU
baseline is the oracle's IntoAligner restatement on a few of the same reads (one core)."""
import argparse
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="tree_default")
    ap.add_argument("--reads", type=int, default=0, help="use only the first N reads (0 = all)")
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--cpu-reads", type=int, default=3)
    ap.add_argument("--mem-budget-gb", type=float, default=0, help="bytes of direction bits per batch of pass 2 (0 = the library's default)")
    a = ap.parse_args()
    from repeatresolver_amd import datagen as dg
    from repeatresolver_amd.initial_aligner import InitialAligner
    t0 = time.time()
    cfg = dg.CONFIGS[a.workload]
    seq, _full, _starts, _cids, cut, _ = dg.simulate_dataset(cfg)
    templ = dg.ASCII[seq].tobytes()
    reads = [dg.ASCII[r].tobytes() for r in cut if r is not None and len(r) > 0]
    if a.reads:
        reads = reads[:a.reads]
    gen_s = time.time() - t0
    bases = sum(len(r) for r in reads)
    g = InitialAligner(templ)
    if a.mem_budget_gb:
        g.set_option("mem_budget", int(a.mem_budget_gb * 2**30))
    walls = []
    for _ in range(a.repeats + 1):                       # the first call is the warm-up
        s0 = g.stats()
        t0 = time.time()
        place, dist = g.align(reads)
        walls.append(time.time() - t0)
        s1 = g.stats()
    cells = s1["cells"] - s0["cells"]
    fill_ms = s1["fill_ms"] - s0["fill_ms"]
    wall = min(walls[1:])
    out = {"metric": "InitialAligner matrix cells/sec", "value": cells / wall, "unit": "cells/s", "workload": a.workload,
           "reads": len(reads), "bases": bases, "template": len(templ), "cells": cells, "wall_s": wall, "fill_ms": fill_ms,
           "fill_cells_per_s": cells / (fill_ms * 1e-3), "mean_error": float((dist / [max(len(r), 1) for r in reads]).mean()),
           "generate_s": round(gen_s, 1), "last_align_ms": {k: round(v, 1) for k, v in s1["last_align_ms"].items()}}
    if a.cpu_reads:
        lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "libiaoracle.so"))
        lib.iao_align.restype = ctypes.c_long
        lib.iao_align.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(ctypes.c_int), ctypes.c_void_p]
        step = max(1, len(reads) // a.cpu_reads)
        sample = list(range(0, len(reads), step))[:a.cpu_reads]
        ccells, t0 = 0, time.time()
        for j in sample:
            r = reads[j]
            al = (ctypes.c_int * len(r))()
            codes = ctypes.create_string_buffer(len(r) * len(templ))
            d = lib.iao_align(r, len(r), templ, len(templ), al, None, codes)
            assert d == int(dist[j]) and list(al) == list(place[j]), j
            ccells += len(r) * len(templ)
        cs = time.time() - t0
        out["cpu_baseline"] = {"value": ccells / cs, "unit": "cells/s", "cores": 1, "kind": "port",
                               "sample": f"{len(sample)} reads of the same set, {ccells} cells, {cs:.1f} s; results equal to the GPU's"}
    g.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
